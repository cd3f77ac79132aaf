"""Vocabulary-tree descent: the oracle against an independent Python re-derivation of DBoW2's
transform (the DBoW2 source is vendored in the reference, so this arithmetic is pinned by source)."""
import numpy as np
import pytest

from vocab_cases import make_vocab, naive_transform


@pytest.mark.parametrize("scoring,weighting", [(0, 0), (1, 1), (5, 0), (0, 2), (5, 3), (2, 1)])
def test_oracle_transform_equals_naive(oracle, scoring, weighting):
    rng = np.random.default_rng(40 + scoring * 4 + weighting)
    voc = make_vocab(rng, 6, 4)
    V = oracle.Vocabulary(6, 4, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    desc = rng.integers(0, 256, size=(300, 32), dtype=np.uint8)
    for levelsup in (0, 2, 4, 6):
        (wi, wv), (fn, fs, fi) = V.transform(desc, levelsup)
        (ni, nv), (nn, ns, nx) = naive_transform(voc, scoring, weighting, desc, levelsup)
        assert np.array_equal(wi, ni) and np.array_equal(wv, nv)          # doubles bit-for-bit
        assert np.array_equal(fn, nn) and np.array_equal(fs, ns) and np.array_equal(fi, nx)
    if scoring == 0:
        assert abs(wv.sum() - 1.0) < 1e-12
