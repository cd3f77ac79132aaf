"""The oracle's BowVector / FeatureVector assembly against the REFERENCE's own object code.

oracle/_ref/libdbow2_ref.so is DBoW2's BowVector.cpp + FeatureVector.cpp compiled from /root/reference where they lie
(oracle/Makefile target _ref; these two files need the C++ standard library only -- everything else on the path
includes OpenCV and is unbuildable here) plus oracle/ref_dbow2_shim.cpp, a caller of the two classes.  It is built in
the authoring container and travels to the GPU box as a prebuilt library; /root/reference is not read at test time.

What this pins, bit for bit (float64 compared with ==):
  BowVector::addWeight / addIfNotExist   accumulation of repeated words, in feature order
  BowVector::normalize(L1 | L2)          the exact sequence of double operations
  FeatureVector::addFeature              grouping and order of feature indices per node
i.e. the part of TemplatedVocabulary::transform that turns the per-feature (word, weight, node) stream into the two
outputs.  NOT pinned by it (TemplatedVocabulary.h needs OpenCV): the tree descent, the weights, and the division by
the number of words when the scoring does not normalise -- that step is done here, explicitly, between the two
reference calls.  The GPU path is held to the oracle in tests/test_gpu_vocab.py, which closes GPU = oracle = reference
for this slice.

What each test can see (checked by mutating the comparison, not committed): on streams that come out of a vocabulary
every occurrence of a word carries the SAME weight, so the order in which a word's weights are added cannot show
(reversing it: 0 of 12 configurations differ) -- test_transform_assembly... does detect the wrong norm (10 of 10),
first-instead-of-sum (10 of 10) and the norm summed over the words in another order (8 of 10).  The accumulation order
itself is pinned by test_summation_order... with unequal weights per word, where reversing the stream changes the bits."""
import numpy as np
import pytest

from vocab_cases import make_vocab


@pytest.fixture(scope="module")
def ref(oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref/libdbow2_ref.so absent and no /root/reference to build it from")
    return oracle


def reference_assembly(ref, word, w, nid, scoring, weighting):
    """TemplatedVocabulary::transform (:1127-1194) with the reference's BowVector / FeatureVector doing their part"""
    keep = w > 0                                   # `if(w > 0)` :1162 / :1178
    feat = np.nonzero(keep)[0].astype(np.uint32)
    tf = weighting in (0, 1)                       # TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    must = scoring != 5                            # DotProductScoring::mustNormalize is false
    if tf and not must:
        ids, val = ref.ref_bow_build(word[keep], w[keep], False, 0)
        if len(val):
            val = val / np.float64(len(val))       # `vit->second /= nd` :1171-1174 (TemplatedVocabulary.h, restated)
    else:
        ids, val = ref.ref_bow_build(word[keep], w[keep], not tf, (2 if scoring == 1 else 1) if must else 0)
    return (ids, val), ref.ref_fv_build(nid[keep], feat)


@pytest.mark.parametrize("weighting", [0, 1, 2, 3])
@pytest.mark.parametrize("scoring", [0, 1, 2, 3, 4, 5])
def test_transform_assembly_equals_reference_object_code(ref, scoring, weighting):
    rng = np.random.default_rng(700 + scoring * 4 + weighting)
    voc = make_vocab(rng, 5, 3)   # 125 words: 400 features repeat most of them
    V = ref.Vocabulary(5, 3, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    desc = rng.integers(0, 256, size=(400, 32), dtype=np.uint8)
    for levelsup in (0, 1, 3):
        word, w, nid = V.transform_one(desc, levelsup)
        (wi, wv), (fn, fs, fi) = V.transform(desc, levelsup)
        (ri, rv), (rn, rs, rx) = reference_assembly(ref, word, w, nid, scoring, weighting)
        assert np.array_equal(wi, ri)
        assert wv.tobytes() == rv.tobytes(), "BowVector values differ from DBoW2's in the last bits"
        assert np.array_equal(fn, rn) and np.array_equal(fs, rs) and np.array_equal(fi, rx)


@pytest.mark.parametrize("norm", [1, 2])
def test_summation_order_matters_and_is_the_references(ref, norm):
    """weights spanning 30 orders of magnitude on few words: any other accumulation or norm order shows in the bits"""
    rng = np.random.default_rng(norm)
    n = 3000
    ids = rng.integers(0, 40, size=n).astype(np.uint32)
    w = (10.0 ** rng.uniform(-15, 15, size=n)) * rng.choice([1.0, 1.0 + 2 ** -40, 3.0], size=n)
    ri, rv = ref.ref_bow_build(ids, w, False, norm)
    # the oracle's rule, restated in numpy float64 in the same order: per word in feature order, then the norm in map order
    acc = {}
    for i in range(n):
        acc[int(ids[i])] = acc[int(ids[i])] + w[i] if int(ids[i]) in acc else np.float64(w[i])
    keys = sorted(acc)
    val = np.array([acc[k] for k in keys], np.float64)
    s = np.float64(0.0)
    for x in val:
        s = s + (abs(x) if norm == 1 else x * x)
    if norm == 2:
        s = np.sqrt(s)
    val = val / s
    assert np.array_equal(ri, np.array(keys, np.uint32)) and rv.tobytes() == val.tobytes()
    # and it IS order sensitive: summing the same weights per word in reverse feature order changes the bits
    _, rv_rev = ref.ref_bow_build(ids[::-1], w[::-1], False, norm)
    assert rv_rev.tobytes() != rv.tobytes()


def test_add_if_not_exist_keeps_the_first_weight(ref):
    ids = np.array([7, 3, 7, 3, 9], np.uint32)
    w = np.array([0.5, 0.25, 100.0, 200.0, 0.125])
    ri, rv = ref.ref_bow_build(ids, w, True, 0)
    assert ri.tolist() == [3, 7, 9] and rv.tolist() == [0.25, 0.5, 0.125]
    rn, rs, rx = ref.ref_fv_build(np.array([5, 2, 5, 2, 2], np.uint32), np.array([10, 11, 12, 13, 14], np.uint32))
    assert rn.tolist() == [2, 5] and rs.tolist() == [0, 3, 5] and rx.tolist() == [11, 13, 14, 10, 12]
