"""Synthetic DBoW2 vocabularies (the real ORBvoc.txt is absent from the reference checkout,
SURVEY.md F9): a k-ary tree of random ORB-like descriptors in loadFromTextFile order."""
import numpy as np


def make_vocab(rng, k, L, stop_fraction=0.05, ragged=True):
    """returns dict(parent, is_leaf, desc, weight) with nodes in breadth-first file order"""
    parent, is_leaf, desc, weight = [], [], [], []
    frontier = [(0, 0, rng.integers(0, 256, 32, dtype=np.uint8))]  # (node id, level, descriptor)
    next_id = 1
    while frontier:
        nxt = []
        for pid, lvl, pd in frontier:
            nchild = k if not ragged or lvl == 0 else int(rng.integers(2, k + 1))
            for _ in range(nchild):
                d = pd.copy()
                for b in rng.integers(0, 256, max(4, 60 >> lvl)):  # children = perturbed parents
                    d[b // 8] ^= np.uint8(1 << (b % 8))
                leaf = lvl + 1 == L or (ragged and lvl + 1 >= 2 and rng.uniform() < 0.1)
                parent.append(pid)
                is_leaf.append(leaf)
                desc.append(d)
                w = 0.0 if (leaf and rng.uniform() < stop_fraction) else float(rng.uniform(0.1, 9.0))
                weight.append(w)
                if not leaf:
                    nxt.append((next_id, lvl + 1, d))
                next_id += 1
        frontier = nxt
    return dict(parent=np.array(parent, np.int32), is_leaf=np.array(is_leaf, np.uint8), desc=np.stack(desc),
                weight=np.array(weight, np.float64), k=k, L=L)


def write_text(path, voc, scoring, weighting):
    """saveToTextFile format (TemplatedVocabulary.h:1430-1460): 'k L scoring weighting' then
    'parent is_leaf b0..b31 weight' per node"""
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (voc["k"], voc["L"], scoring, weighting))
        for i in range(len(voc["parent"])):
            f.write("%d %d %s %r\n" % (voc["parent"][i], int(voc["is_leaf"][i]), " ".join(str(int(b)) for b in voc["desc"][i]),
                                      float(voc["weight"][i])))


def naive_transform(voc, scoring, weighting, desc, levelsup):
    """independent Python re-derivation of DBoW2's transform (ordered dicts = std::map)"""
    n = len(voc["parent"])
    children = {i: [] for i in range(n + 1)}
    word = {}
    for i in range(n):
        children[int(voc["parent"][i])].append(i + 1)
        if voc["is_leaf"][i]:
            word[i + 1] = len(word)
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint8)
    bow, fv = {}, {}
    for fi, f in enumerate(desc):
        node, lvl, nid = 0, 0, 0
        nid_level = voc["L"] - levelsup
        while True:
            lvl += 1
            ch = children[node]
            ds = [int(lut[np.bitwise_xor(f, voc["desc"][c - 1])].sum()) for c in ch]
            node = ch[int(np.argmin(ds))]  # argmin = first minimum
            if lvl == nid_level:
                nid = node
            if not children[node]:
                break
        w = float(voc["weight"][node - 1])
        if w > 0:
            wid = word[node]
            if weighting in (0, 1):
                bow[wid] = bow.get(wid, 0.0) + w if wid in bow else w
            else:
                bow.setdefault(wid, w)
            fv.setdefault(nid, []).append(fi)
    ids = sorted(bow)
    vals = [bow[i] for i in ids]
    must = scoring != 5
    if weighting in (0, 1) and vals and not must:
        vals = [v / float(len(vals)) for v in vals]
    if must:
        norm = 0.0
        if scoring != 1:
            for v in vals:
                norm += abs(v)
        else:
            for v in vals:
                norm += v * v
            norm = float(np.sqrt(np.float64(norm)))
        if norm > 0:
            vals = [v / norm for v in vals]
    nodes = sorted(fv)
    start = [0]
    idx = []
    for nd in nodes:
        idx.extend(fv[nd])
        start.append(len(idx))
    return (np.array(ids, np.uint32), np.array(vals, np.float64)), (np.array(nodes, np.uint32), np.array(start, np.int32), np.array(idx, np.int32))
