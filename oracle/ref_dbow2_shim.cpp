// TEST INFRASTRUCTURE -- C-linkage caller of the REFERENCE's own DBoW2::BowVector / DBoW2::FeatureVector.
//
// Built only where /root/reference exists, by oracle/Makefile target `_ref`, together with the reference's
//   Thirdparty/DBoW2/DBoW2/BowVector.cpp  and  Thirdparty/DBoW2/DBoW2/FeatureVector.cpp
// compiled from where they lie (both depend on the C++ standard library only); the output goes to
// oracle/_ref/libdbow2_ref.so (git-ignored, travels to the GPU box).  Nothing of the reference is copied here, and
// this file is not a stand-in for anything the reference needs: it only CALLS the two classes, so that
// tests/test_oracle_vs_reference_cpu.py can hold the oracle's restatement of
//   BowVector::addWeight / addIfNotExist / normalize   (BowVector.cpp:28-92)
//   FeatureVector::addFeature                          (FeatureVector.cpp:31-47)
// to the reference's object code, bit for bit.  The rest of TemplatedVocabulary::transform (tree descent, weights,
// the division by the word count when the scoring does not normalise) sits in TemplatedVocabulary.h, which needs
// OpenCV and stays a restatement (oracle/orb_vocab.c).
#include <cstdint>

#include "BowVector.h"      // -I <reference>/Thirdparty/DBoW2/DBoW2
#include "FeatureVector.h"

extern "C" {

// feed (ids[i], w[i]), i = 0..n-1 in this order; add_mode 0 = addWeight, 1 = addIfNotExist;
// norm 0 = none, 1 = normalize(L1), 2 = normalize(L2).  Writes the map in iteration order, returns its size.
int ref_bow_build(const uint32_t* ids, const double* w, int n, int add_mode, int norm, uint32_t* out_id, double* out_w)
{
    DBoW2::BowVector v;
    for (int i = 0; i < n; i++) {
        if (add_mode == 0) v.addWeight(ids[i], w[i]);
        else v.addIfNotExist(ids[i], w[i]);
    }
    if (norm == 1) v.normalize(DBoW2::L1);
    else if (norm == 2) v.normalize(DBoW2::L2);
    int m = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++m) { out_id[m] = it->first; out_w[m] = it->second; }
    return m;
}

// addFeature(node[i], feat[i]) in this order; CSR dump in map order: out_node[k], out_idx[out_start[k] .. out_start[k+1])
int ref_fv_build(const uint32_t* node, const uint32_t* feat, int n, uint32_t* out_node, int32_t* out_start, int32_t* out_idx)
{
    DBoW2::FeatureVector fv;
    for (int i = 0; i < n; i++) fv.addFeature(node[i], feat[i]);
    int k = 0, p = 0;
    out_start[0] = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++k) {
        out_node[k] = it->first;
        for (size_t j = 0; j < it->second.size(); j++) out_idx[p++] = (int32_t)it->second[j];
        out_start[k + 1] = p;
    }
    return k;
}

}  // extern "C"
