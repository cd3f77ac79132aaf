/*
 * orb_vocab.c -- CPU ORACLE (test infrastructure only) of the vocabulary-tree descent that
 * sits between extraction and SearchByBoW.
 *
 * Restates /root/reference/SingleRobotScenario/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h
 *   transform(features, BowVector&, FeatureVector&, levelsup)   :1127-1194
 *   transform(feature, word_id, weight, nid, levelsup)          :1218-1259
 *   loadFromTextFile node numbering                              :1338-1425
 * and BowVector::addWeight/addIfNotExist/normalize (BowVector.cpp:34-84),
 * FeatureVector::addFeature (FeatureVector.cpp:31-45), FORB::distance (FORB.cpp:81-101).
 * DBoW2 is vendored in the reference, so this arithmetic is fully visible (pinned by source).
 */
#include "orb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

struct OrcVocab {
    int k, L, scoring, weighting;
    int n_nodes;          /* including the root (id 0) */
    int32_t* child_start; /* n_nodes + 1 */
    int32_t* child_idx;   /* children in push_back order */
    uint8_t* desc;        /* n_nodes x 32 (root unused) */
    int32_t* word_id;     /* -1 for inner nodes */
    double* weight;
    int n_words;
};

OrcVocab* orc_vocab_create(int k, int L, int scoring, int weighting, int n, const int32_t* parent,
                           const uint8_t* is_leaf, const uint8_t* desc, const double* weight)
{
    OrcVocab* v = (OrcVocab*)calloc(1, sizeof(OrcVocab));
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
    v->n_nodes = n + 1;
    v->child_start = (int32_t*)calloc((size_t)n + 2, sizeof(int32_t));
    v->child_idx = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
    v->desc = (uint8_t*)calloc((size_t)(n + 1) * 32, 1);
    v->word_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
    v->weight = (double*)calloc((size_t)n + 1, sizeof(double));
    int32_t* cnt = (int32_t*)calloc((size_t)n + 2, sizeof(int32_t));
    for (int i = 0; i < n; i++) {
        if (parent[i] < 0 || parent[i] > i) { /* a parent precedes its children in the file */
            free(cnt); orc_vocab_free(v); return NULL;
        }
        cnt[parent[i]]++;
    }
    for (int i = 0; i <= n; i++) v->child_start[i + 1] = v->child_start[i] + cnt[i];
    memset(cnt, 0, sizeof(int32_t) * (size_t)(n + 1));
    v->word_id[0] = -1;
    for (int i = 0; i < n; i++) {
        const int nid = i + 1, pid = parent[i];
        v->child_idx[v->child_start[pid] + cnt[pid]++] = nid; /* children.push_back(nid) */
        memcpy(v->desc + (size_t)nid * 32, desc + (size_t)i * 32, 32);
        v->weight[nid] = weight[i];
        v->word_id[nid] = is_leaf[i] ? v->n_words++ : -1;
    }
    free(cnt);
    return v;
}

void orc_vocab_free(OrcVocab* v)
{
    if (!v) return;
    free(v->child_start); free(v->child_idx); free(v->desc); free(v->word_id); free(v->weight); free(v);
}

/* :1218-1259 */
static void transform_one(const OrcVocab* v, const uint8_t* f, int levelsup, uint32_t* word, double* w, uint32_t* nid)
{
    const int nid_level = v->L - levelsup;
    if (nid_level <= 0) *nid = 0;
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const int cs = v->child_start[final_id], ce = v->child_start[final_id + 1];
        final_id = v->child_idx[cs];
        double best_d = orc_descriptor_distance(f, v->desc + (size_t)final_id * 32);
        for (int c = cs + 1; c < ce; c++) {
            const int id = v->child_idx[c];
            const double d = orc_descriptor_distance(f, v->desc + (size_t)id * 32);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (current_level == nid_level) *nid = (uint32_t)final_id;
    } while (v->child_start[final_id + 1] > v->child_start[final_id]);
    *word = (uint32_t)v->word_id[final_id];
    *w = v->weight[final_id];
}

/* the per-feature result of the descent, for tests that feed the REFERENCE's BowVector / FeatureVector classes
 * (oracle/_ref, tests/test_oracle_vs_reference_cpu.py) with the same (word, weight, node) stream */
void orc_vocab_transform_one(const OrcVocab* v, const uint8_t* f, int levelsup, uint32_t* word, double* w, uint32_t* nid)
{
    *nid = 0;
    transform_one(v, f, levelsup, word, w, nid);
}

typedef struct { uint32_t key; uint32_t feat; double w; } Ent;
static int cmp_ent(const void* a, const void* b)
{
    const Ent *x = (const Ent*)a, *y = (const Ent*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->feat < y->feat ? -1 : (x->feat > y->feat ? 1 : 0);
}

/* :1127-1194 */
int orc_vocab_transform(const OrcVocab* v, const uint8_t* desc, int n, int levelsup,
                        uint32_t* word_id, double* word_w, int* n_words,
                        uint32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int* n_fv)
{
    *n_words = 0; *n_fv = 0;
    fv_start[0] = 0;
    if (v->n_words == 0 || n <= 0) return 0;
    Ent* bw = (Ent*)malloc(sizeof(Ent) * (size_t)n);
    Ent* fv = (Ent*)malloc(sizeof(Ent) * (size_t)n);
    int m = 0;
    for (int i = 0; i < n; i++) {
        uint32_t id, nid = 0; double w;
        transform_one(v, desc + (size_t)i * 32, levelsup, &id, &w, &nid);
        if (w > 0) { bw[m].key = id; bw[m].feat = (uint32_t)i; bw[m].w = w; fv[m].key = nid; fv[m].feat = (uint32_t)i; fv[m].w = 0; m++; }
    }
    qsort(bw, (size_t)m, sizeof(Ent), cmp_ent);
    qsort(fv, (size_t)m, sizeof(Ent), cmp_ent);
    const int tf = v->weighting == 0 || v->weighting == 1; /* TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist */
    int nw = 0;
    for (int i = 0; i < m;) {
        int j = i;
        double acc = bw[i].w;
        for (j = i + 1; j < m && bw[j].key == bw[i].key; j++) if (tf) acc += bw[j].w; /* feature order */
        word_id[nw] = bw[i].key; word_w[nw] = acc; nw++;
        i = j;
    }
    const int must = v->scoring != 5;            /* DOT_PRODUCT does not normalise */
    const int l2 = v->scoring == 1;
    if (tf && nw > 0 && !must) { const double nd = (double)nw; for (int i = 0; i < nw; i++) word_w[i] /= nd; }
    if (must) {
        double norm = 0.0;
        if (!l2) { for (int i = 0; i < nw; i++) norm += fabs(word_w[i]); }
        else { for (int i = 0; i < nw; i++) norm += word_w[i] * word_w[i]; norm = sqrt(norm); }
        if (norm > 0.0) for (int i = 0; i < nw; i++) word_w[i] /= norm;
    }
    int nn = 0;
    for (int i = 0; i < m;) {
        int j = i;
        fv_node[nn] = fv[i].key;
        for (; j < m && fv[j].key == fv[i].key; j++) fv_idx[j] = (int32_t)fv[j].feat;
        fv_start[++nn] = j;
        i = j;
    }
    *n_words = nw; *n_fv = nn;
    free(bw); free(fv);
    return 0;
}
