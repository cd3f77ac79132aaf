// orbv_kernels.hip -- vocabulary-tree descent (DBoW2 transform) on gfx950, SURVEY.md 8(f) rank 2.
//
// Reference: /root/reference/SingleRobotScenario/Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h
//   transform(feature, word_id, weight, nid, levelsup) :1218-1259  -> k_voc_descend
//   transform(features, BowVector, FeatureVector, levelsup) :1127-1194 -> k_voc_aggregate
// called per frame by Frame::ComputeBoW (src/Frame.cc:395-402, levelsup = 4).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbv {

#ifdef ORBT_PHASE_TIMING  // tools/proj_phases.sh
__device__ unsigned long long g_orbvPhase[16];
#define ORBV_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_orbvPhase[i] = wall_clock64(); } while (0)
#else
#define ORBV_MARK(i) do { } while (0)
#endif

struct VocDev {
    const int32_t* childStart;  // n_nodes + 1: the children of node i are the nodes [childStart[i], childStart[i + 1]) (breadth-first numbering, orbv_create)
    const int32_t* childIdx;    // origId: the file's id of node i (what the FeatureVector is keyed by)
    const uint8_t* desc;        // n_nodes x 32
    const int32_t* wordId;      // -1 for inner nodes
    const double* weight;
    int32_t L, scoring, weighting;
};

__device__ __forceinline__ int ham256(const uint32_t q[8], const uint32_t* __restrict__ t)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d += __popc(q[i] ^ t[i]);
    return d;
}

// one thread per feature: L sequential k-way Hamming argmins, first child wins ties
__global__ void k_voc_descend(VocDev v, const uint8_t* __restrict__ desc, int n, int levelsup,
                              uint32_t* __restrict__ word, uint32_t* __restrict__ node, double* __restrict__ w)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t q[8];
    const uint32_t* qp = (const uint32_t*)(desc + (int64_t)i * 32);
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = qp[k];
    const int nidLevel = v.L - levelsup;
    uint32_t nid = 0;
    int finalId = 0, level = 0;
    do {
        ++level;
        const int cs = v.childStart[finalId], ce = v.childStart[finalId + 1];
        finalId = cs;
        int best = ham256(q, (const uint32_t*)(v.desc + (int64_t)cs * 32));
        for (int c = cs + 1; c < ce; c++) {
            const int d = ham256(q, (const uint32_t*)(v.desc + (int64_t)c * 32));
            if (d < best) { best = d; finalId = c; }
        }
        if (level == nidLevel) nid = (uint32_t)v.childIdx[finalId];
    } while (v.childStart[finalId + 1] > v.childStart[finalId]);
    word[i] = (uint32_t)v.wordId[finalId];
    node[i] = nid;
    w[i] = v.weight[finalId];
}

constexpr int kAggThreads = 1024;

__device__ __forceinline__ void bitonic_sort_lds(uint64_t* a, int P)
{
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += kAggThreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = a[i], y = a[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
}

// block-wide exclusive scan of flags over the sorted keys (boundary = new key value)
__device__ __forceinline__ int boundaries_scan(const uint64_t* a, int m, uint32_t* pos, int* wsum)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (m + kAggThreads - 1) / kAggThreads;
    const int b = min(tid * per, m), e = min(b + per, m);
    int s = 0;
    for (int i = b; i < e; i++) s += (i == 0 || (a[i] >> 32) != (a[i - 1] >> 32)) ? 1 : 0;
    int incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    __syncthreads();
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int k = 0; k < kAggThreads / 64; k++) { if (k < wave) woff += wsum[k]; tot += wsum[k]; }
    int run = woff + incl - s;
    for (int i = b; i < e; i++) {
        const bool nb = (i == 0 || (a[i] >> 32) != (a[i - 1] >> 32));
        if (nb) run++;
        pos[i] = (uint32_t)(run - 1);  // group index of entry i
    }
    __syncthreads();
    return tot;
}

// BowVector + FeatureVector of one descriptor set.  P = power of two >= max(n, 2), LDS: P * 36 bytes (ACC_LDS: keys, group
// index, word values, bucket-sort scratch) or P * 12.
// Sort of the valid keys (everything but ~0) of keys[0..P): a bucket sort instead of the bitonic network's 66 barrier
// steps (50 us of this kernel's 123 each time).  A key is (id << 32 | feature); compressed to (id - idMin) * 8192 + feature
// the keys of a frame are spread nearly evenly, so a linear map onto P buckets leaves about one key per bucket: histogram,
// scan, scatter, a short insertion sort per bucket -- six barriers.  tmp: P keys, cnt / cur: P ints each.
__device__ __forceinline__ void bucket_sort_lds(uint64_t* keys, uint64_t* tmp, int* cnt, int* cur, int P, int m, int* wsum, uint32_t* sMinMax)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { sMinMax[0] = 0xFFFFFFFFu; sMinMax[1] = 0u; }
    for (int i = tid; i < P; i += kAggThreads) { cnt[i] = 0; cur[i] = 0; }
    __syncthreads();
    {   // id range: per wave first (two thousand atomics on one LDS word serialise)
        uint32_t lo = 0xFFFFFFFFu, hi = 0u;
        for (int i = tid; i < P; i += kAggThreads) {
            const uint64_t k = keys[i];
            if (k != ~0ull) { lo = min(lo, (uint32_t)(k >> 32)); hi = max(hi, (uint32_t)(k >> 32)); }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, d)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, d)); }
        if (lane == 0) { atomicMin(&sMinMax[0], lo); atomicMax(&sMinMax[1], hi); }
    }
    __syncthreads();
    const uint32_t idMin = sMinMax[0];
    const double scale = (double)P / ((double)(sMinMax[1] - idMin + 1u) * 8192.0);
    auto bucket = [&](uint64_t k) {
        const double c = (double)((uint64_t)((uint32_t)(k >> 32) - idMin) * 8192ull + (uint32_t)k);
        return min((int)(c * scale), P - 1);   // monotone in the key
    };
    for (int i = tid; i < P; i += kAggThreads) { const uint64_t k = keys[i]; if (k != ~0ull) atomicAdd(&cnt[bucket(k)], 1); }
    __syncthreads();
    int carry = 0;
    for (int base = 0; base < P; base += kAggThreads) {  // exclusive scan of cnt
        const int i = base + tid;
        const int v = i < P ? cnt[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        __syncthreads();
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int k = 0; k < kAggThreads / 64; k++) { if (k < wave) woff += wsum[k]; tot += wsum[k]; }
        if (i < P) cnt[i] = carry + woff + incl - v;
        carry += tot;
    }
    __syncthreads();
    for (int i = tid; i < P; i += kAggThreads) {
        const uint64_t k = keys[i];
        if (k != ~0ull) { const int b = bucket(k); tmp[cnt[b] + atomicAdd(&cur[b], 1)] = k; }
    }
    // Crowded buckets -- many features on ONE word or vocabulary node (the linear map cannot part them: a ragged tree's
    // early leaves collect hundreds of a frame's features) -- are not left to one thread's insertion sort (quadratic: a
    // bucket of 500 keys cost 1.5 ms) but rank-sorted by the whole workgroup, one after the other.
    constexpr int kCrowd = 16;
    __shared__ int sBigN;
    __shared__ uint16_t sBig[256];   // more than kCrowd keys each: at most P / 17 <= 240 of them (P <= 4096 here)
    if (tid == 0) sBigN = 0;
    __syncthreads();
    for (int b = tid; b < P; b += kAggThreads) {
        const int s0 = cnt[b], e0 = s0 + cur[b];
        if (e0 - s0 > kCrowd) { const int at = atomicAdd(&sBigN, 1); if (at < 256) sBig[at] = (uint16_t)b; continue; }
        for (int i = s0 + 1; i < e0; i++) {
            const uint64_t x = tmp[i];
            int j = i - 1;
            while (j >= s0 && tmp[j] > x) { tmp[j + 1] = tmp[j]; j--; }
            tmp[j + 1] = x;
        }
    }
    __syncthreads();
    const int nBig = min(sBigN, 256);
    for (int bi = 0; bi < nBig; bi++) {   // (keys[] is free since the scatter: the sorted run is built there, then moved back)
        const int b = sBig[bi], s0 = cnt[b], k = cur[b];
        for (int e = tid; e < k; e += kAggThreads) {
            const uint64_t x = tmp[s0 + e];
            int rank = 0;
            for (int j = 0; j < k; j++) rank += tmp[s0 + j] < x ? 1 : 0;   // keys are distinct (the feature index is part of them)
            keys[s0 + rank] = x;
        }
        __syncthreads();
        for (int e = tid; e < k; e += kAggThreads) tmp[s0 + e] = keys[s0 + e];
        __syncthreads();
    }
    for (int i = tid; i < P; i += kAggThreads) keys[i] = i < m ? tmp[i] : ~0ull;
    __syncthreads();
}

// KEYS_MEM: more than 8192 descriptors (P * 12 bytes of keys and group indices no longer fit a CU's LDS): the same sort and scans on a
// scratch block in memory (gkeys: P * 12 bytes) -- DBoW2's transform takes any number of features (TemplatedVocabulary.h:1120-1160)
template <bool ACC_LDS, bool KEYS_MEM = false>  // word values staged in LDS (P <= 4096) or worked on in place in outW
__device__ __forceinline__ void voc_aggregate_body(const VocDev& v, int n, int P,
                                                   const uint32_t* __restrict__ word, const uint32_t* __restrict__ node,
                                                   const double* __restrict__ w,
                                                   uint32_t* __restrict__ outWord, double* __restrict__ outW,
                                                   uint32_t* __restrict__ fvNode, int32_t* __restrict__ fvStart,
                                                   int32_t* __restrict__ fvIdx, int32_t* __restrict__ counts, uint64_t* gkeys = nullptr)
{
    static_assert(!(ACC_LDS && KEYS_MEM), "the memory form keeps nothing in LDS");
    extern __shared__ __attribute__((aligned(16))) uint64_t alds[];
    uint64_t* keys; uint32_t* pos;
    if constexpr (KEYS_MEM) { keys = gkeys; pos = (uint32_t*)(gkeys + P); }
    else { keys = alds; pos = (uint32_t*)(alds + P); }
    double* accL = ACC_LDS ? (double*)(alds + P + P / 2) : outW;   // P >= 2
    uint64_t* tmp = alds + 2 * P + P / 2;       // ACC_LDS only: the bucket sort's scratch (P keys + 2 P ints)
    int* bcnt = (int*)(tmp + P);
    int* bcur = bcnt + P;
    __shared__ int wsum[kAggThreads / 64];
    __shared__ int sM;
    __shared__ uint32_t sMinMax[2];
    const int tid = threadIdx.x;
    const bool tf = v.weighting == 0 || v.weighting == 1;  // TF_IDF, TF: addWeight; IDF, BINARY: addIfNotExist
    const bool must = v.scoring != 5;                       // DotProductScoring does not normalise
    const bool l2 = v.scoring == 1;

    // ---- BowVector: sort (word, feature) of the features with w > 0 ("not stopped", :1156)
    for (int i = tid; i < P; i += kAggThreads)
        keys[i] = (i < n && w[i] > 0) ? (((uint64_t)word[i] << 32) | (uint32_t)i) : ~0ull;
    if (tid == 0) sM = 0;
    __syncthreads();
    {
        int c = 0;
        for (int i = tid; i < n; i += kAggThreads) c += w[i] > 0 ? 1 : 0;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
        if ((tid & 63) == 0) atomicAdd(&sM, c);
    }
    __syncthreads();
    ORBV_MARK(0);
    const int m = sM;
    if (ACC_LDS) bucket_sort_lds(keys, tmp, bcnt, bcur, P, m, wsum, sMinMax); else bitonic_sort_lds(keys, P);
    ORBV_MARK(1);
    const int nw = boundaries_scan(keys, m, pos, wsum);
    // the word values stay in LDS until they are final: the normalisation sum below is one thread walking them in
    // map (word id) order, which from memory cost a load latency per addend (0.19 ms of this kernel's 0.2)
    for (int i = tid; i < m; i += kAggThreads) {
        if (i == 0 || pos[i] != pos[i - 1]) {
            int cnt = 1;
            while (i + cnt < m && pos[i + cnt] == pos[i]) cnt++;
            const double wi = w[(uint32_t)keys[i]];
            double acc = wi;                     // addWeight: += in feature order (BowVector.cpp:34-46)
            if (tf) for (int k = 1; k < cnt; k++) acc += wi;
            outWord[pos[i]] = (uint32_t)(keys[i] >> 32);
            accL[pos[i]] = acc;
        }
    }
    __syncthreads();
    if (tf && nw > 0 && !must) {
        const double nd = (double)nw;
        for (int i = tid; i < nw; i += kAggThreads) accL[i] /= nd;
        __syncthreads();
    }
    ORBV_MARK(2);
    if (must) {  // BowVector::normalize: the sum runs in map (word id) order, sequentially
        __shared__ double sNorm;
        if (tid == 0) {
            double norm = 0.0;
            // one thread, map order (BowVector.cpp:60-79): 64 values per step come in with 32 wide LDS reads issued
            // together, then the dependent adds run from registers
            const double2* a2 = (const double2*)accL;
            int i = 0;
            if (!l2) {
                for (; i + 64 <= nw; i += 64) {
                    double2 v[32];
#pragma unroll
                    for (int k = 0; k < 32; k++) v[k] = a2[(i >> 1) + k];
#pragma unroll
                    for (int k = 0; k < 32; k++) { norm += fabs(v[k].x); norm += fabs(v[k].y); }
                }
                for (; i < nw; i++) norm += fabs(accL[i]);
            } else {
                for (; i + 64 <= nw; i += 64) {
                    double2 v[32];
#pragma unroll
                    for (int k = 0; k < 32; k++) v[k] = a2[(i >> 1) + k];
#pragma unroll
                    for (int k = 0; k < 32; k++) { norm += v[k].x * v[k].x; norm += v[k].y * v[k].y; }
                }
                for (; i < nw; i++) norm += accL[i] * accL[i];
                norm = sqrt(norm);
            }
            sNorm = norm;
        }
        __syncthreads();
        const double norm = sNorm;
        if (norm > 0.0) for (int i = tid; i < nw; i += kAggThreads) accL[i] /= norm;
        __syncthreads();
    }
    if (ACC_LDS) for (int i = tid; i < nw; i += kAggThreads) outW[i] = accL[i];
    __syncthreads();

    ORBV_MARK(3);
    // ---- FeatureVector: sort (node, feature); addFeature appends in feature order (:1159)
    for (int i = tid; i < P; i += kAggThreads)
        keys[i] = (i < n && w[i] > 0) ? (((uint64_t)node[i] << 32) | (uint32_t)i) : ~0ull;
    __syncthreads();
    ORBV_MARK(4);
    if (ACC_LDS) bucket_sort_lds(keys, tmp, bcnt, bcur, P, m, wsum, sMinMax); else bitonic_sort_lds(keys, P);
    ORBV_MARK(5);
    const int nf = boundaries_scan(keys, m, pos, wsum);
    for (int i = tid; i < m; i += kAggThreads) {
        fvIdx[i] = (int32_t)(uint32_t)keys[i];
        if (i == 0 || pos[i] != pos[i - 1]) { fvNode[pos[i]] = (uint32_t)(keys[i] >> 32); fvStart[pos[i]] = i; }
    }
    if (tid == 0) { fvStart[nf] = m; counts[0] = nw; counts[1] = nf; }
    ORBV_MARK(6);
}

__global__ __launch_bounds__(kAggThreads) void k_voc_aggregate_mem(VocDev v, int n, int P,
                                                                  const uint32_t* __restrict__ word, const uint32_t* __restrict__ node,
                                                                  const double* __restrict__ w,
                                                                  uint32_t* __restrict__ outWord, double* __restrict__ outW,
                                                                  uint32_t* __restrict__ fvNode, int32_t* __restrict__ fvStart,
                                                                  int32_t* __restrict__ fvIdx, int32_t* __restrict__ counts, uint64_t* gkeys)
{
    voc_aggregate_body<false, true>(v, n, P, word, node, w, outWord, outW, fvNode, fvStart, fvIdx, counts, gkeys);
}

template <bool ACC_LDS>
__global__ __launch_bounds__(kAggThreads) void k_voc_aggregate(VocDev v, int n, int P,
                                                              const uint32_t* __restrict__ word, const uint32_t* __restrict__ node,
                                                              const double* __restrict__ w,
                                                              uint32_t* __restrict__ outWord, double* __restrict__ outW,
                                                              uint32_t* __restrict__ fvNode, int32_t* __restrict__ fvStart,
                                                              int32_t* __restrict__ fvIdx, int32_t* __restrict__ counts)
{
    voc_aggregate_body<ACC_LDS>(v, n, P, word, node, w, outWord, outW, fvNode, fvStart, fvIdx, counts);
}

// ---- Frame::ComputeBoW for the frames of a frame set, one launch each: slot s of every array at + s * cap (fvStart:
// + s * (cap + 1), counts: + 2 s); the feature count is read on the device
struct VocSetArgs {
    const uint8_t* desc; const int32_t* n; int32_t cap, slot0, slotMod, P;
    uint32_t* word; uint32_t* node; double* w;                    // scratch of the descent
    uint32_t* outWord; double* outW;                              // BowVector
    uint32_t* fvNode; int32_t* fvStart; int32_t* fvIdx; int32_t* counts;  // FeatureVector as CSR, counts = {words, nodes}
    uint64_t* gkeys;                                              // frames of more than 8192 features: [slot][P * 12 bytes] of sort scratch in memory
};

__global__ void k_voc_descend_set(VocDev v, VocSetArgs a, int levelsup)
{
    const int slot = (a.slot0 + blockIdx.y) % a.slotMod;
    const int n = min(a.n[slot], a.cap);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t o = (int64_t)slot * a.cap;
    uint32_t q[8];
    const uint32_t* qp = (const uint32_t*)(a.desc + (o + i) * 32);
#pragma unroll
    for (int k = 0; k < 8; k++) q[k] = qp[k];
    const int nidLevel = v.L - levelsup;
    uint32_t nid = 0;
    int finalId = 0, level = 0;
    do {  // as k_voc_descend
        ++level;
        const int cs = v.childStart[finalId], ce = v.childStart[finalId + 1];
        finalId = cs;
        int best = ham256(q, (const uint32_t*)(v.desc + (int64_t)cs * 32));
        for (int c = cs + 1; c < ce; c++) {
            const int d = ham256(q, (const uint32_t*)(v.desc + (int64_t)c * 32));
            if (d < best) { best = d; finalId = c; }
        }
        if (level == nidLevel) nid = (uint32_t)v.childIdx[finalId];
    } while (v.childStart[finalId + 1] > v.childStart[finalId]);
    a.word[o + i] = (uint32_t)v.wordId[finalId];
    a.node[o + i] = nid;
    a.w[o + i] = v.weight[finalId];
}

template <bool ACC_LDS>
__global__ __launch_bounds__(kAggThreads) void k_voc_aggregate_set(VocDev v, VocSetArgs a)
{
    const int slot = (a.slot0 + blockIdx.x) % a.slotMod;
    const int n = min(a.n[slot], a.cap);
    const int64_t o = (int64_t)slot * a.cap;
    if (n == 0 || v.L == 0) {
        if (threadIdx.x == 0) { a.counts[2 * slot] = 0; a.counts[2 * slot + 1] = 0; a.fvStart[(int64_t)slot * (a.cap + 1)] = 0; }
        return;
    }
    voc_aggregate_body<ACC_LDS>(v, n, a.P, a.word + o, a.node + o, a.w + o, a.outWord + o, a.outW + o, a.fvNode + o,
                                a.fvStart + (int64_t)slot * (a.cap + 1), a.fvIdx + o, a.counts + 2 * slot);
}

__global__ __launch_bounds__(kAggThreads) void k_voc_aggregate_set_mem(VocDev v, VocSetArgs a)
{
    const int slot = (a.slot0 + blockIdx.x) % a.slotMod;
    const int n = min(a.n[slot], a.cap);
    const int64_t o = (int64_t)slot * a.cap;
    if (n == 0 || v.L == 0) {
        if (threadIdx.x == 0) { a.counts[2 * slot] = 0; a.counts[2 * slot + 1] = 0; a.fvStart[(int64_t)slot * (a.cap + 1)] = 0; }
        return;
    }
    voc_aggregate_body<false, true>(v, n, a.P, a.word + o, a.node + o, a.w + o, a.outWord + o, a.outW + o, a.fvNode + o,
                                    a.fvStart + (int64_t)slot * (a.cap + 1), a.fvIdx + o, a.counts + 2 * slot,
                                    (uint64_t*)((uint8_t*)a.gkeys + (int64_t)slot * a.P * 12));
}

}  // namespace orbv
