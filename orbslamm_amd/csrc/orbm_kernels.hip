// orbm_kernels.hip -- gfx950 kernels of the ORBmatcher hot path (Hamming search).
//
// Reference: /root/reference/SingleRobotScenario/src/ORBmatcher.cc
//   DescriptorDistance :1649-1665, acceptance/ratio :230-232, rotation histogram
//   :238-248, ComputeThreeMaxima :1603-1644, pruning :269-287.
// 256-bit descriptors are 8 dwords; distance = 8 x (v_xor + v_bcnt) in the windowed searches (one lane owns one
// query and keeps (best, second, index); the train descriptor is wave-uniform and comes in through scalar loads).
// The all-pairs scan of the stream matcher runs on the matrix cores instead (k_expand_desc + k_match_mfma): with
// +-32 bytes a . b = 1024 (256 - 2 * Hamming) exactly.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbm {

constexpr int kHistoLength = 30;  // ORBmatcher.cc:39

struct MatchIO {
    const uint8_t* desc;     // slot s at desc + s*descPitch, rows of 32 bytes
    int64_t descPitch;
    const float* ang;        // angle of feature i in slot s: ang[s*angPitch + i*angStride]
    int64_t angPitch;
    int32_t angStride;
    const int32_t* count;    // features per slot
};

__device__ __forceinline__ int hamming256(const uint32_t q[8], const uint32_t* __restrict__ t)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d += __popc(q[i] ^ t[i]);
    return d;
}

// ORBmatcher.cc:238-245: factor = 1.0f/HISTO_LENGTH (sic), round() half away from zero
__device__ __forceinline__ int rot_bin(float aq, float at)
{
    constexpr float factor = 1.0f / kHistoLength;
    float rot = __fsub_rn(aq, at);
    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
    int bin = (int)roundf(__fmul_rn(rot, factor));
    if (bin == kHistoLength) bin = 0;
    return bin;
}

// ComputeThreeMaxima (ORBmatcher.cc:1603-1644) over the 30 rotation bins in sh[0..31], by the workgroup's FIRST WAVE (call it
// with tid < 64, all 64 lanes): the bins come in with one LDS trip, the walk -- strict >, ties keep the earlier bin, second /
// third dropped below a tenth of the first -- runs on wave-uniform values (v_readlane).  One thread reading the bins one
// after the other cost thirty dependent LDS round trips (3 us) with the whole workgroup waiting at the barrier behind it.
__device__ __forceinline__ void three_maxima_wave(const int* sh, int* sInd)
{
    const int hl = sh[threadIdx.x & 31];
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
#pragma unroll
    for (int i = 0; i < kHistoLength; i++) {  // :1609-1633
        const int s = __builtin_amdgcn_readlane(hl, i);
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
    else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
    if ((threadIdx.x & 63) == 0) { sInd[0] = ind1; sInd[1] = ind2; sInd[2] = ind3; }
}

// Brute-force best/second over a CHUNK of the train descriptors.
// grid (queryBlocks, frames, chunks): chunking multiplies the wave count (2000 queries are
// only 32 waves per frame) and the scalar loads of U train descriptors are issued one
// iteration ahead of their use (software pipeline), so the loop is VALU-bound.
// partial[(f*nchunks + chunk)*pitch + q] = {k1, k2}: k1 = best<<20 | trainIdx, k2 = second<<20 | 0xFFFFF.
constexpr int kMatchUnroll = 4;

__device__ __forceinline__ void best2_update(int d, int j, int& best1, int& best2, int& bestIdx)
{
    if (d < best1) { best2 = best1; best1 = d; bestIdx = j; }
    else if (d < best2) best2 = d;
}

__global__ __launch_bounds__(256) void k_match_best2(MatchIO q, MatchIO t, int qslot0, int tslot0, int nchunks,
                                                    uint2* __restrict__ partial, int64_t pitch)
{
    const int f = blockIdx.y, chunk = blockIdx.z;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int nq = q.count[qslot0 + f], nt = t.count[tslot0 + f];
    if (blockIdx.x * 256 >= nq) return;
    const uint8_t* qd = q.desc + (int64_t)(qslot0 + f) * q.descPitch;
    const uint32_t* __restrict__ td = (const uint32_t*)(t.desc + (int64_t)(tslot0 + f) * t.descPitch);
    uint32_t qw[8];
    {
        const int qq = qi < nq ? qi : nq - 1;
        const uint4 a = ((const uint4*)(qd + (int64_t)qq * 32))[0];
        const uint4 b = ((const uint4*)(qd + (int64_t)qq * 32))[1];
        qw[0] = a.x; qw[1] = a.y; qw[2] = a.z; qw[3] = a.w;
        qw[4] = b.x; qw[5] = b.y; qw[6] = b.z; qw[7] = b.w;
    }
    constexpr int U = kMatchUnroll;
    int chunkLen = (nt + nchunks - 1) / nchunks;
    chunkLen = (chunkLen + U - 1) / U * U;
    const int j0 = chunk * chunkLen;
    const int j1 = min(nt, j0 + chunkLen);
    int best1 = 256, best2 = 256, bestIdx = -1;
    int j = j0;
    if (j + U <= j1) {
        uint32_t cur[U][8], nxt[U][8];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) cur[u][i] = td[8 * (j + u) + i];
        for (; j + 2 * U <= j1; j += U) {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int i = 0; i < 8; i++) nxt[u][i] = td[8 * (j + U + u) + i];
#pragma unroll
            for (int u = 0; u < U; u++) best2_update(hamming256(qw, cur[u]), j + u, best1, best2, bestIdx);
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int i = 0; i < 8; i++) cur[u][i] = nxt[u][i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) best2_update(hamming256(qw, cur[u]), j + u, best1, best2, bestIdx);
        j += U;
    }
    for (; j < j1; j++) best2_update(hamming256(qw, td + 8 * j), j, best1, best2, bestIdx);
    if (qi >= nq) return;
    uint2 r;
    r.x = bestIdx < 0 ? 0xFFFFFFFFu : (((uint32_t)best1 << 20) | (uint32_t)bestIdx);
    r.y = ((uint32_t)best2 << 20) | 0xFFFFFu;
    partial[((int64_t)f * nchunks + chunk) * pitch + qi] = r;
}

// acceptance rule (ORBmatcher.cc:230-232) + rotation histogram (:238-248) for one query, given its merged
// keys k1 = best << 20 | index (0xFFFFFFFF: none), k2 = second << 20 | ...
struct AcceptArgs {
    MatchIO q, t;
    int qslot0, tslot0;
    float nnratio;
    int thLow, checkOri;
    int32_t* match;
    int64_t matchPitch;
    uint8_t* binOf;
    int32_t* hist;
};
// returns the accepted train index or -1; `bin` = rotation bin of an accepted match when checkOri
__device__ __forceinline__ int accept_decide(const AcceptArgs& a, int f, int qi, uint32_t k1, uint32_t k2, int& bin)
{
    int m = -1;
    bin = -1;
    if (k1 != 0xFFFFFFFFu) {
        const int best1 = (int)(k1 >> 20), bestIdx = (int)(k1 & 0xFFFFFu);
        const int best2 = k2 == 0xFFFFFFFFu ? 256 : min(256, (int)(k2 >> 20));
        if (best1 <= a.thLow && (float)best1 < __fmul_rn(a.nnratio, (float)best2)) {
            m = bestIdx;
            if (a.checkOri) {
                const float aq = a.q.ang[(int64_t)(a.qslot0 + f) * a.q.angPitch + (int64_t)qi * a.q.angStride];
                const float at = a.t.ang[(int64_t)(a.tslot0 + f) * a.t.angPitch + (int64_t)bestIdx * a.t.angStride];
                bin = rot_bin(aq, at);
            }
        }
    }
    return m;
}
__device__ __forceinline__ void accept_one(const AcceptArgs& a, int f, int qi, uint32_t k1, uint32_t k2)
{
    int bin;
    const int m = accept_decide(a, f, qi, k1, k2, bin);
    if (bin >= 0) {
        a.binOf[(int64_t)f * a.matchPitch + qi] = (uint8_t)bin;
        atomicAdd(&a.hist[f * 32 + bin], 1);
    }
    a.match[(int64_t)f * a.matchPitch + qi] = m;
}

// ------------------------------------------------------------------ brute-force scan on the matrix cores
// The scan is a binary GEMM: with descriptor bits expanded to +-1, a . b = 256 - 2 * Hamming(a, b).  +-1 are exact in
// the 4-bit E2M1 format (0x2 / 0xA) of v_mfma_scale_f32_32x32x64_f8f6f4, the two block scales 2^5 make every product
// +-1024, and sums of at most 256 of them plus a preset below 2^20 are integers the f32 accumulator holds exactly
// (tools/ubench/mfma_fp4.hip checks the instruction against the host).  Against the int8 form (v_mfma_i32_32x32x32_i8 on
// +-32 bytes, the first half of round 3) a descriptor is 128 bytes instead of 256 and an instruction of the same 19 ns
// covers 64 elements instead of 32: half the products' time, half the tile bytes in LDS and from L2.  The popcount
// formulation above is bound by the v_bcnt issue rate (tools/ubench/valu_rate.hip).
//
// Expanded layout of one slot: block b (32 features) at b*4096 bytes; inside a block the 32-element chunk c of
// feature r sits at (c*32 + r)*16, i.e. MFMA step s (64 elements) is the contiguous KiB [s*1024, (s+1)*1024)
// with lane l = (c&1)*32 + r owning 16 bytes -- one coalesced 16-byte load per lane and step.  A and B fragments
// are read the same way, so the element order inside a step cancels out.
constexpr int kMfmaDescBytes = 128;  // one expanded descriptor
__global__ __launch_bounds__(256) void k_expand_desc(MatchIO io, int slot0, int xslot0, uint8_t* __restrict__ xdesc, int64_t xPitch)
{
    const int slot = slot0 + blockIdx.y;
    const int n = io.count[slot];
    const int t = blockIdx.x * 256 + threadIdx.x;  // (block, chunk, row)
    const int blk = t >> 8, c = (t >> 5) & 7, r = t & 31;
    if (blk * 32 >= n) return;
    const int kp = blk * 32 + r;
    uint4 o = make_uint4(0, 0, 0, 0);
    if (kp < n) {
        const uint32_t bits = *(const uint32_t*)(io.desc + (int64_t)slot * io.descPitch + (int64_t)kp * 32 + 4 * c);
        auto pm1 = [](uint32_t b8) {  // 8 bits -> 8 nibbles, set = +1 (0x2), clear = -1 (0xA)
            uint32_t x = (b8 | (b8 << 12)) & 0x000F000Fu;
            x = (x | (x << 6)) & 0x03030303u;
            x = (x | (x << 3)) & 0x11111111u;       // bit i at 4 i
            return 0x22222222u | ((x ^ 0x11111111u) << 3);
        };
        o.x = pm1(bits & 255); o.y = pm1((bits >> 8) & 255); o.z = pm1((bits >> 16) & 255); o.w = pm1(bits >> 24);
    }
    ((uint4*)(xdesc + (int64_t)(xslot0 + blockIdx.y) * xPitch))[t] = o;
}

// median of three in the form the backend selects as v_med3_i32 (IntMed3Pat)
__device__ __forceinline__ int med3i(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int kMfmaWaves = 4;                          // waves of a k_match_mfma workgroup, two query blocks of 32 each
constexpr int kMfmaThreads = 64 * kMfmaWaves;
constexpr int kMfmaRowsPerBlock = 64 * kMfmaWaves;

// One workgroup = 256 queries of one frame pair; the train side streams through LDS in tiles of 32 features (4 KiB,
// a ring filled by LDS-direct loads, one barrier per two tiles); each wave holds two query blocks in registers (32 VGPRs).
//
// The product is taken with the TRAIN tile as the A side and the queries as the B side: accumulator element r of lane l
// is then train row (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the tile against query l & 31 -- every lane's sixteen
// elements belong to ONE query and arrive in ascending train index, tile after tile.  So the running (best, second) of
// a query are two registers of its lane (and of lane + 32, merged once at the end) instead of two per accumulator
// element, and the key needs no instruction at all: the queries are stored negated, every product is -+1024, and the
// accumulators start from the constants 2^19 + r, r = 0..15:
//     acc[r] = 2^19 + 1024 (2 H - 256) + r,
// a positive float holding an integer below 2^20 -- floats of one sign order like their bit patterns, so the fold
// compares those as integers.  Before a tile is folded 16 is subtracted from both running keys, so a key of d tiles ago
// carries -16 d + r in its low field: min = smallest H, then the EARLIEST tile, then the lowest row -- the reference's
// scan with its strict < (ORBmatcher.cc:214-226 form).  -16 d + r > -1024 holds for 64 tiles; every 62 tiles the keys
// are decoded to H << 16 | j and merged into absolute ones.  Per pair: v_min_i32 + v_med3_i32 (med3(best, key, second) =
// the new second), nothing else.  The fold of tile t is written beside the products of tile t + 1 (two accumulator
// sets): the VALU work of a wave sits in the shadow of its own MFMAs.
// Few frames (the one-frame-per-call entry): gridDim.y > 1 cuts the train side into chunks of whole tiles, one
// workgroup each, which leave their (k1, k2) keys in `partial` for k_match_accept to merge -- 9 workgroups walking
// 63 tiles each become 72 walking 8.
constexpr int kMfmaEmpty = 0x7FFFFFFF;                 // absolute keys (H << 16 | j)
constexpr float kMfmaEmptyRelF = 3.402823466e+38f;     // running keys: FLT_MAX (stays FLT_MAX under "- 16")
constexpr int kMfmaTileBytes = 32 * kMfmaDescBytes;    // 4 KiB: one global_load_lds_dwordx4 per thread
#ifndef ORBM_MATCH_PRIO
#define ORBM_MATCH_PRIO 3
#endif
constexpr int kMfmaRing = 16;                          // train tiles in LDS
constexpr int kMfmaLdsBytes = kMfmaRing * kMfmaTileBytes;
constexpr int kMfmaGroup = 4;               // tiles per barrier (even)
constexpr int kMfmaAhead = kMfmaRing - kMfmaGroup; // a group's loads are issued this many tiles ahead of its first tile
static_assert((kMfmaRing & (kMfmaRing - 1)) == 0 && kMfmaAhead - kMfmaGroup <= 63, "ring slot by mask; vmcnt has six bits");
__device__ __forceinline__ int mfma_row_of(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
// running key (bits of the float 2^19 + 1024 (2 H - 256) - 16 d + r, d = tiles before `tileNow`) -> H << 16 | j; anything
// that is not a product (the initial value, a masked row) stays "empty"
__device__ __forceinline__ int mfma_key_abs(int bits, int tileNow, int half)
{
    const float f = __int_as_float(bits);
    if (!(f < 2097152.f)) return kMfmaEmpty;
    const int v = (int)f - (1 << 19) + 1008, low = v & 1023, r = low & 15;
    const int H = ((v >> 10) + 256) >> 1;
    const int j = (tileNow - (63 - (low >> 4))) * 32 + mfma_row_of(r, half);
    return (H << 16) | j;
}
__global__ __launch_bounds__(kMfmaThreads, 2) void k_match_mfma(const uint8_t* __restrict__ xdesc, int64_t xPitch,
                                                      AcceptArgs acc, int nqb, int nframes,
                                                      uint2* __restrict__ partial, int64_t partialPitch)
{
#if ORBM_MATCH_PRIO
    __builtin_amdgcn_s_setprio(ORBM_MATCH_PRIO);
#endif
    const int32_t* __restrict__ count = acc.q.count;  // q.count and t.count index the same slot table here
    const int qslot0 = acc.qslot0, tslot0 = acc.tslot0;
    extern __shared__ uint4 tileB[];  // [kMfmaRing][256]
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs, so the query blocks of one frame
    // pair are given to one XCD and share that frame's train tiles in its L2 -- and (round 5) an XCD takes a RUN of
    // consecutive pairs, whose slots overlap: 41.9 -> 25.0 MB of HBM traffic per 64-pair step, 0.054 -> 0.050 ms alone
    // (profiles/r05_match_xcd_ab.txt; 18.4 MB of +-1 descriptors + the keypoint angles and the tables is the floor)
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    // an XCD takes a RUN of consecutive frame pairs: pair f reads slots f (train) and f + 1 (queries), pair f + 1 slots
    // f + 1 and f + 2 -- with pairs dealt round-robin every slot is fetched into two L2s
    const int f = xcd * ((nframes + 7) >> 3) + k / nqb;
    if (f >= nframes) return;
    const int nq = count[qslot0 + f], nt = count[tslot0 + f];
    const int q0 = (k % nqb) * kMfmaRowsPerBlock;
    if (q0 >= nq) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;
    const int nchunks = gridDim.y, chunk = blockIdx.y;
    const int tilesPer = (((nt + 31) >> 5) + nchunks - 1) / nchunks;
    const int tile0 = chunk * tilesPer;                                  // this workgroup's first train tile
    const int ntiles = max(0, min((nt + 31) >> 5, tile0 + tilesPer) - tile0);
    if (nchunks > 1 && ntiles == 0) {  // an empty chunk still owes its partials
        const int qi = q0 + tid;
        if (qi < nq) partial[((int64_t)f * nchunks + chunk) * partialPitch + qi] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        return;
    }
    constexpr int kTileItems = kMfmaTileBytes / 16;  // 16-byte items of a tile: one per thread
    const uint8_t* qx = xdesc + (int64_t)(qslot0 + f) * xPitch;
    const uint4* tsrc = (const uint4*)(xdesc + (int64_t)(tslot0 + f) * xPitch) + (int64_t)tile0 * kTileItems;
    const int qblk0 = (q0 >> 5) + wave * 2;

    // train tiles go from memory straight into the LDS ring (global_load_lds_dwordx4: the wave's 64 lanes fill one
    // contiguous KiB at M0; wave w owns KiB w of a tile), kMfmaAhead tiles ahead: a tile's first reader in an XCD
    // waits for HBM, and a few tiles of distance do not cover that (int8 form, 0.65 us per tile: four ahead 0.094 ms,
    // six ahead 0.081).  The loads and their counted waits are inline asm -- hipcc would drain the queue at every
    // barrier.  Every step issues exactly one load per tile (past the end: the last tile again), so
    // "vmcnt(kMfmaAhead - kMfmaGroup)" always means "the next two tiles have landed".
    const uint32_t ldsWave = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)&tileB[wave * 64]);
    auto issue = [&](int tile) {
        const uint4* p = tsrc + (int64_t)min(tile, ntiles - 1) * kTileItems + tid;
        const uint32_t dst = ldsWave + (uint32_t)(tile & (kMfmaRing - 1)) * (uint32_t)kMfmaTileBytes;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
    };
    // the ring's first fills go out BEFORE the query fragments are fetched: both wait for memory, and behind each other
    // they cost the workgroup two round trips before its first product (all 512 workgroups of a step start together)
    if (ntiles > 0) {
#pragma unroll
        for (int t = 0; t < kMfmaAhead; t++) issue(t);
    }
    v4i Q[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; qb++)
#pragma unroll
        for (int s = 0; s < 4; s++) Q[qb][s] = *(const v4i*)(qx + (int64_t)(qblk0 + qb) * kMfmaTileBytes + s * 1024 + lane * 16);

    // (tied to the last key of the step so that the wait stays behind the step's arithmetic)
    auto landed = [&](float& after) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(after) : "n"(kMfmaAhead - kMfmaGroup) : "memory"); };
    // queries negated (E2M1 sign bits: x ^ 0x88888888): the accumulator counts -(a . b); using the fragments here also
    // retires their loads before the asm loads start counting
#pragma unroll
    for (int qb = 0; qb < 2; qb++)
#pragma unroll
        for (int s = 0; s < 4; s++) {
            Q[qb][s] ^= (int)0x88888888;
            asm volatile("" : "+v"(Q[qb][s]));  // pinned here: sunk below the tile loads, hipcc's own wait for Q would drain them
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    constexpr float kBias = 524288.f;  // 2^19: every key a positive float
    v16f rowIdx = {kBias, kBias + 1, kBias + 2, kBias + 3, kBias + 4, kBias + 5, kBias + 6, kBias + 7,
                   kBias + 8, kBias + 9, kBias + 10, kBias + 11, kBias + 12, kBias + 13, kBias + 14, kBias + 15};
    // sixteen registers that stay: as constants hipcc rebuilds them in front of every tile's first products (8 v_mov_b64 per
    // tile and wave, two passes each), as registers they are just the C operand of a chain's first MFMA
    asm volatile("" : "+v"(rowIdx));
    // running keys, relative to the tile folded last -- kept and compared AS FLOATS: every key is a positive normal float (no NaN, no
    // denormal), so the float order is the integer order of the bits, and on floats hipcc forms v_min3_f32 and takes v_med3_f32
    // from a builtin -- on the bit patterns it shares min(best, k) between its med3 pattern and the best's update and loses the v_min3
    float b0 = kMfmaEmptyRelF, s0 = kMfmaEmptyRelF, b1 = kMfmaEmptyRelF, s1 = kMfmaEmptyRelF;
    int B0 = kMfmaEmpty, S0 = kMfmaEmpty, B1 = kMfmaEmpty, S1 = kMfmaEmpty;              // H << 16 | j, of the epochs flushed so far
    int nfold = 0;
    auto mfma = [](const v4i& a, const v4i& b, const v16f& c) {  // fp4 x fp4 (cbsz = blgp = 4), block scales 2^5 each
        const v8i a8 = {a.x, a.y, a.z, a.w, 0, 0, 0, 0}, b8 = {b.x, b.y, b.z, b.w, 0, 0, 0, 0};
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, c, 4, 4, 0, 132, 0, 132);
    };
    // a tile's fragments are read from the ring one tile AHEAD of their products (two register sets): read at its point of
    // use, the LDS latency of every tile stood in front of its first MFMA -- a quarter of the kernel (round 5 ablation)
    struct Tile { v4i s[4]; };
    auto read_tile = [&](int slot, Tile& T) {
        const v4i* bt = (const v4i*)(tileB + slot * kTileItems);
#pragma unroll
        for (int s = 0; s < 4; s++) T.s[s] = bt[s * 64 + lane];
    };
    auto products = [&](const Tile& T, v16f& a0, v16f& a1) {
        a0 = mfma(T.s[0], Q[0][0], rowIdx);
        a1 = mfma(T.s[0], Q[1][0], rowIdx);
#pragma unroll
        for (int s = 1; s < 4; s++) {
            a0 = mfma(T.s[s], Q[0][s], a0);
            a1 = mfma(T.s[s], Q[1][s], a1);
        }
    };
    auto flush = [&]() {  // relative keys of this epoch -> absolute, merged behind the earlier epochs (which win ties)
        const int now = tile0 + nfold - 1;
        const int cb0 = mfma_key_abs(__float_as_int(b0), now, half), cs0 = mfma_key_abs(__float_as_int(s0), now, half);
        const int cb1 = mfma_key_abs(__float_as_int(b1), now, half), cs1 = mfma_key_abs(__float_as_int(s1), now, half);
        S0 = min(max(B0, cb0), min(S0, cs0)); B0 = min(B0, cb0);
        S1 = min(max(B1, cb1), min(S1, cs1)); B1 = min(B1, cb1);
        b0 = s0 = b1 = s1 = kMfmaEmptyRelF;
    };
    auto older = [](float key) { return key - 16.f; };
    auto fold = [&](const v16f& a0, const v16f& a1) {
        // compiler-visible VALU ops on the accumulator: hipcc pads the MFMA -> VALU read hazard itself
        // (an inline-asm consumer would read the accumulator too early)
        b0 = older(b0); s0 = older(s0); b1 = older(b1); s1 = older(s1);
        // two keys per step: the new second is min(second, median(best, k, k')) -- the runner-up of {best, k, k'} is their
        // median, and the old second only has to beat that -- the new best min3(best, k, k'): v_med3 + v_min3 per two keys and
        // one v_min3 per four for the seconds, 2.5 instructions per two keys where min + med3 per key took four (round 5)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float k0 = a0[r], k0n = a0[r + 1], k1 = a1[r], k1n = a1[r + 1];
            s0 = __builtin_fminf(s0, __builtin_amdgcn_fmed3f(b0, k0, k0n));
            b0 = __builtin_fminf(__builtin_fminf(b0, k0), k0n);
            s1 = __builtin_fminf(s1, __builtin_amdgcn_fmed3f(b1, k1, k1n));
            b1 = __builtin_fminf(__builtin_fminf(b1, k1), k1n);
        }
        ++nfold;  // (the epoch's flush is the caller's: a branch here parts the fold from the products it should run beside)
    };
    auto fold_last = [&](const v16f& a0, const v16f& a1) {  // the last tile may hold rows past the end of the frame
        b0 = older(b0); s0 = older(s0); b1 = older(b1); s1 = older(s1);
        const int jrow = (tile0 + ntiles - 1) * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const bool in = jrow + (r & 3) + 8 * (r >> 2) < nt, inn = jrow + ((r + 1) & 3) + 8 * ((r + 1) >> 2) < nt;
            const float k0 = in ? a0[r] : kMfmaEmptyRelF, k0n = inn ? a0[r + 1] : kMfmaEmptyRelF;
            const float k1 = in ? a1[r] : kMfmaEmptyRelF, k1n = inn ? a1[r + 1] : kMfmaEmptyRelF;
            s0 = __builtin_fminf(s0, __builtin_amdgcn_fmed3f(b0, k0, k0n));
            b0 = __builtin_fminf(__builtin_fminf(b0, k0), k0n);
            s1 = __builtin_fminf(s1, __builtin_amdgcn_fmed3f(b1, k1, k1n));
            b1 = __builtin_fminf(__builtin_fminf(b1, k1), k1n);
        }
        nfold++;
        flush();
    };

    if (ntiles > 0) {
        v16f accE0, accE1, accO0, accO1;
        landed(s1);
        __syncthreads();
        // two tiles per barrier: tile a's products into one accumulator set while the other (tile a - 1) is folded,
        // then the same with the sets exchanged
        // (a whole group is one basic block: the tile reads of the group's later tiles can rise above the folds before them)
        auto group_step = [&](int a, auto first, auto whole) {
#pragma unroll
            for (int i = 0; i < kMfmaGroup; i++) issue(a + kMfmaAhead + i);
            Tile TE, TO;
            read_tile(a & (kMfmaRing - 1), TE);
#pragma unroll
            for (int i = 0; i < kMfmaGroup; i++) {
                if (!decltype(whole)::value && i > 0 && a + i >= ntiles) break;
                // (a slot past the frame's last tile holds that tile again: read, never multiplied)
                if (i + 1 < kMfmaGroup) read_tile((a + i + 1) & (kMfmaRing - 1), (i & 1) ? TE : TO);
                __builtin_amdgcn_sched_barrier(0);  // the reads stay up here
                if (i & 1) { products(TO, accO0, accO1); fold(accE0, accE1); }
                else {
                    products(TE, accE0, accE1);
                    if (i > 0 || !decltype(first)::value) fold(accO0, accO1);
                }
            }
            landed(s1);
            __syncthreads();
        };
        if (kMfmaGroup <= ntiles) group_step(0, std::true_type(), std::true_type());
        else group_step(0, std::true_type(), std::false_type());
        // epochs of whole groups, 64 - kMfmaGroup tiles at most (the first holds the first group's folds as well, the last fold_last's): 64 folds at most between flushes
        constexpr int kEpoch = (64 - kMfmaGroup) / kMfmaGroup * kMfmaGroup;
        for (int e0 = kMfmaGroup; e0 < ntiles; e0 += kEpoch) {
            const int e1 = min(ntiles, e0 + kEpoch);
            int a = e0;
            for (; a + kMfmaGroup <= e1; a += kMfmaGroup) group_step(a, std::false_type(), std::true_type());
            if (a < e1) group_step(a, std::false_type(), std::false_type());
            flush();
        }
        if (ntiles & 1) fold_last(accE0, accE1); else fold_last(accO0, accO1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // the two halves of the wave hold the two halves of every tile's rows: merge lane l with lane l ^ 32; then the
    // lower half-wave takes the rows of query block 0, the upper that of block 1, and the acceptance rule runs on all
    // 64 lanes
    {
        const int oB0 = __shfl_xor(B0, 32), oS0 = __shfl_xor(S0, 32), oB1 = __shfl_xor(B1, 32), oS1 = __shfl_xor(S1, 32);
        const int mS0 = min(max(B0, oB0), min(S0, oS0)), mB0 = min(B0, oB0);
        const int mS1 = min(max(B1, oB1), min(S1, oS1)), mB1 = min(B1, oB1);
        const int myB = half ? mB1 : mB0, myS = half ? mS1 : mS0;
        const int qi = q0 + wave * 64 + lane;  // = block (lane >> 5), row lane & 31
        const uint32_t h1 = (uint32_t)myB >> 16, h2 = (uint32_t)myS >> 16;  // empty: 0x7FFF
        const uint32_t k1 = h1 >= 256u ? 0xFFFFFFFFu : ((h1 << 20) | ((uint32_t)myB & 0xFFFFu));
        const uint32_t k2 = ((h2 >= 256u ? 256u : h2) << 20) | 0xFFFFFu;
        if (nchunks > 1) {
            if (qi < nq) partial[((int64_t)f * nchunks + chunk) * partialPitch + qi] = make_uint2(k1, k2);
            return;
        }
        // the rotation histogram of the workgroup's 256 queries is counted in LDS (the ring is free: every wave's tile loads
        // have landed) and leaves as at most 30 global atomics: one atomic per accepted match was 2.4 of the kernel's 2.6 MB
        // of writes per step
        int* const sh = (int*)tileB;
        __syncthreads();
        if (tid < 32) sh[tid] = 0;
        __syncthreads();
        if (qi < nq) {
            int bin;
            const int m = accept_decide(acc, f, qi, k1, k2, bin);
            if (bin >= 0) {
                acc.binOf[(int64_t)f * acc.matchPitch + qi] = (uint8_t)bin;
                atomicAdd(&sh[bin], 1);
            }
            acc.match[(int64_t)f * acc.matchPitch + qi] = m;
        }
        __syncthreads();
        if (tid < 32 && sh[tid]) atomicAdd(&acc.hist[f * 32 + tid], sh[tid]);
    }
}

// merge the chunk partials in index order, apply the acceptance rule (ORBmatcher.cc:230-232)
// and histogram the rotation bin (:238-248)
__global__ __launch_bounds__(256) void k_match_accept(AcceptArgs a, int nchunks, const uint2* __restrict__ partial, int64_t pitch)
{
    const int f = blockIdx.y;
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int nq = a.q.count[a.qslot0 + f];
    if (qi >= nq) return;
    uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
    for (int c = 0; c < nchunks; c++) {
        const uint2 p = partial[((int64_t)f * nchunks + c) * pitch + qi];
        const uint32_t lo = min(k1, p.x), hi = max(k1, p.x);
        k2 = min(hi, min(k2, p.y));
        k1 = lo;
    }
    accept_one(a, f, qi, k1, k2);
}

// ComputeThreeMaxima + pruning, one workgroup per frame; leaves hist zeroed for the next call
__global__ __launch_bounds__(256) void k_match_prune(MatchIO q, int qslot0, int checkOri,
                                                    int32_t* __restrict__ match, int64_t matchPitch,
                                                    const uint8_t* __restrict__ binOf,
                                                    int32_t* __restrict__ hist, int32_t* __restrict__ nmatch)
{
    __shared__ int sh[32];
    __shared__ int sInd[3];
    __shared__ int sCnt;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int nq = q.count[qslot0 + f];
    if (tid < 32) { sh[tid] = hist[f * 32 + tid]; hist[f * 32 + tid] = 0; }
    if (tid == 0) sCnt = 0;
    __syncthreads();
    if (tid < 64) three_maxima_wave(sh, sInd);
    __syncthreads();
    int local = 0;
    for (int i = tid; i < nq; i += 256) {
        int m = match[(int64_t)f * matchPitch + i];
        if (m >= 0 && checkOri) {
            const int bin = binOf[(int64_t)f * matchPitch + i];
            if (bin != sInd[0] && bin != sInd[1] && bin != sInd[2]) {
                m = -1;
                match[(int64_t)f * matchPitch + i] = -1;
            }
        }
        local += m >= 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if ((tid & 63) == 0) atomicAdd(&sCnt, local);
    __syncthreads();
    if (tid == 0) nmatch[f] = sCnt;
}

// k_match_accept + k_match_prune in one workgroup per frame, for the few-frame (latency) mode: merge of the chunk
// partials, acceptance rule, rotation histogram in LDS, three maxima, pruning, count -- one launch instead of two and
// no global histogram.
__global__ __launch_bounds__(1024) void k_match_accept_prune(AcceptArgs a, int nchunks, const uint2* __restrict__ partial,
                                                            int64_t pitch, int32_t* __restrict__ nmatch)
{
    __shared__ int sh[32];
    __shared__ int sInd[3];
    __shared__ int sCnt;
    const int f = blockIdx.x, tid = threadIdx.x;
    const int nq = a.q.count[a.qslot0 + f];
    if (tid < 32) sh[tid] = 0;
    if (tid == 0) sCnt = 0;
    __syncthreads();
    for (int qi = tid; qi < nq; qi += 1024) {
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        for (int c = 0; c < nchunks; c++) {
            const uint2 p = partial[((int64_t)f * nchunks + c) * pitch + qi];
            const uint32_t lo = min(k1, p.x), hi = max(k1, p.x);
            k2 = min(hi, min(k2, p.y));
            k1 = lo;
        }
        int bin;
        const int m = accept_decide(a, f, qi, k1, k2, bin);
        a.match[(int64_t)f * a.matchPitch + qi] = m;
        if (bin >= 0) { a.binOf[(int64_t)f * a.matchPitch + qi] = (uint8_t)bin; atomicAdd(&sh[bin], 1); }
    }
    __syncthreads();
    if (tid < 64) three_maxima_wave(sh, sInd);
    __syncthreads();
    int local = 0;
    for (int qi = tid; qi < nq; qi += 1024) {  // the thread re-reads what it wrote above
        int m = a.match[(int64_t)f * a.matchPitch + qi];
        if (m >= 0 && a.checkOri) {
            const int bin = a.binOf[(int64_t)f * a.matchPitch + qi];
            if (bin != sInd[0] && bin != sInd[1] && bin != sInd[2]) {
                m = -1;
                a.match[(int64_t)f * a.matchPitch + qi] = -1;
            }
        }
        local += m >= 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if ((tid & 63) == 0) atomicAdd(&sCnt, local);
    __syncthreads();
    if (tid == 0) nmatch[f] = sCnt;
}

// full distance matrix (DescriptorDistance for every pair)
__global__ __launch_bounds__(256) void k_distance_matrix(const uint8_t* __restrict__ qd, int nq,
                                                        const uint8_t* __restrict__ td, int nt,
                                                        int32_t* __restrict__ out)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= nq) return;
    const int qq = qi < nq ? qi : nq - 1;
    uint32_t qw[8];
    const uint32_t* qp = (const uint32_t*)(qd + (int64_t)qq * 32);
#pragma unroll
    for (int i = 0; i < 8; i++) qw[i] = qp[i];
    const uint32_t* tp = (const uint32_t*)td;
    for (int j = blockIdx.y; j < nt; j += gridDim.y) {
        const int d = hamming256(qw, tp + 8 * j);
        if (qi < nq) out[(int64_t)qi * nt + j] = d;
    }
}


// ------------------------------------------------------------------ SearchByBoW (greedy inside a vocabulary node)
// A train feature sits in exactly one node of its FeatureVector, so matched node
// pairs are independent; inside a pair the reference walks the query list in order
// and skips train features matched earlier in the call (ORBmatcher.cc:205-211,
// :574-579).  One wave per node pair: queries sequential, candidates lane-parallel,
// top-2 by (distance, scan position) so ties resolve like the serial scan.
struct BowArgs {
    const uint8_t* qdesc; const float* qang; const uint8_t* qvalid;
    const uint8_t* tdesc; const float* tang; const uint8_t* tvalid;
    const int32_t* qstart; const int32_t* qidx;   // CSR of the query feature vector
    const int32_t* tstart; const int32_t* tidx;   // CSR of the train feature vector
    const int32_t* pairQ; const int32_t* pairT;   // matched node pairs (host lock-step walk :180-266)
    uint8_t* matched;                              // per train feature, zero-initialised
    int32_t* match; uint8_t* binOf; int32_t* hist; // outputs
    float nnratio; int32_t thLow; int32_t checkOri; int32_t outByTrain;
};

__device__ __forceinline__ void top2_merge(uint32_t& k1, uint32_t& k2, uint32_t o1, uint32_t o2)
{
    const uint32_t lo = min(k1, o1), hi = max(k1, o1);
    k2 = min(hi, min(k2, o2));
    k1 = lo;
}

// minimum over the 64 lanes without touching LDS: four DPP steps reduce every row of 16, row_bcast15 / row_bcast31 carry
// the row minima to the last lane (identity 0xFFFFFFFF where a lane has no source)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#define ORBM_DPP_MIN(ctrl, rows) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, ctrl, rows, 0xF, false))
    ORBM_DPP_MIN(0xB1, 0xF);    // quad_perm [1, 0, 3, 2]
    ORBM_DPP_MIN(0x4E, 0xF);    // quad_perm [2, 3, 0, 1]
    ORBM_DPP_MIN(0x141, 0xF);   // row_half_mirror
    ORBM_DPP_MIN(0x140, 0xF);   // row_mirror
    ORBM_DPP_MIN(0x142, 0xA);   // row_bcast15 into rows 1 and 3
    ORBM_DPP_MIN(0x143, 0xC);   // row_bcast31 into rows 2 and 3
#undef ORBM_DPP_MIN
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ void bow_pair_body(const BowArgs& a, int nodeQ, int nodeT)
{
    const int lane = threadIdx.x;
    const int qs = a.qstart[nodeQ], qe = a.qstart[nodeQ + 1];
    const int ts = a.tstart[nodeT], te = a.tstart[nodeT + 1];
    constexpr int kPer = 4;  // features of a node per lane on either side
    if (qe - qs <= 64 * kPer && te - ts <= 64 * kPer) {
        // The common case (a vocabulary node holds a few dozen features of a frame, seldom more than a hundred): lane l
        // keeps train features l, l + 64, .. AND queries l, l + 64, .. of the node in registers -- descriptor, angle,
        // flags -- so everything is loaded once, up front and in parallel, and the sequential walk over the queries
        // (:205-211) is register work: the query's descriptor goes lane -> scalar registers (v_readlane), the two minima
        // through DPP, the "already matched" flag lives with the lane that owns the train feature.
        const int ntj = (te - ts + 63) >> 6, nqj = (qe - qs + 63) >> 6;
        int t[kPer], q[kPer];
        uint32_t tw[kPer][8], qw[kPer][8];
        float tang[kPer], qang[kPer];
        bool tfree[kPer], qok[kPer];
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const bool hasT = j < ntj && lane + 64 * j < te - ts, hasQ = j < nqj && lane + 64 * j < qe - qs;
            t[j] = hasT ? a.tidx[ts + lane + 64 * j] : 0;
            q[j] = hasQ ? a.qidx[qs + lane + 64 * j] : 0;
            tfree[j] = hasT; qok[j] = hasQ;
        }
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            if (j < ntj) {
                const uint32_t* tp = (const uint32_t*)(a.tdesc + (int64_t)t[j] * 32);
#pragma unroll
                for (int i = 0; i < 8; i++) tw[j][i] = tp[i];
                tang[j] = a.checkOri ? a.tang[t[j]] : 0.f;
                tfree[j] = tfree[j] && !a.matched[t[j]] && !(a.tvalid && !a.tvalid[t[j]]);
            }
            if (j < nqj) {
                const uint32_t* qp = (const uint32_t*)(a.qdesc + (int64_t)q[j] * 32);
#pragma unroll
                for (int i = 0; i < 8; i++) qw[j][i] = qp[i];
                qang[j] = a.checkOri ? a.qang[q[j]] : 0.f;
                qok[j] = qok[j] && !(a.qvalid && !a.qvalid[q[j]]);
            }
        }
#pragma unroll
        for (int jq = 0; jq < kPer; jq++) {
            if (jq >= nqj) break;
            const uint64_t qokMask = __builtin_amdgcn_ballot_w64(qok[jq]);
            const int nHere = min(64, qe - qs - 64 * jq);
            for (int iq = 0; iq < nHere; iq++) {
                if (!((qokMask >> iq) & 1)) continue;  // wave-uniform
                uint32_t sq[8];
#pragma unroll
                for (int i = 0; i < 8; i++) sq[i] = (uint32_t)__builtin_amdgcn_readlane((int)qw[jq][i], iq);
                // key = dist << 22 | scan position (lane + 64 j): unique per (lane, j)
                uint32_t key[kPer];
                uint32_t kmin = 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < kPer; j++) {
                    key[j] = 0xFFFFFFFFu;
                    if (j < ntj) {
                        int d = 0;
#pragma unroll
                        for (int i = 0; i < 8; i++) d += __popc(tw[j][i] ^ sq[i]);
                        if (tfree[j]) key[j] = ((uint32_t)d << 22) | (uint32_t)(lane + 64 * j);
                        kmin = min(kmin, key[j]);
                    }
                }
                const uint32_t k1 = wave_min_u32(kmin);
                uint32_t kmin2 = 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < kPer; j++) if (j < ntj && key[j] != k1) kmin2 = min(kmin2, key[j]);
                const uint32_t k2 = wave_min_u32(kmin2);
                const int best1 = k1 == 0xFFFFFFFFu ? 256 : (int)(k1 >> 22);
                const int best2 = k2 == 0xFFFFFFFFu ? 256 : (int)(k2 >> 22);
                if (best1 <= a.thLow && (float)best1 < __fmul_rn(a.nnratio, (float)best2)) {
                    const int win = (int)(k1 & 0x3FFFFFu);
                    const int qi = __builtin_amdgcn_readlane(q[jq], iq);
                    const float qa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qang[jq]), iq));
#pragma unroll
                    for (int j = 0; j < kPer; j++) {
                        if (j < ntj && lane + 64 * j == win) {
                            tfree[j] = false;
                            a.matched[t[j]] = 1;
                            const int o = a.outByTrain ? t[j] : qi;
                            a.match[o] = a.outByTrain ? qi : t[j];
                            if (a.checkOri) a.binOf[o] = (uint8_t)rot_bin(qa, tang[j]);  // the histogram is counted by the pruning kernel, in LDS
                        }
                    }
                }
            }
        }
        return;
    }
    for (int iq = qs; iq < qe; iq++) {
        const int q = a.qidx[iq];
        if (a.qvalid && !a.qvalid[q]) continue;  // wave-uniform
        uint32_t qw[8];
        const uint32_t* qp = (const uint32_t*)(a.qdesc + (int64_t)q * 32);
#pragma unroll
        for (int i = 0; i < 8; i++) qw[i] = qp[i];
        // key = dist << 22 | scan position (dist <= 256, position < 2^22)
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        for (int it = ts + lane; it < te; it += 64) {
            const int t = a.tidx[it];
            if (a.matched[t]) continue;
            if (a.tvalid && !a.tvalid[t]) continue;
            const int d = hamming256(qw, (const uint32_t*)(a.tdesc + (int64_t)t * 32));
            const uint32_t key = ((uint32_t)d << 22) | (uint32_t)(it - ts);
            if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
        }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            const uint32_t o1 = __shfl_xor(k1, dd), o2 = __shfl_xor(k2, dd);
            top2_merge(k1, k2, o1, o2);
        }
        const int best1 = k1 == 0xFFFFFFFFu ? 256 : (int)(k1 >> 22);
        const int best2 = k2 == 0xFFFFFFFFu ? 256 : (int)(k2 >> 22);
        if (best1 <= a.thLow && (float)best1 < __fmul_rn(a.nnratio, (float)best2)) {
            const int t = a.tidx[ts + (int)(k1 & 0x3FFFFFu)];
            if (lane == 0) {
                a.matched[t] = 1;
                const int o = a.outByTrain ? t : q;
                a.match[o] = a.outByTrain ? q : t;
                if (a.checkOri) a.binOf[o] = (uint8_t)rot_bin(a.qang[q], a.tang[t]);
            }
        }
        __syncthreads();  // matched[] visible to the whole wave before the next query
    }
}

__global__ __launch_bounds__(64) void k_bow_pairs(BowArgs a)
{
    bow_pair_body(a, a.pairQ[blockIdx.x], a.pairT[blockIdx.x]);
}

// SearchByBoW(KeyFrame = slot kf[p], Frame = slot fr[p]) for pairs of a frame set's slots, one launch: workgroup (b, p)
// takes vocabulary node b of the KeyFrame's FeatureVector and looks its id up in the Frame's (the lock-step walk of the
// two sorted maps, ORBmatcher.cc:180-266, as a binary search), so no node list visits the host.
constexpr int kBowMaxPairs = 128;
struct BowSetArgs {
    const uint8_t* desc; const float* ang; const int32_t* n; int32_t cap;       // frame set: slot s at + s * cap (desc: * 32)
    const uint32_t* fvNode; const int32_t* fvStart; const int32_t* fvIdx; const int32_t* counts;  // fvStart: + s * (cap + 1), counts: + 2 s + 1 = nodes
    uint8_t* matched; int32_t* match; uint8_t* binOf; int32_t* hist;             // per pair p: + p * cap (hist: + p * 32); zero / -1 / - / zero on entry
    float nnratio; int32_t thLow, checkOri;
    int16_t kf[kBowMaxPairs], fr[kBowMaxPairs];
};

__global__ __launch_bounds__(64) void k_bow_set(BowSetArgs s)
{
    const int p = blockIdx.y, qs = s.kf[p], ts = s.fr[p];
    const int nodeQ = blockIdx.x, lane = threadIdx.x;
    const int64_t C = s.cap;
    // first round trip: both node counts, this workgroup's node id, the KeyFrame node's feature range
    const int nQ = s.counts[2 * qs + 1], nT = s.counts[2 * ts + 1];
    const uint32_t* qn = s.fvNode + qs * C;
    const uint32_t* tn = s.fvNode + ts * C;
    const uint32_t id = qn[min(nodeQ, (int)C - 1)];
    if (nodeQ >= nQ) return;
    // second: the Frame's node with the same id (the lock-step walk of the two sorted maps, :180-266).  Up to 128 nodes are
    // looked at by all lanes at once -- a binary search is seven dependent loads
    int lo = -1;
    if (nT <= 128) {
        const uint32_t t0 = lane < nT ? tn[lane] : 0xFFFFFFFFu, t1 = lane + 64 < nT ? tn[lane + 64] : 0xFFFFFFFFu;
        const uint64_t m0 = __builtin_amdgcn_ballot_w64(lane < nT && t0 == id), m1 = __builtin_amdgcn_ballot_w64(lane + 64 < nT && t1 == id);
        if (m0) lo = __ffsll((long long)m0) - 1; else if (m1) lo = 64 + __ffsll((long long)m1) - 1;
    } else {
        int a0 = 0, hi = nT;   // first train node with id >= ours (DBoW2's lower_bound, :255-263)
        while (a0 < hi) { const int mid = (a0 + hi) >> 1; if (tn[mid] < id) a0 = mid + 1; else hi = mid; }
        if (a0 < nT && tn[a0] == id) lo = a0;
    }
    if (lo < 0) return;
    BowArgs a;
    a.qdesc = s.desc + qs * C * 32; a.qang = s.ang + qs * C; a.qvalid = nullptr;
    a.tdesc = s.desc + ts * C * 32; a.tang = s.ang + ts * C; a.tvalid = nullptr;
    a.qstart = s.fvStart + (int64_t)qs * (C + 1); a.qidx = s.fvIdx + qs * C;
    a.tstart = s.fvStart + (int64_t)ts * (C + 1); a.tidx = s.fvIdx + ts * C;
    a.pairQ = nullptr; a.pairT = nullptr;
    a.matched = s.matched + p * C; a.match = s.match + p * C; a.binOf = s.binOf + p * C; a.hist = s.hist + p * 32;
    a.nnratio = s.nnratio; a.thLow = s.thLow; a.checkOri = s.checkOri; a.outByTrain = 1;
    bow_pair_body(a, nodeQ, lo);
}

// three maxima + pruning of every pair's match table (k_prune_flat's rule), final table and count to `outMatch` / `outN`
// (pinned host memory: every entry is written once)
__global__ __launch_bounds__(256) void k_bow_prune_set(BowSetArgs s, int32_t* __restrict__ outMatch, int32_t* __restrict__ outN)
{
    __shared__ int sh[32];
    __shared__ int sInd[3];
    __shared__ int sCnt;
    const int tid = threadIdx.x, p = blockIdx.x;
    const int64_t C = s.cap;
    const int n = min(s.n[s.fr[p]], s.cap);
    // the rotation histogram (:238-248) is counted here, in LDS: a few hundred matches of a pair fall into two or three
    // bins, and as global atomics on those few words they took longer than the search itself
    if (tid < 32) sh[tid] = 0;
    if (tid == 0) sCnt = 0;
    __syncthreads();
    if (s.checkOri)
        for (int i = tid; i < n; i += 256) if (s.match[p * C + i] >= 0) atomicAdd(&sh[s.binOf[p * C + i]], 1);
    __syncthreads();
    if (tid < 64) three_maxima_wave(sh, sInd);
    __syncthreads();
    int local = 0;
    for (int i = tid; i < n; i += 256) {
        int m = s.match[p * C + i];
        if (m >= 0 && s.checkOri) {
            const int bin = s.binOf[p * C + i];
            if (bin != sInd[0] && bin != sInd[1] && bin != sInd[2]) m = -1;
        }
        outMatch[p * C + i] = m;
        local += m >= 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if ((tid & 63) == 0) atomicAdd(&sCnt, local);
    __syncthreads();
    if (tid == 0) outN[p] = sCnt;
}

// three maxima + pruning over a flat match array (same rule as k_match_prune), one workgroup
__global__ __launch_bounds__(256) void k_prune_flat(int32_t* __restrict__ match, int n, int checkOri,
                                                   const uint8_t* __restrict__ binOf, int32_t* __restrict__ hist,
                                                   int32_t* __restrict__ nmatch)
{
    __shared__ int sh[32];
    __shared__ int sInd[3];
    __shared__ int sCnt;
    const int tid = threadIdx.x;
    if (tid < 32) sh[tid] = 0;   // (`hist` is no longer an input: the bins are counted here, in LDS, from binOf)
    if (tid == 0) sCnt = 0;
    __syncthreads();
    if (checkOri)
        for (int i = tid; i < n; i += 256) if (match[i] >= 0) atomicAdd(&sh[binOf[i]], 1);
    __syncthreads();
    if (tid < 64) three_maxima_wave(sh, sInd);
    __syncthreads();
    int local = 0;
    for (int i = tid; i < n; i += 256) {
        int m = match[i];
        if (m >= 0 && checkOri) {
            const int bin = binOf[i];
            if (bin != sInd[0] && bin != sInd[1] && bin != sInd[2]) { m = -1; match[i] = -1; }
        }
        local += m >= 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if ((tid & 63) == 0) atomicAdd(&sCnt, local);
    __syncthreads();
    if (tid == 0) *nmatch = sCnt;
}

// ------------------------------------------------------------------ Frame grid (Frame.cc:230-245, 327-392)
struct GridDev { float minX, minY, invW, invH; int32_t cols, rows; };
struct KeyDev { float x, y, size, angle, response; int32_t octave, class_id; };

__device__ __forceinline__ bool pos_in_grid(const GridDev& g, float x, float y, int& px, int& py)
{
    px = (int)roundf(__fmul_rn(__fsub_rn(x, g.minX), g.invW));  // Frame.cc:384-385
    py = (int)roundf(__fmul_rn(__fsub_rn(y, g.minY), g.invH));
    return !(px < 0 || px >= g.cols || py < 0 || py >= g.rows);
}

__global__ void k_grid_count(GridDev g, const KeyDev* __restrict__ keys, int n, int32_t* __restrict__ cellCnt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int px, py;
    if (pos_in_grid(g, keys[i].x, keys[i].y, px, py)) atomicAdd(&cellCnt[px * g.rows + py], 1);
}

// exclusive scan of cellCnt (ncell entries) into cellStart (ncell+1), one workgroup of 1024
__global__ __launch_bounds__(1024) void k_scan_small(const int32_t* __restrict__ in, int n, int32_t* __restrict__ out)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? in[i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < 16; w++) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
        if (i < n) out[i] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 0) carry += tot;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry;
}

__global__ void k_grid_fill(GridDev g, const KeyDev* __restrict__ keys, int n, const int32_t* __restrict__ cellStart,
                            int32_t* __restrict__ cellFill, int32_t* __restrict__ cellIdx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int px, py;
    if (pos_in_grid(g, keys[i].x, keys[i].y, px, py)) {
        const int c = px * g.rows + py;
        cellIdx[cellStart[c] + atomicAdd(&cellFill[c], 1)] = i;
    }
}

// restore insertion order (ascending feature index) inside every cell
__global__ void k_grid_sort(int ncell, const int32_t* __restrict__ cellStart, int32_t* __restrict__ cellIdx)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const int s = cellStart[c], e = cellStart[c + 1];
    for (int i = s + 1; i < e; i++) {
        const int v = cellIdx[i];
        int j = i - 1;
        while (j >= s && cellIdx[j] > v) { cellIdx[j + 1] = cellIdx[j]; j--; }
        cellIdx[j + 1] = v;
    }
}

// the features in GRID order as {x, y, index | octave << 24, 0}: what the projection searches walk (orbt::FrameSetDev::rec), for
// frames too large for the one-workgroup build (orbt::k_frame_build keeps the whole grid in LDS)
__global__ void k_grid_records(const KeyDev* __restrict__ keys, const int32_t* __restrict__ cellStart, int ncell,
                               const int32_t* __restrict__ cellIdx, uint4* __restrict__ rec)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cellStart[ncell]) return;
    const int i = cellIdx[j];
    const KeyDev k = keys[i];
    rec[j] = make_uint4(__float_as_uint(k.x), __float_as_uint(k.y), (uint32_t)i | ((uint32_t)k.octave << 24), 0u);
}

// GetFeaturesInArea: calls f(featureIndex) in reference order (ix outer, iy inner, insertion order)
template <class F>
__device__ __forceinline__ void for_each_in_area(const GridDev& g, const KeyDev* __restrict__ keys,
                                                 const int32_t* __restrict__ cellStart, const int32_t* __restrict__ cellIdx,
                                                 float x, float y, float r, int minLevel, int maxLevel, F f)
{
    int nMinCellX = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, g.minX), r), g.invW));
    if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= g.cols) return;
    int nMaxCellX = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, g.minX), r), g.invW));
    if (nMaxCellX > g.cols - 1) nMaxCellX = g.cols - 1;
    if (nMaxCellX < 0) return;
    int nMinCellY = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, g.minY), r), g.invH));
    if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= g.rows) return;
    int nMaxCellY = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, g.minY), r), g.invH));
    if (nMaxCellY > g.rows - 1) nMaxCellY = g.rows - 1;
    if (nMaxCellY < 0) return;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * g.rows + iy;
            for (int j = cellStart[c]; j < cellStart[c + 1]; j++) {
                const int i = cellIdx[j];
                const KeyDev& kp = keys[i];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                const float distx = __fsub_rn(kp.x, x), disty = __fsub_rn(kp.y, y);
                if (fabsf(distx) < r && fabsf(disty) < r) f(i);
            }
        }
}

// ------------------------------------------------------------------ SearchByProjection family
struct ProjArgs {
    GridDev grid;
    const KeyDev* tkeys; const int32_t* cellStart; const int32_t* cellIdx;
    const float* quvr; const int8_t* qlvl; const uint8_t* qdesc; const float* qang;
    const uint8_t* qvalid; const uint8_t* qobs;
    const uint8_t* tdesc;
    const float* qur; const float* turight;   // stereo gate of modes 3/4 (null for mono): |q_ur - mvuRight[t]| <= radius
    int32_t nq, nt;
    int32_t* candCnt;      // [nq]   pass 1
    int32_t* candOff;      // [nq+1] after scan
    uint32_t* candKey;     // CSR: dist<<22 | pos<<4 | octave   (pos = rank in the reference's scan order)
    int32_t* candIdx;      // CSR: train feature index
    uint8_t* tocc; int32_t* assign; int32_t* nmatch;
    int32_t* pushT; uint8_t* pushBin;   // rotation-histogram pushes, replayed by the pruning step (k_init_resolve)
    int32_t mode; float nnratio; int32_t checkOri; int32_t thDist;
};

// pass 0: count candidates per query; pass 1: fill (index, distance key)
__global__ void k_proj_candidates(ProjArgs a, int pass)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.nq) return;
    if (a.qvalid && !a.qvalid[q]) { if (pass == 0) a.candCnt[q] = 0; return; }
    const float u = a.quvr[3 * q], v = a.quvr[3 * q + 1], r = a.quvr[3 * q + 2];
    const int minL = a.qlvl[2 * q], maxL = a.qlvl[2 * q + 1];
    // ORBmatcher.cc:91-96 / :1409-1415: a train feature with a right coordinate (mvuRight > 0) must also agree in it.
    // The test is independent of everything the sequential pass decides, so it is applied here, on both passes alike.
    const float qur = a.turight ? a.qur[q] : 0.f;
    auto stereo_ok = [&](int t) {
        if (!a.turight) return true;
        const float tr = a.turight[t];
        return !(tr > 0.f) || !(fabsf(__fsub_rn(qur, tr)) > r);
    };
    if (pass == 0) {
        int cnt = 0;
        for_each_in_area(a.grid, a.tkeys, a.cellStart, a.cellIdx, u, v, r, minL, maxL, [&](int t) { if (stereo_ok(t)) cnt++; });
        a.candCnt[q] = cnt;
    } else {
        uint32_t qw[8];
        const uint32_t* qp = (const uint32_t*)(a.qdesc + (int64_t)q * 32);
#pragma unroll
        for (int i = 0; i < 8; i++) qw[i] = qp[i];
        int pos = 0;
        const int base = a.candOff[q];
        for_each_in_area(a.grid, a.tkeys, a.cellStart, a.cellIdx, u, v, r, minL, maxL, [&](int t) {
            if (!stereo_ok(t)) return;
            const int d = hamming256(qw, (const uint32_t*)(a.tdesc + (int64_t)t * 32));
            a.candKey[base + pos] = ((uint32_t)d << 22) | ((uint32_t)(pos & 0x3FFFF) << 4) | (uint32_t)(a.tkeys[t].octave & 15);
            a.candIdx[base + pos] = t;
            pos++;
        });
    }
}

// one-query GetFeaturesInArea (tests)
__global__ void k_features_in_area(GridDev g, const KeyDev* keys, const int32_t* cellStart, const int32_t* cellIdx,
                                   float x, float y, float r, int minLevel, int maxLevel, int32_t* out, int cap, int32_t* nout)
{
    int n = 0;
    for_each_in_area(g, keys, cellStart, cellIdx, x, y, r, minLevel, maxLevel, [&](int i) { if (n < cap) out[n] = i; n++; });
    *nout = n;
}

// ------------------------------------------------------------------ remaining matchers
// Independent windowed best search: Fuse (:908-946, chi2 = 1), Fuse(KF,Scw) (:1053-1079) and both
// passes of SearchBySim3 (:1196-1222, :1276-1302).  One thread per projected map point.
struct WinArgs {
    GridDev grid;
    const KeyDev* tkeys; const int32_t* cellStart; const int32_t* cellIdx;
    const float* quvr; const float* qur; const int8_t* qpred; const uint8_t* qdesc; const uint8_t* qvalid;
    const uint8_t* tdesc; const float* turight; const float* invSigma2;
    int32_t nq; int32_t chi2;
    int32_t* bestIdx; int32_t* bestDist;
};

__global__ void k_window_best(WinArgs a)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.nq) return;
    int bestDist = 256, bestIdx = -1;
    if (!a.qvalid || a.qvalid[q]) {
        const float u = a.quvr[3 * q], v = a.quvr[3 * q + 1], radius = a.quvr[3 * q + 2];
        const int pred = a.qpred[q];
        const float ur = a.qur ? a.qur[q] : 0.f;
        uint32_t qw[8];
        const uint32_t* qp = (const uint32_t*)(a.qdesc + (int64_t)q * 32);
#pragma unroll
        for (int i = 0; i < 8; i++) qw[i] = qp[i];
        for_each_in_area(a.grid, a.tkeys, a.cellStart, a.cellIdx, u, v, radius, -1, -1, [&](int idx) {
            const KeyDev& kp = a.tkeys[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < pred - 1 || kpLevel > pred) return;
            if (a.chi2) {
                const float ex = __fsub_rn(u, kp.x), ey = __fsub_rn(v, kp.y);
                if (a.turight && a.turight[idx] >= 0) {
                    const float er = __fsub_rn(ur, a.turight[idx]);
                    const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                    if ((double)__fmul_rn(e2, a.invSigma2[kpLevel]) > 7.8) return;
                } else {
                    const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                    if ((double)__fmul_rn(e2, a.invSigma2[kpLevel]) > 5.99) return;
                }
            }
            const int d = hamming256(qw, (const uint32_t*)(a.tdesc + (int64_t)idx * 32));
            if (d < bestDist) { bestDist = d; bestIdx = idx; }
        });
    }
    a.bestIdx[q] = bestIdx;
    a.bestDist[q] = bestDist;
}

// SearchForInitialization (:407-522): sequential over the queries; per-train best distance so
// far lives in LDS.  Candidates/distances come from k_proj_candidates (window, octave 0).
__global__ __launch_bounds__(64) void k_init_resolve(ProjArgs a, int32_t* __restrict__ m12, int32_t* __restrict__ m21)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ilds[];
    uint16_t* matchedDist = (uint16_t*)ilds;  // [nt], 0xFFFF = INT_MAX
    __shared__ int hist[32];
    const int lane = threadIdx.x;
    for (int i = lane; i < a.nt; i += 64) { matchedDist[i] = 0xFFFF; m21[i] = -1; }
    for (int i = lane; i < a.nq; i += 64) m12[i] = -1;
    if (lane < 32) hist[lane] = 0;
    __syncthreads();
    int nPush = 0;  // the match count is recounted from m12 at the end (steals and pruning clear entries)
    for (int q = 0; q < a.nq; q++) {
        const int cs = a.candOff[q], ce = a.candOff[q + 1];
        if (ce == cs) continue;
        uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
        for (int c = cs + lane; c < ce; c += 64) {
            const uint32_t key = a.candKey[c];
            if ((uint32_t)matchedDist[a.candIdx[c]] <= (key >> 22)) continue;  // :441-442
            if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
        }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            const uint32_t o1 = __shfl_xor(k1, dd), o2 = __shfl_xor(k2, dd);
            top2_merge(k1, k2, o1, o2);
        }
        if (k1 == 0xFFFFFFFFu) continue;
        const int best = (int)(k1 >> 22);
        const float best2f = k2 == 0xFFFFFFFFu ? 2147483648.0f : (float)(int)(k2 >> 22);
        if (best <= a.thDist && (float)best < __fmul_rn(best2f, a.nnratio)) {
            const int t = a.candIdx[cs + (int)((k1 >> 4) & 0x3FFFF)];
            if (lane == 0) {
                const int prev = m21[t];
                if (prev >= 0) m12[prev] = -1;
                m12[q] = t;
                m21[t] = q;
                matchedDist[t] = (uint16_t)best;
                if (a.checkOri) {
                    const int bin = rot_bin(a.qang[q], a.tkeys[t].angle);
                    a.pushT[nPush] = q;
                    a.pushBin[nPush] = (uint8_t)bin;
                    hist[bin]++;
                }
            }
            if (a.checkOri) nPush++;
            __syncthreads();
        }
    }
    __syncthreads();
    if (a.checkOri && lane == 0) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < kHistoLength; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        for (int p = 0; p < nPush; p++) {
            const int bin = a.pushBin[p];
            if (bin != ind1 && bin != ind2 && bin != ind3) m12[a.pushT[p]] = -1;
        }
    }
    __syncthreads();
    int local = 0;
    for (int i = lane; i < a.nq; i += 64) local += m12[i] >= 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if (lane == 0) *a.nmatch = local;
}

// SearchForTriangulation (:659-825): vbMatched2 is never set in the reference, so every query
// is independent: best = smallest distance among candidates that pass the epipole-distance and
// epipolar-line tests, LAST one wins ties (the scan keeps dist <= bestDist).
struct TriArgs {
    const KeyDev* k1; const uint8_t* d1; const uint8_t* skip1; const float* ur1;
    const KeyDev* k2; const uint8_t* d2; const uint8_t* skip2; const float* ur2;
    const int32_t* start1; const int32_t* idx1; const int32_t* start2; const int32_t* idx2;
    const int32_t* pairA; const int32_t* pairB;
    float F[9]; float ex, ey;
    float sf2[16]; float sigma2[16];
    int32_t onlyStereo, checkOri;
    int32_t* m12; uint8_t* binOf; int32_t* hist;
};

__global__ __launch_bounds__(64) void k_triangulation_pairs(TriArgs a)
{
    const int lane = threadIdx.x;
    const int na = a.pairA[blockIdx.x], nb = a.pairB[blockIdx.x];
    const int s1 = a.start1[na], e1 = a.start1[na + 1], s2 = a.start2[nb], e2 = a.start2[nb + 1];
    for (int i1 = s1; i1 < e1; i1++) {
        const int q = a.idx1[i1];
        if (a.skip1 && a.skip1[q]) continue;
        const bool st1 = a.ur1 && a.ur1[q] >= 0;
        if (a.onlyStereo && !st1) continue;
        const KeyDev kp1 = a.k1[q];
        uint32_t qw[8];
        const uint32_t* qp = (const uint32_t*)(a.d1 + (int64_t)q * 32);
#pragma unroll
        for (int i = 0; i < 8; i++) qw[i] = qp[i];
        // epipolar line in image 2: l = x1' F12 (:141-146)
        const float la = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, a.F[0]), __fmul_rn(kp1.y, a.F[3])), a.F[6]);
        const float lb = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, a.F[1]), __fmul_rn(kp1.y, a.F[4])), a.F[7]);
        const float lc = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, a.F[2]), __fmul_rn(kp1.y, a.F[5])), a.F[8]);
        const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
        uint32_t best = 0xFFFFFFFFu;
        for (int i2 = s2 + lane; i2 < e2; i2 += 64) {
            const int t = a.idx2[i2];
            if (a.skip2 && a.skip2[t]) continue;
            const bool st2 = a.ur2 && a.ur2[t] >= 0;
            if (a.onlyStereo && !st2) continue;
            const int d = hamming256(qw, (const uint32_t*)(a.d2 + (int64_t)t * 32));
            if (d > 50) continue;  // TH_LOW
            const KeyDev& kp2 = a.k2[t];
            if (!st1 && !st2) {
                const float dx = __fsub_rn(a.ex, kp2.x), dy = __fsub_rn(a.ey, kp2.y);
                if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.f, a.sf2[kp2.octave & 15])) continue;
            }
            if (den == 0) continue;
            const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, kp2.x), __fmul_rn(lb, kp2.y)), lc);
            const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
            if (!((double)dsqr < 3.84 * (double)a.sigma2[kp2.octave & 15])) continue;
            const uint32_t key = ((uint32_t)d << 22) | (0x3FFFFFu - (uint32_t)(i2 - s2));
            best = min(best, key);
        }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) best = min(best, (uint32_t)__shfl_xor(best, dd));
        if (best != 0xFFFFFFFFu && lane == 0) {
            const int t = a.idx2[s2 + (int)(0x3FFFFFu - (best & 0x3FFFFFu))];
            a.m12[q] = t;
            if (a.checkOri) {
                const int bin = rot_bin(kp1.angle, a.k2[t].angle);
                a.binOf[q] = (uint8_t)bin;
                atomicAdd(&a.hist[bin], 1);
            }
        }
    }
}

// ------------------------------------------------------------------ SURVEY 8(f).4
// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307), one wave per map point.
// Row i's "median" is element k = (int)(0.5*(N-1)) of the sorted row = the value whose stable
// rank is k; the row with the least median wins (strict <, first index).
__global__ __launch_bounds__(64) void k_distinctive(const uint8_t* __restrict__ desc, const int32_t* __restrict__ start,
                                                   int32_t* __restrict__ bestIdx)
{
    extern __shared__ __attribute__((aligned(16))) int32_t rowd[];  // N distances of the current row
    const int p = blockIdx.x, lane = threadIdx.x;
    const int s0 = start[p], N = start[p + 1] - s0;
    if (N <= 0) { if (lane == 0) bestIdx[p] = -1; return; }
    const uint8_t* D = desc + (int64_t)s0 * 32;
    const int k = (int)(0.5 * (N - 1));
    int bestMedian = 0x7FFFFFFF, best = 0;
    for (int i = 0; i < N; i++) {
        uint32_t qw[8];
        const uint32_t* qp = (const uint32_t*)(D + (int64_t)i * 32);
#pragma unroll
        for (int t = 0; t < 8; t++) qw[t] = qp[t];
        for (int j = lane; j < N; j += 64) rowd[j] = i == j ? 0 : hamming256(qw, (const uint32_t*)(D + (int64_t)j * 32));
        __syncthreads();
        int median = -1;
        for (int j = lane; j < N; j += 64) {
            const int v = rowd[j];
            int rank = 0;
            for (int t = 0; t < N; t++) { const int u = rowd[t]; rank += (u < v || (u == v && t < j)) ? 1 : 0; }
            if (rank == k) median = v;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) median = max(median, __shfl_xor(median, d));
        if (median < bestMedian) { bestMedian = median; best = i; }
        __syncthreads();
    }
    if (lane == 0) bestIdx[p] = best;
}

// ------------------------------------------------------------------ SURVEY 8(f).3
// Frame::UndistortKeyPoints (src/Frame.cc:404-434) = cv::undistortPoints(..., mK, mDistCoef, Mat(), mK):
// 5 fixed-point iterations of the radial-tangential model in double, one thread per keypoint.
// Every operation is a separately rounded IEEE binary64 op (no contraction), like the host code.
// queries of SearchForInitialization: only level-0 keypoints search (ORBmatcher.cc:424-426)
__global__ void k_octave0_flags(const KeyDev* __restrict__ keys, int n, uint8_t* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = keys[i].octave <= 0;
}

// void Frame::ComputeStereoFromRGBD(const cv::Mat& imDepth)   src/Frame.cc:641-663: the depth under every (distorted) keypoint,
// imDepth.at<float>(v, u) with the FLOAT coordinates converted to int (truncation); d > 0 -> mvDepth = d, mvuRight =
// kpU.pt.x - mbf / d (binary32, the division first); else both -1.  A keypoint outside the depth image (the reference would
// read out of bounds) counts as d = 0.
__global__ void k_stereo_from_rgbd(const KeyDev* __restrict__ keys, const KeyDev* __restrict__ keysUn, int n, const float* __restrict__ depth,
                                   int w, int h, int stride, float mbf, float* __restrict__ uRight, float* __restrict__ outDepth)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = (int)keys[i].y, u = (int)keys[i].x;
    const float d = (u >= 0 && u < w && v >= 0 && v < h) ? depth[(int64_t)v * stride + u] : 0.f;
    const bool ok = d > 0.f;
    outDepth[i] = ok ? d : -1.f;
    uRight[i] = ok ? __fsub_rn(keysUn[i].x, __fdiv_rn(mbf, d)) : -1.f;
}

struct UndistArgs { double fx, fy, cx, cy, k1, k2, p1, p2, k3; };
__global__ void k_undistort(const KeyDev* __restrict__ in, int n, UndistArgs a, KeyDev* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    KeyDev kp = in[i];
    const double ifx = __ddiv_rn(1.0, a.fx), ify = __ddiv_rn(1.0, a.fy);
    double x = kp.x, y = kp.y;
    const double x0 = x = __dmul_rn(__dsub_rn(x, a.cx), ifx);
    const double y0 = y = __dmul_rn(__dsub_rn(y, a.cy), ify);
    for (int j = 0; j < 5; j++) {
        const double r2 = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
        // (1 + ((k7*r2 + k6)*r2 + k5)*r2) with k5..k7 = 0 evaluates to exactly 1
        const double num = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(0.0, r2), 0.0), r2), 0.0), r2));
        const double den = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(a.k3, r2), a.k2), r2), a.k1), r2));
        const double icdist = __ddiv_rn(num, den);
        const double deltaX = __dadd_rn(__dmul_rn(__dmul_rn(__dmul_rn(2.0, a.p1), x), y),
                                        __dmul_rn(a.p2, __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, x), x))));
        const double deltaY = __dadd_rn(__dmul_rn(a.p1, __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, y), y))),
                                        __dmul_rn(__dmul_rn(__dmul_rn(2.0, a.p2), x), y));
        x = __dmul_rn(__dsub_rn(x0, deltaX), icdist);
        y = __dmul_rn(__dsub_rn(y0, deltaY), icdist);
    }
    const double xx = __dadd_rn(__dadd_rn(__dmul_rn(a.fx, x), __dmul_rn(0.0, y)), a.cx);
    const double yy = __dadd_rn(__dadd_rn(__dmul_rn(0.0, x), __dmul_rn(a.fy, y)), a.cy);
    const double ww = __ddiv_rn(1.0, __dadd_rn(__dadd_rn(__dmul_rn(0.0, x), __dmul_rn(0.0, y)), 1.0));
    kp.x = (float)__dmul_rn(xx, ww);
    kp.y = (float)__dmul_rn(yy, ww);
    out[i] = kp;
}

}  // namespace orbm
