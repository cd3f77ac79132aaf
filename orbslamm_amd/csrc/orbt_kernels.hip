// orbt_kernels.hip -- the Tracking-shaped searches as ONE launch per call, batched over frame pairs (round 3).
//
// Reference: /root/reference/SingleRobotScenario/src
//   Frame::Frame tail: UndistortKeyPoints + AssignFeaturesToGrid   Frame.cc:196-210, 404-434, 230-245  -> k_frame_build
//   ORBmatcher::SearchByProjection x4                              ORBmatcher.cc:45-129, 292-405, 1330-1472, 1474-1601
//                                                                                                      -> k_proj_fused
// What Tracking runs per frame is SearchByProjection(CurrentFrame, LastFrame, th, bMono) (Tracking.cc:925-936).  Its
// queries are resolved one after the other in the reference: a query skips the train features that an EARLIER query
// (whose MapPoint has observations) took.  k_proj_fused keeps that order-dependent result exactly, without walking
// the queries serially:
//   * a query's outcome depends only on which of ITS candidates lower-indexed blocking queries have taken;
//   * so query q may decide as soon as, for the candidate(s) that determine its decision (the best one; best and second
//     in mode 3), no lower-indexed undecided blocking query lists that candidate -- nothing can take it away any more,
//     and everything ranked better is already taken for good (occupancy only grows);
//   * rounds: every undecided blocking query posts its index on its free candidates (LDS atomicMin), then every
//     undecided query evaluates its top-2 among the candidates free FOR IT and commits if they are posted by nobody
//     lower.  The lowest undecided query always commits, windows are local, so a frame pair takes a handful of rounds.
//   * "free for q" carries a time stamp: a feature taken by blocker b is occupied only for queries > b -- a
//     non-blocking query (MapPoint without observations, ORBmatcher.cc:87-89, 1405-1407) decides late but must see the
//     occupancy of its own turn.  assign[t] is the LAST writer in query order = the maximum index (atomicMax).
// One workgroup of 1024 threads per frame pair, grid = pairs.  The train frame's grid (positions, octaves, cell starts)
// is staged in LDS; candidates (distance | octave | index, in the reference's scan order) are listed once, in LDS when
// they fit behind the tables, else in a per-pair arena whose size is fixed up front (overflow is reported through
// nmatch < 0, never written past).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbt {

using orbm::GridDev;
using orbm::KeyDev;
using orbm::UndistArgs;

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kFree = 0x7FFFFFFF;
constexpr int kMaxQueryIters = 32;  // decided-bit per (thread, iteration): nq <= 32 * 1024

// exclusive scan of one value per thread over the workgroup; returns the exclusive prefix, *total = sum
__device__ __forceinline__ int block_scan_excl(int v, int* wsum, int* total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    __syncthreads();  // wsum may still be read from the previous use
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) { const int s = wsum[w]; if (w < wave) woff += s; tot += s; }
    *total = tot;
    return woff + incl - v;
}

// ------------------------------------------------------------------ frame set: B device-resident frames, one launch
struct FrameSetDev {  // slot s: keysUn + s*cap, desc + s*cap*32, ang + s*cap, cellStart + s*(ncell+1), cellIdx + s*cap, n + s
    KeyDev* keysUn; uint8_t* desc; float* ang; int32_t* cellStart; int32_t* cellIdx; int32_t* n;
    int32_t cap, ncell;
};

struct FrameBuildArgs {
    FrameSetDev fs;
    const KeyDev* srcKeys; const uint8_t* srcDesc; const int32_t* srcCount;  // frame i at srcKeys + i*srcCap, srcDesc + i*srcCap*32
    int32_t srcCap, srcN;  // srcCount == nullptr: every source frame holds srcN features
    int32_t slot0, slotMod;
    GridDev grid; UndistArgs und; int32_t undistort;
};

// cv::undistortPoints for one point, as k_undistort (orbm_kernels.hip) does it
__device__ __forceinline__ void undistort_point(const UndistArgs& a, float& px, float& py)
{
    const double ifx = __ddiv_rn(1.0, a.fx), ify = __ddiv_rn(1.0, a.fy);
    double x = px, y = py;
    const double x0 = x = __dmul_rn(__dsub_rn(x, a.cx), ifx);
    const double y0 = y = __dmul_rn(__dsub_rn(y, a.cy), ify);
    for (int j = 0; j < 5; j++) {
        const double r2 = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
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
    px = (float)__dmul_rn(xx, ww);
    py = (float)__dmul_rn(yy, ww);
}

// One workgroup per frame: mvKeysUn, the angle column, a private copy of the descriptors and mGrid as CSR
// (cell-major ix*rows+iy, ascending feature index inside a cell = the reference's push_back order, Frame.cc:236-244).
// LDS: 2 * ncell ints (counts -> starts, fill cursors).
__global__ __launch_bounds__(kThreads) void k_frame_build(FrameBuildArgs a)
{
    extern __shared__ int32_t gl[];
    __shared__ int wsum[kWaves];
    int32_t* cnt = gl;
    int32_t* cur = gl + a.fs.ncell;
    const int tid = threadIdx.x;
    const int src = blockIdx.x;
    const int slot = (a.slot0 + src) % a.slotMod;
    const int n = min(a.srcCount ? a.srcCount[src] : a.srcN, a.fs.cap);
    const KeyDev* __restrict__ sk = a.srcKeys + (int64_t)src * a.srcCap;
    KeyDev* __restrict__ dk = a.fs.keysUn + (int64_t)slot * a.fs.cap;
    float* __restrict__ da = a.fs.ang + (int64_t)slot * a.fs.cap;
    int32_t* __restrict__ cs = a.fs.cellStart + (int64_t)slot * (a.fs.ncell + 1);
    int32_t* __restrict__ ci = a.fs.cellIdx + (int64_t)slot * a.fs.cap;
    for (int c = tid; c < a.fs.ncell; c += kThreads) { cnt[c] = 0; cur[c] = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += kThreads) {
        KeyDev kp = sk[i];
        if (a.undistort) undistort_point(a.und, kp.x, kp.y);
        dk[i] = kp;
        da[i] = kp.angle;
        int px, py;
        if (orbm::pos_in_grid(a.grid, kp.x, kp.y, px, py)) atomicAdd(&cnt[px * a.grid.rows + py], 1);
    }
    {   // descriptors: 32 bytes per feature as two 16-byte lanes
        const uint4* __restrict__ s4 = (const uint4*)(a.srcDesc + (int64_t)src * a.srcCap * 32);
        uint4* __restrict__ d4 = (uint4*)(a.fs.desc + (int64_t)slot * a.fs.cap * 32);
        for (int i = tid; i < 2 * n; i += kThreads) d4[i] = s4[i];
    }
    __syncthreads();
    int carry = 0;
    for (int base = 0; base < a.fs.ncell; base += kThreads) {
        const int c = base + tid;
        const int v = c < a.fs.ncell ? cnt[c] : 0;
        int tot;
        const int ex = block_scan_excl(v, wsum, &tot);
        if (c < a.fs.ncell) { cnt[c] = carry + ex; cs[c] = carry + ex; }
        carry += tot;
    }
    if (tid == 0) { cs[a.fs.ncell] = carry; a.fs.n[slot] = n; }
    __syncthreads();
    for (int i = tid; i < n; i += kThreads) {
        const KeyDev kp = dk[i];  // written by this very thread
        int px, py;
        if (orbm::pos_in_grid(a.grid, kp.x, kp.y, px, py)) {
            const int c = px * a.grid.rows + py;
            ci[cnt[c] + atomicAdd(&cur[c], 1)] = i;
        }
    }
    __syncthreads();
    for (int c = tid; c < a.fs.ncell; c += kThreads) {  // restore insertion order inside every cell
        const int s = cnt[c], e = s + cur[c];
        for (int i = s + 1; i < e; i++) {
            const int v = ci[i];
            int j = i - 1;
            while (j >= s && ci[j] > v) { ci[j + 1] = ci[j]; j--; }
            ci[j + 1] = v;
        }
    }
}

// ------------------------------------------------------------------ SearchByProjection, fused and batched
struct ProjPair {
    // train side = the frame whose grid is searched (CurrentFrame of modes 3-5, the KeyFrame of mode 6)
    GridDev grid;
    const KeyDev* tkeys; const int32_t* cellStart; const int32_t* cellIdx; const uint8_t* tdesc;
    const int32_t* ntPtr; int32_t nt;            // ntPtr: count read on the device (frame set), else nt
    // query side: explicit arrays ...
    const float* quvr; const int8_t* qlvl; const uint8_t* qdesc; const float* qang;
    const uint8_t* qvalid; const uint8_t* qobs;
    const float* qur; const float* turight;      // stereo gate of modes 3/4 (null for mono)
    // ... or derived from a frame's undistorted keys (quvr == null): u, v = pt (identity pose), radius = th *
    // mvScaleFactors[octave], levels octave-1 .. octave+1 (ORBmatcher.cc:1381-1392), inside the image bounds (:1375-1378)
    const KeyDev* qkeys;
    const int32_t* nqPtr; int32_t nq;
    float th; float minX, maxX, minY, maxY;
    // in / out per train feature
    const uint8_t* toccIn; uint8_t* toccOut;     // toccIn null: nothing occupied
    int32_t* assign; int32_t initAssign;         // initAssign: assign starts as all -1 (not read)
    int32_t* nmatch;                             // < 0: candidate arena overflow, -(needed entries) - 1
    // scratch
    int32_t* candOff; uint32_t* cand; int32_t candCap; int32_t* qres;
    int32_t* stats;                              // optional: [0] rounds, [1] candidates
};

struct ProjCommon {
    int32_t mode; float nnratio; int32_t checkOri; int32_t thDist;
    const float* scale; int32_t nlevels;
    int32_t tCap;      // LDS entries per per-train array (>= every pair's nt)
    int32_t cellCap;   // LDS entries of the cell-start table (>= cols*rows + 1 of every pair)
    int32_t ldsCand;   // candidate entries that fit in LDS behind the tables; longer lists go to the pair's arena
};

// LDS (dynamic), in this order:
//   recX, recY (float) and recI (index | octave << 24) of the train features in GRID order  3 * tCap dwords
//   occBy, minUnd[2], winner                                                                 4 * tCap dwords
//   candidates                                                                               ldsCand dwords
//   cell starts (uint16)                                                                     cellCap halves
// The whole matcher-side state of a frame (2000 features: 62 KB) sits in one CU's 160 KB: GetFeaturesInArea becomes
// a walk at LDS latency, and since cells are stored column-major (ix * rows + iy) one grid column of a window is ONE
// contiguous run of records in the reference's scan order (Frame.cc:352-376).
template <class F>
__device__ __forceinline__ void lds_area(const GridDev& g, const float* recX, const float* recY, const uint32_t* recI,
                                         const uint16_t* cst, float x, float y, float r, int minLevel, int maxLevel, F f)
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
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
        const int j1 = cst[ix * g.rows + nMaxCellY + 1];
        for (int j = cst[ix * g.rows + nMinCellY]; j < j1; j++) {
            const uint32_t io = recI[j];
            const int oct = (int)(io >> 24);
            if (bCheckLevels) {
                if (oct < minLevel) continue;
                if (maxLevel >= 0 && oct > maxLevel) continue;
            }
            const float distx = __fsub_rn(recX[j], x), disty = __fsub_rn(recY[j], y);
            if (fabsf(distx) < r && fabsf(disty) < r) f((int)(io & 0xFFFFFF), oct);
        }
    }
}

__device__ __forceinline__ void proj_body(const ProjPair& P, const ProjCommon& c)
{
    extern __shared__ int32_t tl[];
    __shared__ int hist[32];
    __shared__ int wsum[kWaves];
    __shared__ int sPending[3];
    __shared__ int sInd[3];
    __shared__ int sCount;
    float* recX = (float*)tl;
    float* recY = recX + c.tCap;
    uint32_t* recI = (uint32_t*)(recY + c.tCap);
    int32_t* occBy = (int32_t*)(recI + c.tCap);  // kFree, -1 (occupied on entry) or the blocking query that took the feature
    int32_t* minUnd0 = occBy + c.tCap;           // two copies, used by alternate rounds (the idle one is cleared meanwhile)
    int32_t* winner = occBy + 3 * c.tCap;        // last query (in query order) that took the feature in this call
    uint32_t* candL = (uint32_t*)(occBy + 4 * c.tCap);
    uint16_t* cst = (uint16_t*)(candL + c.ldsCand);
    const int tid = threadIdx.x;
    const int nt = min(P.ntPtr ? *P.ntPtr : P.nt, c.tCap);
    const int nq = min(P.nqPtr ? *P.nqPtr : P.nq, kMaxQueryIters * kThreads);
    const bool useRot = c.checkOri && (c.mode == 4 || c.mode == 5);
    const bool obsRule = c.mode == 3 || c.mode == 4;
    const int ncell = min(P.grid.cols * P.grid.rows, c.cellCap - 1);

    for (int t = tid; t < nt; t += kThreads) {
        occBy[t] = (P.toccIn && P.toccIn[t]) ? -1 : kFree;
        minUnd0[t] = kFree; minUnd0[c.tCap + t] = kFree;
        winner[t] = -1;
    }
    if (tid < 32) hist[tid] = 0;
    if (tid < 3) sPending[tid] = 0;
    if (tid == 0) sCount = 0;
    if (nq <= 0 || nt <= 0) {
        for (int t = tid; t < nt; t += kThreads) {
            if (P.initAssign) P.assign[t] = -1;
            P.toccOut[t] = (P.toccIn && P.toccIn[t]) ? 1 : 0;
        }
        if (tid == 0) { *P.nmatch = 0; if (P.stats) { P.stats[0] = 0; P.stats[1] = 0; } }
        return;
    }
    // the train frame's grid into LDS
    for (int ci = tid; ci <= ncell; ci += kThreads) cst[ci] = (uint16_t)P.cellStart[ci];
    const int ngrid = min(P.cellStart[ncell], nt);
    for (int j = tid; j < ngrid; j += kThreads) {
        const int i = P.cellIdx[j];
        const KeyDev& kp = P.tkeys[i];
        recX[j] = kp.x; recY[j] = kp.y;
        recI[j] = (uint32_t)i | ((uint32_t)kp.octave << 24);
    }
    __syncthreads();

    // the query's search window; false = the reference skips this query before GetFeaturesInArea
    auto window = [&](int q, float& u, float& v, float& r, int& minL, int& maxL) -> bool {
        if (P.qvalid && !P.qvalid[q]) return false;
        if (P.quvr) {
            u = P.quvr[3 * q]; v = P.quvr[3 * q + 1]; r = P.quvr[3 * q + 2];
            minL = P.qlvl[2 * q]; maxL = P.qlvl[2 * q + 1];
            return true;
        }
        const KeyDev& k = P.qkeys[q];
        u = k.x; v = k.y;
        if (u < P.minX || u > P.maxX) return false;
        if (v < P.minY || v > P.maxY) return false;
        const int o = k.octave;
        r = __fmul_rn(P.th, c.scale[min(max(o, 0), c.nlevels - 1)]);
        minL = o - 1; maxL = o + 1;
        return true;
    };
    auto stereo_ok = [&](int q, int t, float r) {
        if (!P.turight) return true;
        const float tr = P.turight[t];
        return !(tr > 0.f) || !(fabsf(__fsub_rn(P.qur[q], tr)) > r);
    };

    // ---- candidates: count, scan, fill (reference scan order: GetFeaturesInArea, Frame.cc:327-380).
    // Only mode 3 looks at anything but the best free candidate (its ratio test reads the second best whatever its
    // distance); in the other modes a candidate farther than the acceptance threshold can never be taken, so it is
    // not listed at all -- lists shrink to the plausible matches and so does the contention between queries.
    const bool listAll = c.mode == 3;
    int carry = 0;
    for (int base = 0; base < nq; base += kThreads) {
        const int q = base + tid;
        int cnt = 0;
        float u, v, r; int minL, maxL;
        if (q < nq && window(q, u, v, r, minL, maxL)) {
            uint32_t qw[8];
            const uint32_t* qp = (const uint32_t*)(P.qdesc + (int64_t)q * 32);
#pragma unroll
            for (int i = 0; i < 8; i++) qw[i] = qp[i];
            lds_area(P.grid, recX, recY, recI, cst, u, v, r, minL, maxL, [&](int t, int) {
                if (!stereo_ok(q, t, r)) return;
                if (listAll || orbm::hamming256(qw, (const uint32_t*)(P.tdesc + (int64_t)t * 32)) <= c.thDist) cnt++;
            });
        }
        int tot;
        const int ex = block_scan_excl(cnt, wsum, &tot);
        if (q < nq) P.candOff[q] = carry + ex;
        carry += tot;
    }
    if (tid == 0) P.candOff[nq] = carry;
    const bool inLds = carry <= c.ldsCand;
    if (!inLds && carry > P.candCap) {  // uniform: nothing is written past the arena
        if (tid == 0) { *P.nmatch = -carry - 1; if (P.stats) { P.stats[0] = 0; P.stats[1] = carry; } }
        return;
    }
    __syncthreads();
    uint32_t decided = 0;
    for (int j = 0, q = tid; q < nq; q += kThreads, j++) {
        const int b0 = P.candOff[q], b1 = P.candOff[q + 1];
        P.qres[q] = -1;
        if (b1 == b0) { decided |= 1u << j; continue; }
        float u, v, r; int minL, maxL;
        window(q, u, v, r, minL, maxL);
        uint32_t qw[8];
        const uint32_t* qp = (const uint32_t*)(P.qdesc + (int64_t)q * 32);
#pragma unroll
        for (int i = 0; i < 8; i++) qw[i] = qp[i];
        int pos = b0;
        lds_area(P.grid, recX, recY, recI, cst, u, v, r, minL, maxL, [&](int t, int oct) {
            if (!stereo_ok(q, t, r)) return;
            const int d = orbm::hamming256(qw, (const uint32_t*)(P.tdesc + (int64_t)t * 32));
            if (!listAll && d > c.thDist) return;
            // entries stay in the reference's scan order, so "first of equals wins" (strict <) needs no rank
            const uint32_t e = ((uint32_t)d << 20) | ((uint32_t)(oct & 15) << 16) | (uint32_t)t;
            if (inLds) candL[pos] = e; else P.cand[pos] = e;
            pos++;
        });
    }
    __syncthreads();

    // ---- rounds
    int nAcc = 0, round = 0;
    for (;; round++) {
        int32_t* minUnd = minUnd0 + (round & 1) * c.tCap;
        int32_t* idle = minUnd0 + ((round + 1) & 1) * c.tCap;
        for (int j = 0, q = tid; q < nq; q += kThreads, j++) {
            if (decided & (1u << j)) continue;
            if (obsRule && P.qobs && !P.qobs[q]) continue;  // takes nothing away from anybody
            for (int k = P.candOff[q], e = P.candOff[q + 1]; k < e; k++) {
                const uint32_t cd = inLds ? candL[k] : P.cand[k];
                const int t = (int)(cd & 0xFFFF);
                if ((int)(cd >> 20) <= c.thDist && occBy[t] >= q) atomicMin(&minUnd[t], q);  // (it can only ever take one within the threshold)
            }
        }
        if (tid == 0) sPending[(round + 1) % 3] = 0;  // three counters in rotation: the one being read after a round's last barrier is not reset before the round after next
        __syncthreads();
        int pend = 0;
        for (int j = 0, q = tid; q < nq; q += kThreads, j++) {
            if (decided & (1u << j)) continue;
            int b1 = 256, b2 = 256;
            uint32_t e1 = 0, e2 = 0;
            bool has2 = false;
            for (int k = P.candOff[q], e = P.candOff[q + 1]; k < e; k++) {
                const uint32_t cd = inLds ? candL[k] : P.cand[k];
                if (occBy[cd & 0xFFFF] < q) continue;  // taken before this query's turn (or occupied on entry)
                const int d = (int)(cd >> 20);
                if (d < b1) { b2 = b1; e2 = e1; has2 = b2 < 256; b1 = d; e1 = cd; }   // ORBmatcher.cc:102-114
                else if (d < b2) { b2 = d; e2 = cd; has2 = true; }
            }
            if (b1 > c.thDist || b1 >= 256) { decided |= 1u << j; continue; }  // can only get worse: no match
            const int t1 = (int)(e1 & 0xFFFF), t2 = (int)(e2 & 0xFFFF);
            const bool stable = minUnd[t1] >= q && (c.mode != 3 || !has2 || minUnd[t2] >= q);
            if (!stable) { pend++; continue; }
            decided |= 1u << j;
            if (c.mode == 3 && has2 && ((e1 >> 16) & 15) == ((e2 >> 16) & 15) &&
                (float)b1 > __fmul_rn(c.nnratio, (float)b2)) continue;  // ORBmatcher.cc:120-121
            nAcc++;
            atomicMax(&winner[t1], q);
            if (!obsRule || !P.qobs || P.qobs[q]) occBy[t1] = q;
            int res = t1;
            if (useRot) {
                const int bin = orbm::rot_bin(P.qang ? P.qang[q] : P.qkeys[q].angle, P.tkeys[t1].angle);
                atomicAdd(&hist[bin], 1);
                res |= bin << 24;
            }
            P.qres[q] = res;
        }
        for (int t = tid; t < nt; t += kThreads) idle[t] = kFree;
        if (pend) atomicAdd(&sPending[round % 3], pend);
        __syncthreads();
        if (sPending[round % 3] == 0) break;
    }

    // ---- write back; rotation consistency (ComputeThreeMaxima :1603-1644 + pruning)
    if (useRot && tid == 0) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < orbm::kHistoLength; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        sInd[0] = ind1; sInd[1] = ind2; sInd[2] = ind3;
    }
    __syncthreads();
    // every assign entry is written ONCE (it may live in pinned host memory): the pruning goes through the LDS table
    int nPruned = 0;
    if (useRot) {
        for (int q = tid; q < nq; q += kThreads) {
            const int res = P.qres[q];
            if (res < 0) continue;
            const int bin = res >> 24;
            if (bin != sInd[0] && bin != sInd[1] && bin != sInd[2]) { winner[res & 0xFFFF] = -2; nPruned++; }
        }
        __syncthreads();
    }
    for (int t = tid; t < nt; t += kThreads) {
        const int w = winner[t];
        if (w == -2) P.assign[t] = -1;
        else if (P.initAssign || w >= 0) P.assign[t] = w;
        P.toccOut[t] = occBy[t] != kFree;
    }
    int local = nAcc - nPruned;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if ((tid & 63) == 0 && local) atomicAdd(&sCount, local);
    __syncthreads();
    if (tid == 0) { *P.nmatch = sCount; if (P.stats) { P.stats[0] = round + 1; P.stats[1] = carry; } }
}

__global__ __launch_bounds__(kThreads) void k_proj_fused(const ProjPair* __restrict__ pairs, ProjCommon c)
{
    const ProjPair P = pairs[blockIdx.x];
    proj_body(P, c);
}

// The frame-to-frame search over pairs of a frame set's slots: the pair records are made here from the slot numbers
// (kernel arguments), so a call uploads nothing and waits for nothing.
constexpr int kTrackMaxPairs = 128;
struct TrackArgs {
    FrameSetDev fs;
    GridDev grid;
    float th, minX, maxX, minY, maxY;
    uint8_t* occ; int32_t* assign; int32_t* nmatch; int32_t* stats;   // [pair][cap], [pair][cap] (pinned host), [pair], [pair][2]
    int32_t* candOff; uint32_t* cand; int32_t candCap; int32_t* qres;  // [pair][cap+1], [pair][candCap], [pair][cap]
    int32_t pair0;
    int16_t cur[kTrackMaxPairs], last[kTrackMaxPairs];
};

__global__ __launch_bounds__(kThreads) void k_track_fused(TrackArgs a, ProjCommon c)
{
    const int b = blockIdx.x, p = a.pair0 + b;
    const int64_t C = a.fs.cap;
    const int cs = a.cur[b], ls = a.last[b];
    ProjPair P;
    P.grid = a.grid;
    P.tkeys = a.fs.keysUn + cs * C; P.cellStart = a.fs.cellStart + (int64_t)cs * (a.fs.ncell + 1); P.cellIdx = a.fs.cellIdx + cs * C;
    P.tdesc = a.fs.desc + cs * C * 32; P.ntPtr = a.fs.n + cs; P.nt = 0;
    P.quvr = nullptr; P.qlvl = nullptr; P.qdesc = a.fs.desc + ls * C * 32; P.qang = nullptr; P.qvalid = nullptr; P.qobs = nullptr;
    P.qur = nullptr; P.turight = nullptr;
    P.qkeys = a.fs.keysUn + ls * C; P.nqPtr = a.fs.n + ls; P.nq = 0;
    P.th = a.th; P.minX = a.minX; P.maxX = a.maxX; P.minY = a.minY; P.maxY = a.maxY;
    P.toccIn = nullptr; P.toccOut = a.occ + p * C;
    P.assign = a.assign + p * C; P.initAssign = 1;
    P.nmatch = a.nmatch + p;
    P.candOff = a.candOff + (int64_t)p * (C + 1); P.cand = a.cand + (int64_t)p * a.candCap; P.candCap = a.candCap;
    P.qres = a.qres + p * C; P.stats = a.stats + 2 * p;
    proj_body(P, c);
}

}  // namespace orbt
