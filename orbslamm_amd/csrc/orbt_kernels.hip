// orbt_kernels.hip -- the Tracking-shaped searches, batched over frames and frame pairs (round 3).
//
// Reference: /root/reference/SingleRobotScenario/src
//   Frame::Frame tail: UndistortKeyPoints + AssignFeaturesToGrid   Frame.cc:196-210, 404-434, 230-245  -> k_frame_build
//   ORBmatcher::SearchByProjection x4                              ORBmatcher.cc:45-129, 292-405, 1330-1472, 1474-1601
//                                                   -> k_proj_candidates + k_proj_resolve (k_track_*: frame-set pairs)
// What Tracking runs per frame is SearchByProjection(CurrentFrame, LastFrame, th, bMono) (Tracking.cc:925-936).  Its
// queries are resolved one after the other in the reference: a query skips the train features that an EARLIER query
// (whose MapPoint has observations) took.  The resolve kernel keeps that order-dependent result exactly, without walking
// the queries serially:
//   * a query's outcome depends only on which of ITS candidates lower-indexed blocking queries have taken;
//   * so query q may decide as soon as, for the candidate(s) that determine its decision (the best one; best and second
//     in mode 3), no lower-indexed undecided blocking query lists that candidate -- nothing can take it away any more,
//     and everything ranked better is already taken for good (occupancy only grows);
//   * rounds: every live candidate entry posts its query on its train feature and itself on its query (LDS atomicMin),
//     then every undecided query reads its best among the candidates free FOR IT and commits if nobody lower is posted
//     on it.  The lowest undecided query always commits, windows are local, so a frame pair takes ten to twenty rounds.
//   * "free for q" carries a time stamp: a feature taken by blocker b is occupied only for queries > b -- a
//     non-blocking query (MapPoint without observations, ORBmatcher.cc:87-89, 1405-1407) decides late but must see the
//     occupancy of its own turn.  assign[t] is the LAST writer in query order = the maximum index (atomicMax).
// Candidates come from k_*_candidates: workgroups of 128 queries (four lanes each) walk the train frame's grid, staged in
// LDS, list what lies in the windows in the reference's scan order, fill in the Hamming distances with a flat loop and
// hand what is within the acceptance threshold to a per-pair arena whose size is fixed up front (overflow is reported
// through nmatch < 0, never written past).  k_*_resolve: one workgroup of 1024 threads per frame pair.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orbt {

using orbm::GridDev;
using orbm::KeyDev;
using orbm::UndistArgs;

#ifndef ORBT_THREADS
#define ORBT_THREADS 1024
#endif
constexpr int kThreads = ORBT_THREADS;
constexpr int kWaves = kThreads / 64;
constexpr int kFree = 0x7FFFFFFF;
constexpr int kMaxQueryIters = 64;  // nq <= 64 * 1024: a query's index is 16 bits in the per-train post (proj_resolve_rounds)

#ifndef ORBT_ROUND
#define ORBT_ROUND 1   // the round whose phases ORBT_MARK(11..15) time
#endif
#ifdef ORBT_PHASE_TIMING  // tools/proj_phases.sh: where a workgroup of the resolve / candidates kernels spends its time (100 MHz wall clock)
__device__ unsigned long long g_orbtPhase[16];
__device__ int g_orbtCensus[64][2];
__device__ unsigned long long g_orbtBuild[12];   // phases of k_frame_build (block 0)
#define ORBT_BMARK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_orbtBuild[i] = wall_clock64(); } while (0)   // per round of the resolve: live entries, undecided queries at its head
#define ORBT_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_orbtPhase[i] = wall_clock64(); } while (0)
#else
#define ORBT_MARK(i) do { } while (0)
#define ORBT_BMARK(i) do { } while (0)
#endif

// exclusive scan of one value per thread over the workgroup; returns the exclusive prefix, *total = sum
__device__ __forceinline__ int block_scan_excl(int v, int* wsum, int* total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int incl = orbx::wave_incl_scan(v);
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
struct FrameSetDev {  // slot s: keysUn + s*cap, desc + s*cap*32, ang + s*cap, cellStart + s*(ncell+1), cellIdx + s*cap, rec + s*cap, n + s
    KeyDev* keysUn; uint8_t* desc; float* ang; int32_t* cellStart; int32_t* cellIdx;
    uint4* rec;   // the features in GRID order as {x, y, index | octave << 24, 0}: what a window walk reads, in one piece
    int32_t* n;
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
// LDS: 2 * ncell ints (counts -> starts, fill cursors) + cap uint16 (the cell lists, sorted in place before they go out)
// + cap x {x, y, octave} (12 bytes): every later phase reads the keys from LDS, not back from memory -- the kernel is a
// chain of short phases and a memory round trip per phase was most of its 31 us.
__device__ __forceinline__ void frame_build_body(const FrameBuildArgs& a, const int src)
{
    extern __shared__ __attribute__((aligned(16))) int32_t gl[];
    __shared__ int wsum[kWaves];
    constexpr int kBigCells = 64;
    __shared__ int sBigN;
    __shared__ uint16_t sBig[kBigCells];
    const int capE = (a.fs.cap + 1) & ~1;
    int32_t* cnt = gl;
    int32_t* cur = gl + a.fs.ncell;
    float* kx = (float*)(gl + 2 * a.fs.ncell);
    float* ky = kx + capE;
    int32_t* ko = (int32_t*)(ky + capE);
    uint16_t* lst = (uint16_t*)(ko + capE);
    const int tid = threadIdx.x;
    const int slot = (a.slot0 + src) % a.slotMod;
    const int n = min(a.srcCount ? a.srcCount[src] : a.srcN, a.fs.cap);
    const KeyDev* __restrict__ sk = a.srcKeys + (int64_t)src * a.srcCap;
    KeyDev* __restrict__ dk = a.fs.keysUn + (int64_t)slot * a.fs.cap;
    float* __restrict__ da = a.fs.ang + (int64_t)slot * a.fs.cap;
    int32_t* __restrict__ cs = a.fs.cellStart + (int64_t)slot * (a.fs.ncell + 1);
    int32_t* __restrict__ ci = a.fs.cellIdx + (int64_t)slot * a.fs.cap;
    uint4* __restrict__ rec = a.fs.rec + (int64_t)slot * a.fs.cap;
    ORBT_BMARK(0);
    if (tid == 0) sBigN = 0;
    for (int c = tid; c < a.fs.ncell; c += kThreads) { cnt[c] = 0; cur[c] = 0; }
    __syncthreads();
    ORBT_BMARK(1);
    for (int i = tid; i < n; i += kThreads) {
        KeyDev kp = sk[i];
        if (a.undistort) undistort_point(a.und, kp.x, kp.y);
        dk[i] = kp;
        da[i] = kp.angle;
        kx[i] = kp.x; ky[i] = kp.y; ko[i] = kp.octave;
        int px, py;
        if (orbm::pos_in_grid(a.grid, kp.x, kp.y, px, py)) atomicAdd(&cnt[px * a.grid.rows + py], 1);
    }
    {   // descriptors: 32 bytes per feature as two 16-byte lanes
        const uint4* __restrict__ s4 = (const uint4*)(a.srcDesc + (int64_t)src * a.srcCap * 32);
        uint4* __restrict__ d4 = (uint4*)(a.fs.desc + (int64_t)slot * a.fs.cap * 32);
        ORBT_BMARK(2);
        for (int i = tid; i < 2 * n; i += kThreads) d4[i] = s4[i];
    }
    ORBT_BMARK(3);
    __syncthreads();
    ORBT_BMARK(4);
    // cell starts: ONE scan over the workgroup, a thread owning a run of consecutive cells (3 at 64 x 48 cells) -- it was a
    // scan (two barriers) per 1024 cells
    int carry = 0;
    {
        const int per = (a.fs.ncell + kThreads - 1) / kThreads;
        const int c0 = min(tid * per, a.fs.ncell), c1 = min(c0 + per, a.fs.ncell);
        int sum = 0;
        for (int c = c0; c < c1; c++) sum += cnt[c];
        int run = block_scan_excl(sum, wsum, &carry);
        for (int c = c0; c < c1; c++) { const int v = cnt[c]; cnt[c] = run; cs[c] = run; run += v; }
    }
    if (tid == 0) { cs[a.fs.ncell] = carry; a.fs.n[slot] = n; }
    __syncthreads();
    ORBT_BMARK(5);
    for (int i = tid; i < n; i += kThreads) {
        int px, py;
        if (orbm::pos_in_grid(a.grid, kx[i], ky[i], px, py)) {
            const int c = px * a.grid.rows + py;
            lst[cnt[c] + atomicAdd(&cur[c], 1)] = (uint16_t)i;
        }
    }
    __syncthreads();
    ORBT_BMARK(6);
    // restore insertion order inside every cell.  A thread sorts its cell's few entries by insertion; a crowded cell (a corner
    // cluster: 20 features in one 19 x 8 px cell) would cost that one thread ~n^2 / 2 dependent LDS accesses with the
    // other 1023 waiting -- 4 of this kernel's 12 us -- so cells above kSmall go to a list and are rank-sorted by a wave each.
    constexpr int kSmall = 8;
    for (int c = tid; c < a.fs.ncell; c += kThreads) {
        const int s = cnt[c], k = cur[c];
        if (k < 2) continue;
        if (k > kSmall) {
            const int p = atomicAdd(&sBigN, 1);
            if (p < kBigCells) { sBig[p] = (uint16_t)c; continue; }
            for (int i = s + 1; i < s + k; i++) {   // (more crowded cells than the list holds: sorted here after all)
                const uint16_t v = lst[i];
                int j = i - 1;
                while (j >= s && lst[j] > v) { lst[j + 1] = lst[j]; j--; }
                lst[j + 1] = v;
            }
            continue;
        }
        // up to eight entries: into registers (independent loads), a 19-exchange sorting network, back -- two LDS round trips
        // where the insertion sort made a dependent one per comparison
        uint32_t v[kSmall];
#pragma unroll
        for (int i = 0; i < kSmall; i++) v[i] = i < k ? (uint32_t)lst[s + i] : 0xFFFFFFFFu;
#define ORBT_CX(i, j) do { const uint32_t lo_ = min(v[i], v[j]), hi_ = max(v[i], v[j]); v[i] = lo_; v[j] = hi_; } while (0)
        ORBT_CX(0, 1); ORBT_CX(2, 3); ORBT_CX(4, 5); ORBT_CX(6, 7);
        ORBT_CX(0, 2); ORBT_CX(1, 3); ORBT_CX(4, 6); ORBT_CX(5, 7);
        ORBT_CX(1, 2); ORBT_CX(5, 6); ORBT_CX(0, 4); ORBT_CX(3, 7);
        ORBT_CX(1, 5); ORBT_CX(2, 6);
        ORBT_CX(1, 4); ORBT_CX(3, 6);
        ORBT_CX(2, 4); ORBT_CX(3, 5);
        ORBT_CX(3, 4);
#undef ORBT_CX
#pragma unroll
        for (int i = 0; i < kSmall; i++) if (i < k) lst[s + i] = (uint16_t)v[i];
    }
    __syncthreads();
    {
        const int nBig = min(sBigN, kBigCells), lane = tid & 63;
        for (int b = tid >> 6; b < nBig; b += kWaves) {
            const int c = sBig[b], s = cnt[c], k = cur[c];
            if (k <= 64) {   // one entry per lane; its rank = the entries below it (all different)
                const int v = lane < k ? (int)lst[s + lane] : 0x7FFFFFFF;
                int rank = 0;
                for (int j = 0; j < k; j++) rank += __builtin_amdgcn_readlane(v, j) < v;
                if (lane < k) lst[s + rank] = (uint16_t)v;   // (every lane read its entry before any lane writes: one load instruction above)
            } else if (lane == 0) {
                for (int i = s + 1; i < s + k; i++) {
                    const uint16_t v = lst[i];
                    int j = i - 1;
                    while (j >= s && lst[j] > v) { lst[j + 1] = lst[j]; j--; }
                    lst[j + 1] = v;
                }
            }
        }
    }
    __syncthreads();
    ORBT_BMARK(7);
    for (int j = tid; j < carry; j += kThreads) {
        const int i = lst[j];
        ci[j] = i;
        rec[j] = make_uint4(__float_as_uint(kx[i]), __float_as_uint(ky[i]), (uint32_t)i | ((uint32_t)ko[i] << 24), 0u);
    }
    ORBT_BMARK(8);
}

__global__ __launch_bounds__(kThreads) void k_frame_build(FrameBuildArgs a)
{
    frame_build_body(a, blockIdx.x);
}

// The live chain (a frame set attached to the extractor): the frames' results go to the host from the SAME launch that
// builds their Frame tail -- workgroups [0, nframes) build, the kPackParts per frame behind them write keypoints and
// descriptors into the ticket's pinned block and raise its flag (orbx::pack_host_part).  Both only read the extractor's
// result set; as two launches the copy (6 us per frame over the link, a few waves) ran in front of the build (10 us).
constexpr int kPackParts = 4;   // x 1024 threads = the sixteen 256-thread workgroups of k_pack_host
__global__ __launch_bounds__(kThreads) void k_frame_build_pack(FrameBuildArgs a, orbx::PackArgs pa, int nframes)
{
    if ((int)blockIdx.x < nframes) { frame_build_body(a, blockIdx.x); return; }
    const int q = blockIdx.x - nframes;
    orbx::pack_host_part(pa, q / kPackParts, q % kPackParts, kPackParts, nframes * kPackParts);
}

// ------------------------------------------------------------------ SearchByProjection, batched: candidates + resolve
struct ProjPair {
    // train side = the frame whose grid is searched (CurrentFrame of modes 3-5, the KeyFrame of mode 6)
    GridDev grid;
    const KeyDev* tkeys; const int32_t* cellStart; const uint4* trec; const uint8_t* tdesc;   // trec: FrameSetDev::rec
    const int32_t* ntPtr; int32_t nt;            // ntPtr: count read on the device (frame set), else nt
    // query side: explicit arrays ...
    const float* quvr; const int8_t* qlvl; const uint8_t* qdesc; const float* qang;
    const uint8_t* qvalid; const uint8_t* qobs;
    const float* qur; const float* turight;      // stereo gate of modes 3/4 (null for mono)
    // ... or derived from a frame's undistorted keys (quvr == null): u, v = pt (identity pose), radius = th *
    // mvScaleFactors[octave], levels octave-1 .. octave+1 (ORBmatcher.cc:1381-1392), inside the image bounds (:1375-1378)
    const KeyDev* qkeys;
    const int32_t* nqPtr; int32_t nq;            // nqPtr: count read on the device (capped by nq when nq > 0), else nq
    float th; float minX, maxX, minY, maxY;
    // in / out per train feature
    const uint8_t* toccIn; uint8_t* toccOut;     // toccIn null: nothing occupied
    int32_t* assign; int32_t initAssign;         // initAssign: assign starts as all -1 (not read)
    int32_t* nmatch;                             // < 0: candidate arena overflow, -(needed entries) - 1
    // scratch
    int32_t* total;                              // candidates listed so far; zero on entry, left zero by the resolve
    int32_t* candOff; int32_t* candCnt;          // per query: its list = cand[candOff[q] .. +candCnt[q])
    uint2* cand; int32_t candCap;                // {distance << 20 | octave << 16 | train index, query}
    int32_t* qres;                               // [nq]
    int32_t* qscr;                               // [2 nq ints + nq bytes] best / second / state of a query when they do not fit LDS
    int32_t* tscr;                               // [3 tCap] the per-train tables of the resolve when they do not fit LDS (ProjCommon::big)
    int32_t* stats;                              // optional: [0] rounds, [1] candidates
    int32_t* flag; int32_t flagValue;            // optional (pinned host): raised behind this pair's results -- a caller polling it
                                                 // skips the wake-up of an event wait (the one-frame-per-call path)
};

// every thread's result writes are performed system-wide, then ONE store publishes them (the host polls the word)
__device__ __forceinline__ void proj_publish(const ProjPair& P)
{
    if (!P.flag) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(P.flag, P.flagValue, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct ProjCommon {
    int32_t mode; float nnratio; int32_t checkOri; int32_t thDist;
    const float* scale; int32_t nlevels;
    int32_t tCap;      // LDS entries per per-train array (>= every pair's nt)
    int32_t cellCap;   // LDS entries of the cell-start table (>= cols*rows + 1 of every pair; even)
    int32_t stageCap;  // candidates: entries (8 bytes) of a slice's staging buffer in LDS behind the grid
    int32_t qCap;      // resolve: queries whose tables fit in LDS   } more of either: the rounds work in memory
    int32_t ldsCand;   // resolve: candidate entries that fit in LDS }
    int32_t waveTail;    // resolve: the last rounds by the first wave alone (0: the whole workgroup to the end)
    int32_t lanes;       // candidates: lanes per query, kCandLanes or kCandLanesWide (the host sizes the grid of slices by it)
    int32_t interleave;  // candidates: deal the queries to the slices wave by wave instead of in consecutive runs (few pairs)
    int32_t big;         // more than ~8 K train features (or a grid whose cell table does not fit): the window walk reads the frame's grid
                         // records and cell starts from memory, the resolve keeps its per-train tables in ProjPair::tscr -- the
                         // reference's loops take any size (Frame.cc:327-380, ORBmatcher.cc:45-129); up to 65 535 features a frame
};

constexpr int kCandThreads = 512;
constexpr int kCandCols = 4;                             // column slots per query: the grid columns of its window are dealt round-robin
constexpr int kCandLanes = 4;                            // lanes per query, throughput form: one per column slot
constexpr int kCandLanesWide = 16;                       // latency form (few pairs): four lanes per column slot, each a quarter of the column's run
constexpr int kCandQueries = kCandThreads / kCandLanes;  // queries per workgroup (a "slice") in the throughput form
constexpr int kCandPasses = 4;                           // columns per slot; a wider window is walked by lane 0 alone

// The train frame's grid in LDS: its features in GRID order as {x, y, index | octave << 24, 0} (one 16-byte read per
// visit) and the cell starts (uint16).  GetFeaturesInArea becomes a walk at LDS latency, and since cells are stored
// column-major (ix * rows + iy) one grid column of a window is ONE contiguous run of records in the reference's scan
// order (Frame.cc:352-376).
struct AreaWin { int x0, x1, y0, y1; bool check; };

// cell range of GetFeaturesInArea (Frame.cc:332-346); false: the window misses the grid
__device__ __forceinline__ bool area_window(const GridDev& g, float x, float y, float r, int minLevel, int maxLevel, AreaWin& w)
{
    w.x0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, g.minX), r), g.invW));
    if (w.x0 < 0) w.x0 = 0;
    if (w.x0 >= g.cols) return false;
    w.x1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, g.minX), r), g.invW));
    if (w.x1 > g.cols - 1) w.x1 = g.cols - 1;
    if (w.x1 < 0) return false;
    w.y0 = (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, g.minY), r), g.invH));
    if (w.y0 < 0) w.y0 = 0;
    if (w.y0 >= g.rows) return false;
    w.y1 = (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, g.minY), r), g.invH));
    if (w.y1 > g.rows - 1) w.y1 = g.rows - 1;
    if (w.y1 < 0) return false;
    w.check = (minLevel > 0) || (maxLevel >= 0);
    return true;
}

template <class CST, class F>
__device__ __forceinline__ void area_column(const GridDev& g, const uint4* rec, const CST* cst, const AreaWin& w, int ix,
                                            float x, float y, float r, int minLevel, int maxLevel, int seg, int nseg, F f)
{
    int j0 = cst[ix * g.rows + w.y0], j1 = cst[ix * g.rows + w.y1 + 1];
    if (nseg > 1) {   // this lane's part of the column's run (parts in lane order = the run's order)
        const int len = j1 - j0, per = (len + nseg - 1) / nseg;
        j1 = j0 + min((seg + 1) * per, len);
        j0 = j0 + min(seg * per, len);
    }
    for (int j = j0; j < j1; j++) {
        const uint4 e = rec[j];
        const int oct = (int)(e.z >> 24);
        if (w.check) {
            if (oct < minLevel) continue;
            if (maxLevel >= 0 && oct > maxLevel) continue;
        }
        const float distx = __fsub_rn(__uint_as_float(e.x), x), disty = __fsub_rn(__uint_as_float(e.y), y);
        if (fabsf(distx) < r && fabsf(disty) < r) f((int)(e.z & 0xFFFFFF), oct);
    }
}

__device__ __forceinline__ int hamming_rows(const uint8_t* __restrict__ qd, uint32_t q, const uint8_t* __restrict__ td, uint32_t t)
{
    const uint4* qp = (const uint4*)(qd + (int64_t)q * 32);
    const uint4* tp = (const uint4*)(td + (int64_t)t * 32);
    const uint4 a0 = qp[0], a1 = qp[1], b0 = tp[0], b1 = tp[1];
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// exclusive scan over the 256 threads of a candidate workgroup
__device__ __forceinline__ int wg_scan_excl(int v, int* wsum, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = orbx::wave_incl_scan(v);
    __syncthreads();  // wsum may still be read from the previous use
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kCandThreads / 64; w++) { const int s = wsum[w]; if (w < wave) woff += s; tot += s; }
    *total = tot;
    return woff + incl - v;
}

// exclusive prefix over the LANES lanes of a query, *tot = their sum
template <int LANES>
__device__ __forceinline__ int quad_scan_excl(int v, int c, int* tot)
{
    // (row_shr on the DPP path instead of __shfl_up through the LDS unit; a group never straddles a row of 16 lanes, and the
    // lanes that would read a neighbouring group's value do not add it)
    static_assert(LANES == 4 || LANES == 16, "groups of 4 or 16 lanes");
    int incl = v, t;
    t = __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false); if (c >= 1) incl += t;
    t = __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false); if (c >= 2) incl += t;
    if constexpr (LANES == 16) {
        t = __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false); if (c >= 4) incl += t;
        t = __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false); if (c >= 8) incl += t;
    }
    *tot = __shfl(incl, LANES - 1, LANES);
    return incl - v;
}

// Candidates of a slice of 64 queries, four lanes per query (a query's grid columns are dealt to its lanes, so the
// 7-column window of a coarse-level query costs what the 3-column window of a fine one does): the windows are walked in
// LDS (count, scan, walk again and list into an LDS staging buffer), the Hamming distances are filled in by a flat loop
// -- one entry per thread and step, all loads independent (inside the divergent window walk every gather would cost the
// whole wave a memory round trip) -- and what lies within the threshold goes to a chunk of the pair's arena allocated
// with one atomicAdd.  Lists keep the reference's scan order (GetFeaturesInArea, Frame.cc:327-380): column by column.
template <int LANES, bool GLDS>
__device__ __forceinline__ void proj_candidates_body(const ProjPair& P, const ProjCommon& c, int slice)
{
    constexpr int kSeg = LANES / kCandCols;     // lanes per column slot
    constexpr int kWaveQ = 64 / LANES;          // queries per wave
    constexpr int kSliceQ = kCandThreads / LANES;
    extern __shared__ __attribute__((aligned(16))) int32_t tl[];
    __shared__ int wsum[kCandThreads / 64];
    __shared__ int sBase;
    // GLDS: the train frame's grid records and cell starts are staged in LDS (every shipped shape); otherwise (ProjCommon::big)
    // the walk reads them where they lie -- same lists, same order, at memory latency
    typedef typename std::conditional<GLDS, uint16_t, int32_t>::type CstT;
    const uint4* rec; uint2* stage; const CstT* cst;
    if constexpr (GLDS) { rec = (const uint4*)tl; stage = (uint2*)((uint4*)tl + c.tCap); cst = (const CstT*)(stage + c.stageCap); }
    else { rec = P.trec; stage = (uint2*)tl; cst = (const CstT*)P.cellStart; }
    const int tid = threadIdx.x, lc = tid & (LANES - 1), lcol = lc / kSeg, lseg = lc % kSeg;
    const int nt = min(P.ntPtr ? *P.ntPtr : P.nt, c.tCap);
    const int nq = min(P.nqPtr ? (P.nq > 0 ? min(*P.nqPtr, P.nq) : *P.nqPtr) : P.nq, kMaxQueryIters * kThreads);
    if (nt <= 0 || (c.interleave ? slice * kWaveQ : slice * kSliceQ) >= nq) return;
    ORBT_MARK(4);
    const int ncell = min(P.grid.cols * P.grid.rows, c.cellCap - 1);
    if constexpr (GLDS) {
        uint16_t* cstW = (uint16_t*)(stage + c.stageCap);
        uint4* recW = (uint4*)tl;
        for (int ci = tid; ci <= ncell; ci += kCandThreads) cstW[ci] = (uint16_t)P.cellStart[ci];
        const int ngrid = min(P.cellStart[ncell], nt);
        for (int j = tid; j < ngrid; j += kCandThreads) recW[j] = P.trec[j];
    }
    // the query's search window; !ok = the reference skips this query before GetFeaturesInArea
    // Few pairs (a live stream): the slices INTERLEAVE the queries, in units of a wave's 16.  Queries come ordered by pyramid
    // level and a coarse-level window holds many times the features of a fine one: with consecutive runs the last
    // workgroups of a frame pair walked for 26 us while the first were done after 11 (tools/proj_phases.sh).  A wave's lanes
    // still walk windows of one size.  Many pairs keep the consecutive runs (workgroups abound; 7 % faster there).
    const int q = c.interleave ? ((tid >> 6) * (int)gridDim.x + slice) * kWaveQ + (tid & 63) / LANES : slice * kSliceQ + tid / LANES;
    float u = 0.f, v = 0.f, r = 0.f; int minL = 0, maxL = 0;
    bool ok = q < nq && !(P.qvalid && !P.qvalid[q]);
    if (ok) {
        if (P.quvr) {
            u = P.quvr[3 * q]; v = P.quvr[3 * q + 1]; r = P.quvr[3 * q + 2];
            minL = P.qlvl[2 * q]; maxL = P.qlvl[2 * q + 1];
        } else {
            const KeyDev& k = P.qkeys[q];
            u = k.x; v = k.y;
            if (u < P.minX || u > P.maxX || v < P.minY || v > P.maxY) ok = false;
            const int o = k.octave;
            r = __fmul_rn(P.th, c.scale[min(max(o, 0), c.nlevels - 1)]);
            minL = o - 1; maxL = o + 1;
        }
    }
    AreaWin w{};
    if (ok) ok = area_window(P.grid, u, v, r, minL, maxL, w);
    const bool wide = ok && (w.x1 - w.x0 + 1 > kCandCols * kCandPasses);
    const float qur = (ok && P.turight) ? P.qur[q] : 0.f;
    auto stereo_ok = [&](int t) {
        if (!P.turight) return true;
        const float tr = P.turight[t];
        return !(tr > 0.f) || !(fabsf(__fsub_rn(qur, tr)) > r);
    };
    // f(pass, column) for this lane's columns, in the order the query's list wants them within the lane
    auto my_columns = [&](auto f) {
        if (!ok) return;
        if (wide) { if (lc == 0) for (int ix = w.x0; ix <= w.x1; ix++) f(0, ix); return; }
#pragma unroll
        for (int pi = 0; pi < kCandPasses; pi++) { const int ix = w.x0 + pi * kCandCols + lcol; if (ix <= w.x1) f(pi, ix); }
    };
    __syncthreads();
    ORBT_MARK(5);
    int cnt[kCandPasses] = {0, 0, 0, 0};
    my_columns([&](int pi, int ix) {
        int n = 0;
        area_column(P.grid, rec, cst, w, ix, u, v, r, minL, maxL, lseg, wide ? 1 : kSeg, [&](int t, int) { if (stereo_ok(t)) n++; });
#pragma unroll
        for (int k = 0; k < kCandPasses; k++) if (k == pi) cnt[k] += n;
    });
    ORBT_MARK(6);
    // a query's list: pass by pass, inside a pass lane by lane (= column by column)
    int off[kCandPasses], qtot = 0;
#pragma unroll
    for (int pi = 0; pi < kCandPasses; pi++) {
        int ptot;
        off[pi] = qtot + quad_scan_excl<LANES>(cnt[pi], lc, &ptot);
        qtot += ptot;
    }
    int tot;
    const int qoff = __shfl(wg_scan_excl(lc == 0 ? qtot : 0, wsum, &tot), 0, LANES);
    if (tot == 0) {  // uniform
        if (q < nq && lc == 0) { P.candOff[q] = 0; P.candCnt[q] = 0; }
        return;
    }
    // Only mode 3 looks at anything but the best free candidate (its ratio test reads the second best whatever its
    // distance); in the other modes a candidate farther than the acceptance threshold can never be taken, so it is
    // dropped here -- lists shrink to the plausible matches and so does the contention between queries.
    const int dMax = c.mode == 3 ? 256 : c.thDist;
    const bool staged = tot <= c.stageCap;  // else (very wide windows): straight into the arena, unfiltered
    int base = 0;
    if (!staged) {
        if (tid == 0) sBase = atomicAdd(P.total, tot);
        __syncthreads();
        base = sBase;
        const bool fits = base + tot <= P.candCap;  // uniform; the resolve reports the overflow (total > candCap)
        if (q < nq && lc == 0) { P.candOff[q] = base + qoff; P.candCnt[q] = fits ? qtot : 0; }
        if (!fits) return;
    }
    ORBT_MARK(7);
    uint2* list = staged ? stage : P.cand + base;
    int wrun = off[0];  // wide: lane 0 walks all columns one after the other, positions simply continue
    my_columns([&](int pi, int ix) {
        int pos = qoff + wrun;
        if (!wide) {
#pragma unroll
            for (int k = 0; k < kCandPasses; k++) if (k == pi) pos = qoff + off[k];
        }
        area_column(P.grid, rec, cst, w, ix, u, v, r, minL, maxL, lseg, wide ? 1 : kSeg, [&](int t, int oct) {
            if (!stereo_ok(t)) return;
            list[pos++] = make_uint2(((uint32_t)(oct & 15) << 16) | (uint32_t)t, (uint32_t)q);
        });
        wrun = pos - qoff;
    });
    __syncthreads();
    ORBT_MARK(8);
    for (int k = tid; k < tot; k += kCandThreads) {
        const uint2 e = list[k];
        list[k].x = e.x | ((uint32_t)hamming_rows(P.qdesc, e.y, P.tdesc, e.x & 0xFFFF) << 20);
    }
    __syncthreads();
    if (!staged) {
        // the distance cut in place, inside the query's own segment of the arena (order kept; what is left behind the
        // survivors is marked dead: the resolve walks the whole chunk)
        if (lc == 0 && qtot > 0 && dMax < 256) {
            int keep = 0;
            for (int k = qoff; k < qoff + qtot; k++) {
                const uint2 e = list[k];
                if ((int)(e.x >> 20) <= dMax) list[qoff + keep++] = e;
            }
            for (int k = qoff + keep; k < qoff + qtot; k++) list[k].x = 0xFFFFFFFFu;
            P.candCnt[q] = keep;
        }
        return;
    }
    ORBT_MARK(9);
    // compaction: each lane takes a quarter of its query's list
    const int seg = (qtot + LANES - 1) / LANES;
    const int k0 = qoff + min(lc * seg, qtot), k1 = qoff + min((lc + 1) * seg, qtot);
    int keep = 0;
    for (int k = k0; k < k1; k++) keep += (int)(stage[k].x >> 20) <= dMax;
    int qkeep;
    const int kofs = quad_scan_excl<LANES>(keep, lc, &qkeep);
    int ktot;
    const int kex = __shfl(wg_scan_excl(lc == 0 ? qkeep : 0, wsum, &ktot), 0, LANES);
    if (tid == 0) sBase = ktot ? atomicAdd(P.total, ktot) : 0;
    __syncthreads();
    base = sBase;
    const bool fits = base + ktot <= P.candCap;
    if (q < nq && lc == 0) { P.candOff[q] = base + kex; P.candCnt[q] = fits ? qkeep : 0; }
    if (!fits) return;
    ORBT_MARK(10);
    int pos = base + kex + kofs;
    for (int k = k0; k < k1; k++) {
        const uint2 e = stage[k];
        if ((int)(e.x >> 20) <= dMax) P.cand[pos++] = e;
    }
    ORBT_MARK(11);
}

// The sequential part, in parallel rounds (see the head of this file).  One workgroup of 1024 per pair.
// The rounds are ENTRY-parallel: a flat loop over all candidate entries posts, for every live entry, the query on its
// train feature (atomicMin) and the entry on its query (atomicMin of distance << 22 | position in the query's list: the
// first of equal distances in scan order wins, like the reference's strict <); then one step per undecided query
// reads its best (mode 3: a second flat pass finds the runner-up), checks that nobody lower has posted on it and
// commits.  A thread-per-query walk of the lists costs every wave its longest list, a chain of dependent LDS reads per
// entry, in every round; the flat loop keeps all lanes busy and its reads independent.
// LDS (dynamic): occBy, minUnd, winner (3 * tCap dwords) | per
// query: offset, best, second = result (3 * qCap dwords) | candidates (ldsCand dwords) | their queries (ldsCand halves) |
// two live lists (2 * ldsCand halves) | per query: state byte (qCap bytes) | two lists of undecided queries (2 * qCap halves).
// L = false: the per-query tables and the lists stay in memory (more queries or candidates than the LDS plan holds).
constexpr uint8_t kQDecided = 1, kQBlocking = 2;
#ifndef ORBT_TAIL_LIVE
#define ORBT_TAIL_LIVE 512
#define ORBT_TAIL_QUERIES 128
#endif
constexpr int kTailLive = ORBT_TAIL_LIVE, kTailQueries = ORBT_TAIL_QUERIES;   // at most this much left: the first wave runs the remaining rounds alone

template <bool L>
__device__ __forceinline__ void proj_resolve_rounds(const ProjPair& P, const ProjCommon& c, int nq, int nt, int total,
                                                   int32_t* occBy, int32_t* minUnd0, int32_t* winner, int32_t* qtab, int* hist,
                                                   int* sPending, int* sInd, int* sCount, int* sLive)
{
    const int tid = threadIdx.x;
    const bool useRot = c.checkOri && (c.mode == 4 || c.mode == 5);
    const bool obsRule = c.mode == 3 || c.mode == 4;
    // Only mode 3 looks at anything but the best free candidate (its ratio test reads the second best whatever its
    // distance); in the other modes a candidate farther than the acceptance threshold can never be taken: skipped.
    const int dMax = c.mode == 3 ? 256 : c.thDist;
    int32_t *qoff, *best1, *best2, *qres;
    uint32_t* candL = nullptr; uint16_t* ownL = nullptr; uint16_t* live0 = nullptr; uint8_t* qst;
    uint16_t* pend0 = nullptr;   // L: the undecided queries, two lists used by alternate rounds (a round costs what is still open)
    if constexpr (L) {
        // (the result shares the runner-up's word: a decided query's bests are never touched again, and kFree there reads "none")
        qoff = qtab; best1 = qtab + c.qCap; best2 = qtab + 2 * c.qCap; qres = best2;
        candL = (uint32_t*)(qtab + 3 * c.qCap);
        ownL = (uint16_t*)(candL + c.ldsCand);
        live0 = ownL + c.ldsCand;
        qst = (uint8_t*)(live0 + 2 * c.ldsCand);
        pend0 = (uint16_t*)(qst + c.qCap);
        for (int k = tid; k < total; k += kThreads) { const uint2 e = P.cand[k]; candL[k] = e.x; ownL[k] = (uint16_t)e.y; live0[k] = (uint16_t)k; }
        if (tid == 0) { sLive[0] = total; sLive[1] = 0; }
    } else {
        qoff = P.candOff; best1 = P.qscr; best2 = P.qscr + nq; qres = P.qres; qst = (uint8_t*)(P.qscr + 2 * nq);
    }
    auto entry = [&](int k) -> uint2 { if constexpr (L) return make_uint2(candL[k], ownL[k]); else return P.cand[k]; };
    for (int q = tid; q < nq; q += kThreads) {
        const int cnt = nt > 0 ? P.candCnt[q] : 0;
        if constexpr (L) qoff[q] = P.candOff[q];
        best1[q] = kFree; best2[q] = kFree;
        if constexpr (!L) qres[q] = -1;
        qst[q] = (cnt == 0 ? kQDecided : 0) | ((!obsRule || !P.qobs || P.qobs[q]) ? kQBlocking : 0);
    }
    __syncthreads();
    ORBT_MARK(1);

    // One round, run by the first G threads of the workgroup: all 1024, or -- in the tail, where a handful of live entries and
    // undecided queries are left and a round is nothing but its chain of ~10 dependent LDS round trips and two barriers --
    // the first wave alone: no s_barrier, no other waves in the LDS queue (tools/resolve_round_phases.py: 1.8 us per
    // round for the workgroup however little is left).  Returns whether anybody is still undecided.
    int nAcc = 0;
    auto one_round = [&](auto gc, const int round) -> int {
        constexpr int G = decltype(gc)::value;
        auto bar = [&]() {
            if constexpr (G == kThreads) __syncthreads();
            else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
        };
        // minUnd[t]: the lowest blocking query that posted on train feature t in THIS round, kept as round + 1 << 16 | 0xFFFF - q
        // under atomicMax -- what an earlier round left is simply older, nothing to clear between rounds (a pass over
        // the table and its share of a barrier: 0.7 of a round's ~2.2 us)
        uint32_t* minUnd = (uint32_t*)minUnd0;
        const uint32_t rkey = (uint32_t)(round + 1) << 16;
        auto posted = [&](uint32_t t) -> int { const uint32_t m = minUnd[t]; return (m & 0xFFFF0000u) == rkey ? (int)(0xFFFFu - (m & 0xFFFFu)) : kFree; };
        const int nPendList = L ? sPending[round & 1] : 0;   // (final since the last round's closing barrier)
#ifdef ORBT_PHASE_TIMING
        if (L && threadIdx.x == 0 && blockIdx.x == 0 && round < 64) { g_orbtCensus[round][0] = sLive[round & 1]; g_orbtCensus[round][1] = round ? nPendList : nq; if (round < 63) g_orbtCensus[round + 1][0] = -1; }
#endif
        if (round == ORBT_ROUND) ORBT_MARK(11);
        if constexpr (L) {
            // over the LIVE entries only (those of undecided queries whose train feature is still free for them -- both
            // conditions are final once false), compacting the list for the next round on the way: a round costs what is
            // still open, not what was listed
            const uint16_t* lv = live0 + (round & 1) * c.ldsCand;
            uint16_t* nx = live0 + ((round + 1) & 1) * c.ldsCand;
            const int nLive = sLive[round & 1];
            // two entries per thread and trip, their LDS reads issued side by side: a round is a chain of LDS round trips
            // (live slot -> entry, owner -> state, occupancy), and with one entry per trip the slowest wave of a
            // 1024-thread workgroup spent 2.2 us here in an early round (tools/resolve_round_phases.py)
            for (int i0 = 0; i0 < nLive; i0 += 2 * G) {
                const int ia = i0 + tid, ib = ia + G;
                const bool ina = ia < nLive, inb = ib < nLive;
                const int ka = lv[ina ? ia : 0], kb = lv[inb ? ib : 0];
                const uint32_t exa = candL[ka], exb = candL[kb];
                const int qa = ownL[ka], qb = ownL[kb];
                const int da = (int)(exa >> 20), ta = (int)(exa & 0xFFFF), db = (int)(exb >> 20), tb = (int)(exb & 0xFFFF);
                const uint8_t sa = qst[qa], sb = qst[qb];
                const int oa = occBy[ta], ob = occBy[tb];
                const int fa = qoff[qa], fb = qoff[qb];
                const bool keepa = ina && !(sa & kQDecided) && da <= dMax && oa >= qa;
                const bool keepb = inb && !(sb & kQDecided) && db <= dMax && ob >= qb;
                if (keepa) {
                    if ((sa & kQBlocking) && da <= c.thDist) atomicMax(&minUnd[ta], rkey | (uint32_t)(0xFFFF - qa));
                    atomicMin(&best1[qa], (da << 22) | (ka - fa));
                }
                if (keepb) {
                    if ((sb & kQBlocking) && db <= c.thDist) atomicMax(&minUnd[tb], rkey | (uint32_t)(0xFFFF - qb));
                    atomicMin(&best1[qb], (db << 22) | (kb - fb));
                }
                const uint64_t bala = __builtin_amdgcn_ballot_w64(keepa), balb = __builtin_amdgcn_ballot_w64(keepb);
                if (bala | balb) {
                    const int na = __popcll(bala);
                    int base = 0;
                    if ((tid & 63) == 0) base = atomicAdd(&sLive[(round + 1) & 1], na + __popcll(balb));
                    base = __builtin_amdgcn_readfirstlane(base);   // (not a ds_bpermute round trip)
                    if (keepa) nx[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bala >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bala, 0))] = (uint16_t)ka;
                    if (keepb) nx[base + na + __builtin_amdgcn_mbcnt_hi((uint32_t)(balb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)balb, 0))] = (uint16_t)kb;
                }
            }
        } else {
        for (int k = tid; k < total; k += G) {
                const uint2 e = entry(k);
                const int q = (int)e.y, d = (int)(e.x >> 20), t = (int)(e.x & 0xFFFF);
                const uint8_t st = qst[q];
                if ((st & kQDecided) || d > dMax) continue;
                if (occBy[t] < q) continue;  // taken before this query's turn (or occupied on entry)
                // a query without observations takes nothing away from anybody; nobody can take what is beyond the threshold
                if ((st & kQBlocking) && d <= c.thDist) atomicMax(&minUnd[t], rkey | (uint32_t)(0xFFFF - q));
                atomicMin(&best1[q], (d << 22) | (k - qoff[q]));
            }
        }
        if (round == ORBT_ROUND) ORBT_MARK(12);
        bar();
        if (c.mode == 3) {  // the runner-up: the best of what is left (ORBmatcher.cc:102-114)
            const int nIt = L ? sLive[(round + 1) & 1] : total;
            for (int i = tid; i < nIt; i += G) {
                int k = i;
                if constexpr (L) k = (live0 + ((round + 1) & 1) * c.ldsCand)[i];
                const uint2 e = entry(k);
                const int q = (int)e.y, t = (int)(e.x & 0xFFFF);
                if ((qst[q] & kQDecided) || occBy[t] < q) continue;
                const int key = ((int)(e.x >> 20) << 22) | (k - qoff[q]);
                if (key != best1[q]) atomicMin(&best2[q], key);
            }
            bar();
        }
        if (round == ORBT_ROUND) ORBT_MARK(13);
        int pend = 0;
        // two queries per thread and trip (independent of each other: nothing written here is read here), their reads -- state
        // and bests, list offset, the entries, the posts on them -- issued side by side.  From the second round on the
        // trips run over the list of queries the round before left undecided (L), not over all of them.
        const bool fromList = L && round > 0;
        const int nIter = fromList ? nPendList : nq;
        const uint16_t* pl = pend0 + (round & 1) * c.qCap;
        uint16_t* pn = pend0 + ((round + 1) & 1) * c.qCap;
        if (L && tid == 0) sPending[round & 1] = 0;   // (read at the head of the round, filled again by the next one)
        for (int i0 = 0; i0 < nIter; i0 += 2 * G) {
            int qq[2]; bool und[2], go[2], has2[2], wait[2]; uint8_t st[2]; int k1[2], k2[2], b1[2]; uint32_t e1[2], e2[2]; int m1[2], m2[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u * G + tid;
                const bool in = i < nIter;
                if constexpr (L) qq[u] = fromList ? pl[in ? i : 0] : (in ? i : 0);
                else qq[u] = in ? i : 0;
                st[u] = qst[qq[u]]; k1[u] = best1[qq[u]]; k2[u] = best2[qq[u]];
                und[u] = in && !(st[u] & kQDecided);
                wait[u] = false;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                b1[u] = k1[u] == kFree ? 256 : k1[u] >> 22;
                go[u] = und[u] && !(b1[u] > c.thDist || b1[u] >= 256);
                has2[u] = go[u] && c.mode == 3 && k2[u] != kFree;
                const int off = qoff[qq[u]];
                e1[u] = go[u] ? entry(off + (k1[u] & 0x3FFFFF)).x : 0u;
                e2[u] = has2[u] ? entry(off + (k2[u] & 0x3FFFFF)).x : 0u;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                m1[u] = go[u] ? posted(e1[u] & 0xFFFF) : 0;
                m2[u] = has2[u] ? posted(e2[u] & 0xFFFF) : 0;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                if (!und[u]) continue;
                const int q = qq[u];
                best1[q] = kFree; best2[q] = kFree;
                if (!go[u]) { qst[q] = st[u] | kQDecided; continue; }  // can only get worse: no match
                const int t1 = (int)(e1[u] & 0xFFFF);
                const bool stable = m1[u] >= q && (!has2[u] || m2[u] >= q);
                if (!stable) { pend++; wait[u] = true; continue; }
                qst[q] = st[u] | kQDecided;
                if (has2[u] && ((e1[u] >> 16) & 15) == ((e2[u] >> 16) & 15) &&
                    (float)b1[u] > __fmul_rn(c.nnratio, (float)(k2[u] >> 22))) continue;  // ORBmatcher.cc:120-121
                nAcc++;
                atomicMax(&winner[t1], q);
                if (st[u] & kQBlocking) occBy[t1] = q;
                qres[q] = t1;
            }
            if constexpr (L) {
                const uint64_t bala = __builtin_amdgcn_ballot_w64(wait[0]), balb = __builtin_amdgcn_ballot_w64(wait[1]);
                if (bala | balb) {
                    const int na = __popcll(bala);
                    int base = 0;
                    if ((tid & 63) == 0) base = atomicAdd(&sPending[(round + 1) & 1], na + __popcll(balb));
                    base = __builtin_amdgcn_readfirstlane(base);   // (not a ds_bpermute round trip)
                    if (wait[0]) pn[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bala >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bala, 0))] = (uint16_t)qq[0];
                    if (wait[1]) pn[base + na + __builtin_amdgcn_mbcnt_hi((uint32_t)(balb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)balb, 0))] = (uint16_t)qq[1];
                }
            }
        }
        if (round == ORBT_ROUND) ORBT_MARK(14);
        if (L && tid == 0) sLive[round & 1] = 0;   // read in this round's first pass, filled again by the next round's
        // "anybody still undecided?" rides on the round's last barrier (every thread with pending queries used to add to ONE
        // LDS word: up to a thousand serialised atomics, 2 us of an early round)
        int anyPend;
        if constexpr (G == kThreads) anyPend = __syncthreads_or(pend);
        else { bar(); anyPend = __builtin_amdgcn_ballot_w64(pend != 0) != 0; }
        if (round == ORBT_ROUND) ORBT_MARK(15);
        return anyPend;
    };
    int round = 0;
    bool tail = false;   // the first wave finishes alone
    for (;; round++) {
        if (!one_round(std::integral_constant<int, kThreads>{}, round)) break;
        if constexpr (L) {
            // (both counts are final behind the round's closing barrier, and the same for every thread)
            if (c.waveTail && sLive[(round + 1) & 1] <= kTailLive && sPending[(round + 1) & 1] <= kTailQueries) { tail = true; break; }
        }
    }
    if (tail) {
        if (tid < 64) for (round++;; round++) if (!one_round(std::integral_constant<int, 64>{}, round)) break;
        __syncthreads();
    }
    ORBT_MARK(2);

    // ---- write back; rotation consistency (ComputeThreeMaxima :1603-1644 + pruning).  The bins are computed here, in
    // one go for all accepted queries: inside the rounds the two angle gathers would add a memory round trip to every round
    if (useRot) {
        for (int q = tid; q < nq; q += kThreads) {
            const int t1 = qres[q];
            if (t1 < 0 || t1 == kFree) continue;
            const int bin = orbm::rot_bin(P.qang ? P.qang[q] : P.qkeys[q].angle, P.tkeys[t1].angle);
            atomicAdd(&hist[bin], 1);
            best1[q] = bin;  // (the table is free now)
        }
        __syncthreads();
        if (tid < 64) {   // (the first wave: its lanes fetch the bins in ONE LDS trip, lane 0 then walks them in registers --
                          // thirty dependent LDS reads by one thread were 3 us with the whole workgroup waiting)
            const int hl = hist[tid & 31];
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
#pragma unroll
            for (int i = 0; i < orbm::kHistoLength; i++) {
                const int s = __builtin_amdgcn_readlane(hl, i);
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
            if (tid == 0) { sInd[0] = ind1; sInd[1] = ind2; sInd[2] = ind3; }
        }
        __syncthreads();
    }
    // every assign entry is written ONCE (it may live in pinned host memory): the pruning goes through the LDS table
    int nPruned = 0;
    if (useRot) {
        for (int q = tid; q < nq; q += kThreads) {
            const int t1 = qres[q];
            if (t1 < 0 || t1 == kFree) continue;
            const int bin = best1[q];
            if (bin != sInd[0] && bin != sInd[1] && bin != sInd[2]) { winner[t1] = -2; nPruned++; }
        }
    }
    __syncthreads();
    for (int t = tid; t < nt; t += kThreads) {
        const int w = winner[t];
        if (w == -2) P.assign[t] = -1;
        else if (P.initAssign || w >= 0) P.assign[t] = w;
        P.toccOut[t] = occBy[t] != kFree;
    }
    int local = nAcc - nPruned;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) local += __shfl_xor(local, d);
    if ((tid & 63) == 0 && local) atomicAdd(sCount, local);
    __syncthreads();
    if (tid == 0) {
        if (nq > 0 && nt > 0) *P.total = 0;  // left zero for the next call's candidate kernel
        *P.nmatch = *sCount;
        if (P.stats) { P.stats[0] = (nq > 0 && nt > 0) ? round + 1 : 0; P.stats[1] = total; }
    }
    proj_publish(P);
    ORBT_MARK(3);
}

// the resolve's few static words, declared ONCE in the kernel (a __shared__ inside the templated body would be laid out once per
// instantiation, and every byte of static LDS comes off the dynamic budget the plan hands out: kProjLdsBudget)
struct ResolveShared { int hist[32]; int sPending[3]; int sInd[3]; int sCount; int sLive[2]; };

template <bool TLDS>
__device__ __forceinline__ void proj_resolve_body(const ProjPair& P, const ProjCommon& c, ResolveShared& rs)
{
    extern __shared__ __attribute__((aligned(16))) int32_t tl[];
    int* const hist = rs.hist; int* const sPending = rs.sPending; int* const sInd = rs.sInd; int* const sLive = rs.sLive;
    int& sCount = rs.sCount;
    int32_t* occBy;                        // kFree, -1 (occupied on entry) or the blocking query that took the feature
    if constexpr (TLDS) occBy = tl; else occBy = P.tscr;
    int32_t* minUnd0 = occBy + c.tCap;     // keyed by round (proj_resolve_rounds)
    int32_t* winner = occBy + 2 * c.tCap;  // last query (in query order) that took the feature in this call
    int32_t* qtab = occBy + 3 * c.tCap;
    const int tid = threadIdx.x;
    const int nt = min(P.ntPtr ? *P.ntPtr : P.nt, c.tCap);
    const int nq = min(P.nqPtr ? (P.nq > 0 ? min(*P.nqPtr, P.nq) : *P.nqPtr) : P.nq, kMaxQueryIters * kThreads);
    const int total = (nq > 0 && nt > 0) ? *P.total : 0;
    ORBT_MARK(0);
    for (int t = tid; t < nt; t += kThreads) {
        occBy[t] = (P.toccIn && P.toccIn[t]) ? -1 : kFree;
        minUnd0[t] = 0;   // (no round has posted yet)
        winner[t] = -1;
    }
    if (tid < 32) hist[tid] = 0;
    if (tid < 3) sPending[tid] = 0;
    if (tid == 0) sCount = 0;
    if (total > P.candCap) {  // nothing was listed past the arena; the caller grows it and calls again
        __syncthreads();
        if (tid == 0) { *P.total = 0; *P.nmatch = -total - 1; if (P.stats) { P.stats[0] = 0; P.stats[1] = total; } }
        proj_publish(P);
        return;
    }
    if (TLDS && total <= c.ldsCand && nq <= c.qCap) proj_resolve_rounds<true>(P, c, nq, nt, total, occBy, minUnd0, winner, qtab, hist, sPending, sInd, &sCount, sLive);
    else proj_resolve_rounds<false>(P, c, nq, nt, total, occBy, minUnd0, winner, qtab, hist, sPending, sInd, &sCount, sLive);
}

__global__ __launch_bounds__(kCandThreads) void k_proj_candidates(const ProjPair* __restrict__ pairs, ProjCommon c)
{
    const ProjPair P = pairs[blockIdx.y];
    if (c.big) proj_candidates_body<kCandLanes, false>(P, c, blockIdx.x);
    else if (c.lanes == kCandLanesWide) proj_candidates_body<kCandLanesWide, true>(P, c, blockIdx.x);
    else proj_candidates_body<kCandLanes, true>(P, c, blockIdx.x);
}

__global__ __launch_bounds__(kThreads) void k_proj_resolve(const ProjPair* __restrict__ pairs, ProjCommon c)
{
    const ProjPair P = pairs[blockIdx.x];
    __shared__ ResolveShared rs;
    if (c.big) proj_resolve_body<false>(P, c, rs); else proj_resolve_body<true>(P, c, rs);
}

// The frame-to-frame search over pairs of a frame set's slots: the pair records are made here from the slot numbers
// (kernel arguments), so a call uploads nothing and waits for nothing.
constexpr int kTrackMaxPairs = 128;
struct TrackArgs {
    FrameSetDev fs;
    GridDev grid;
    float th, minX, maxX, minY, maxY;
    uint8_t* occ; int32_t* assign; int32_t* nmatch; int32_t* stats;   // [pair][cap], [pair][cap] (pinned host), [pair], [pair][2]
    int32_t* total; int32_t* candOff; int32_t* candCnt; uint2* cand; int32_t candCap; int32_t* qres; int32_t* qscr; int32_t* tscr;  // [pair], [pair][cap] x2, [pair][candCap], [pair][cap], [pair][3 cap], big: [pair][3 tCap]
    int32_t pair0;
    int32_t* flag; int32_t flagValue;   // [pair] (pinned host) or null
    int16_t cur[kTrackMaxPairs], last[kTrackMaxPairs];
};

__device__ __forceinline__ ProjPair track_pair(const TrackArgs& a, int b)
{
    const int p = a.pair0 + b;
    const int64_t C = a.fs.cap;
    const int cs = a.cur[b], ls = a.last[b];
    ProjPair P;
    P.grid = a.grid;
    P.tkeys = a.fs.keysUn + cs * C; P.cellStart = a.fs.cellStart + (int64_t)cs * (a.fs.ncell + 1); P.trec = a.fs.rec + cs * C;
    P.tdesc = a.fs.desc + cs * C * 32; P.ntPtr = a.fs.n + cs; P.nt = 0;
    P.quvr = nullptr; P.qlvl = nullptr; P.qdesc = a.fs.desc + ls * C * 32; P.qang = nullptr; P.qvalid = nullptr; P.qobs = nullptr;
    P.qur = nullptr; P.turight = nullptr;
    P.qkeys = a.fs.keysUn + ls * C; P.nqPtr = a.fs.n + ls; P.nq = 0;
    P.th = a.th; P.minX = a.minX; P.maxX = a.maxX; P.minY = a.minY; P.maxY = a.maxY;
    P.toccIn = nullptr; P.toccOut = a.occ + p * C;
    P.assign = a.assign + p * C; P.initAssign = 1;
    P.nmatch = a.nmatch + p;
    P.total = a.total + p;
    P.candOff = a.candOff + p * C; P.candCnt = a.candCnt + p * C; P.cand = a.cand + (int64_t)p * a.candCap; P.candCap = a.candCap;
    P.qres = a.qres + p * C; P.qscr = a.qscr + p * C * 3; P.stats = a.stats + 2 * p;
    P.tscr = a.tscr ? a.tscr + (int64_t)p * 3 * (((int64_t)C + 63) & ~(int64_t)63) : nullptr;
    P.flag = a.flag ? a.flag + p : nullptr; P.flagValue = a.flagValue;
    return P;
}

__global__ __launch_bounds__(kCandThreads) void k_track_candidates(TrackArgs a, ProjCommon c)
{
    const ProjPair P = track_pair(a, blockIdx.y);
    if (c.big) proj_candidates_body<kCandLanes, false>(P, c, blockIdx.x);
    else if (c.lanes == kCandLanesWide) proj_candidates_body<kCandLanesWide, true>(P, c, blockIdx.x);
    else proj_candidates_body<kCandLanes, true>(P, c, blockIdx.x);
}

__global__ __launch_bounds__(kThreads) void k_track_resolve(TrackArgs a, ProjCommon c)
{
    const ProjPair P = track_pair(a, blockIdx.x);
    __shared__ ResolveShared rs;
    if (c.big) proj_resolve_body<false>(P, c, rs); else proj_resolve_body<true>(P, c, rs);
}

}  // namespace orbt
