// orbslamm_hip.hip -- C ABI of the MI355X ORB front-end (see include/orbslamm_hip.h).
//
// Host side of the extractor: per-instance tables exactly as the reference
// constructor builds them (/root/reference/SingleRobotScenario/src/ORBextractor.cc:410-470),
// level / cell / quadtree-root geometry (:765-806, :543-563), cv::resize coefficient
// tables, one-time device allocation, kernel launches on the handle's stream.
// No CPU compute path exists: without a HIP device every compute entry fails.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbslamm_hip.h"
#include "orbx_common.hpp"

#include "orbx_kernels.hip"
#include "orbm_kernels.hip"
#include "orbt_kernels.hip"
#include "orbv_kernels.hip"

using namespace orbx;

static_assert(sizeof(OrbxKeyPoint) == 28, "cv::KeyPoint layout");
static_assert(sizeof(OrbxKeyPointDev) == 28, "cv::KeyPoint layout");

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(ORBX_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char* orbx_last_error(void) { return g_err.c_str(); }

// live handles: objects that point at another handle (a frame set at its matcher and at the extractor it last read
// from) check here before touching it in their destructor -- handles may be destroyed in any order
static std::mutex g_liveMu;
static std::vector<const void*> g_live;
static void live_add(const void* h) { std::lock_guard<std::mutex> l(g_liveMu); g_live.push_back(h); }
static void live_remove(const void* h)
{
    std::lock_guard<std::mutex> l(g_liveMu);
    for (size_t i = 0; i < g_live.size(); i++) if (g_live[i] == h) { g_live[i] = g_live.back(); g_live.pop_back(); return; }
}
static bool live_has(const void* h)
{
    std::lock_guard<std::mutex> l(g_liveMu);
    for (const void* p : g_live) if (p == h) return true;
    return false;
}

extern "C" int orbx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// PCI bus id ("0000:c1:00.0") of a device: lets a launcher pin a rank to the GPU's NUMA node (bench.py)
extern "C" int orbx_device_pci_bus_id(int device, char* out, int cap)
{
    if (!out || cap < 16) return fail(ORBX_E_INVALID, "bad argument");
    HIPCHK(hipDeviceGetPCIBusId(out, cap, device));
    return ORBX_OK;
}

// spin-wait hint of the latency path's poll
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#else
    std::this_thread::yield();
#endif
}

static inline int cv_round(double v) { return (int)lrint(v); }  // cvRound: half to even
static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------ profiling
enum ProfId { P_H2D = 0, P_RESIZE, P_FAST, P_DISTRIBUTE, P_BLUR, P_ORIENT_DESC, P_MATCH_BEST2, P_MATCH_ACCEPT, P_MATCH_PRUNE, P_D2H, P_COUNT };
static const char* kProfNames[P_COUNT] = {"h2d", "k_pyramid", "k_fast", "k_distribute", "k_blur",
                                          "k_orient_desc", "k_match_mfma", "k_match_accept", "k_match_prune", "d2h"};
struct ProfSpan { int id; hipEvent_t a, b; };

struct Profiler {
    bool on = false, cur = false;
    int only = -1;  // >= 0: only this kernel's launches are bracketed
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> pool;
    double ms[P_COUNT] = {0};
    int64_t launches[P_COUNT] = {0};
    hipEvent_t get()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    }
    void begin(int id, hipStream_t s)
    {
        cur = on && (only < 0 || only == id);
        if (!cur) return;
        ProfSpan sp{id, get(), get()};
        (void)hipEventRecord(sp.a, s);
        spans.push_back(sp);
    }
    void end(hipStream_t s)
    {
        if (!cur) return;
        (void)hipEventRecord(spans.back().b, s);
    }
    void collect()
    {
        for (auto& sp : spans) {
            (void)hipEventSynchronize(sp.b);
            float t = 0;
            (void)hipEventElapsedTime(&t, sp.a, sp.b);
            ms[sp.id] += t;
            launches[sp.id]++;
            pool.push_back(sp.a);
            pool.push_back(sp.b);
        }
        spans.clear();
    }
    void destroy()
    {
        collect();
        for (auto e : pool) (void)hipEventDestroy(e);
        pool.clear();
    }
};

// ------------------------------------------------------------------ host-side helpers of the host-buffer entries
// A few persistent threads for the bulk memcpys of the host path (pageable frames -> pinned staging, pinned results ->
// caller arrays): 30 MB per 64-frame batch is 3 ms on one core, which alone would cap the path at 20 k frames/s.
struct CopyPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cvWork, cvDone;
    std::function<void(int)> job;
    int nItems = 0, next = 0, pending = 0;
    uint64_t gen = 0;
    bool quit = false;
    void start(int n, int device)
    {
        for (int i = 0; i < n; i++)
            th.emplace_back([this, device] {
                (void)hipSetDevice(device);  // the latency path lets a worker send off the band it has just staged
                uint64_t seen = 0;
                std::unique_lock<std::mutex> lk(m);
                for (;;) {
                    cvWork.wait(lk, [&] { return quit || (gen != seen && next < nItems); });
                    if (quit) return;
                    while (next < nItems) {
                        const int i = next++;
                        lk.unlock();
                        job(i);
                        lk.lock();
                        if (--pending == 0) cvDone.notify_all();
                    }
                    seen = gen;
                }
            });
    }
    // run f(0) .. f(n-1), the caller takes part
    void run(int n, const std::function<void(int)>& f)
    {
        if (th.empty() || n <= 1) { for (int i = 0; i < n; i++) f(i); return; }
        std::unique_lock<std::mutex> lk(m);
        job = f; nItems = n; next = 0; pending = n; gen++;
        cvWork.notify_all();
        while (next < nItems) {
            const int i = next++;
            lk.unlock();
            f(i);
            lk.lock();
            --pending;
        }
        cvDone.wait(lk, [&] { return pending == 0; });
    }
    void stop()
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cvWork.notify_all();
        for (auto& t : th) t.join();
        th.clear();
    }
};

// One batch in flight through the host-buffer entries (orbx_submit_batch .. orbx_release)
struct HostSlot {
    uint8_t* h_in = nullptr;    // pinned staging for pageable caller frames
    uint8_t* d_in = nullptr;    // the batch's frames in HBM (rows 64-byte aligned)
    uint8_t* h_out = nullptr;   // pinned results: [err | n[B] | nmatch[B] | kps[B][maxKp] | desc[B][maxKp][32] | match[B][maxKp]]
    hipEvent_t evUp[4] = {nullptr, nullptr, nullptr, nullptr}, evOut = nullptr;  // evUp[p]: the frames of sub-batch p are in HBM
    int state = 0;              // 0 free, 1 in flight, 2 collected (a view is out)
    bool lat = false;           // results written by k_pack_host: h_out[1] holds ticket + 1 once they are all there
    int ticket = -1, B = 0;
    bool matched = false;
    bool into = false; const int32_t* intoN = nullptr; int intoCap = 0;   // orbx_submit_batch_into: results went to the caller's arrays
};

// ------------------------------------------------------------------ handle
struct orbx_handle {
    OrbxParams prm;
    int device = -1;          // -1: host-only handle (tables, no compute)
    int maxW = 0, maxH = 0, maxB = 0;
    int nlevels = 0;
    float mvScaleFactor[ORBX_MAXL], mvInvScaleFactor[ORBX_MAXL], mvLevelSigma2[ORBX_MAXL], mvInvLevelSigma2[ORBX_MAXL];
    int mnFeaturesPerLevel[ORBX_MAXL];
    int umax[16];

    // geometry of the currently configured frame shape
    Geom geom;
    int curW = 0, curH = 0;
    std::vector<Cell> cells;
    int tileStrideDw = 0, tileRows = 0, fastListCap = 0, tileRows0 = 0, fastListCap0 = 0;
    int nodeCap = 0;
    BlurTiles blurTiles;
    KpBlocks kpBlocks;
    int kpBlocksTotal = 0;
    int pyrBlocks = 0, pyrBufA = 0, pyrBufB = 0, pyrTabCap = 0;
    bool pyrFused = true;
    PyrRange* d_pyrRanges = nullptr; size_t pyrRangesCap = 0;

    hipStream_t stream = nullptr;
    static constexpr int kMaxSplit = 4;
    int nsplit = 2;                           // sub-batches per call (ORBX_SPLIT)
    hipStream_t streamP[kMaxSplit] = {nullptr};  // pipeline stream of sub-batch p (p = 0 uses `stream`)
    hipStream_t streamB[kMaxSplit] = {nullptr};  // blur runs beside FAST + quadtree
    hipEvent_t evStart = nullptr, evPart[kMaxSplit] = {nullptr}, evFast0[kMaxSplit] = {nullptr};
    bool partEverRan[kMaxSplit] = {false};
    int lastParts = 0;                        // evPart[0..lastParts) belong to the last extraction
    int prevB = 0, prevSplit = 0;             // its frame -> sub-batch partition (run_extract: join on a change)
    hipStream_t stream3 = nullptr;            // matching runs beside the next batch's pyramid/FAST
    hipEvent_t evPyr[kMaxSplit] = {nullptr}, evBlur[kMaxSplit] = {nullptr}, evDesc = nullptr, evMatch[2] = {nullptr, nullptr};
    bool matchPending[2] = {false, false};
    // Results (keypoints, descriptors, counts, +-1 descriptors) live in two sets of maxB + 1 slots used by alternate
    // extractions, so that the matching of batch n (set n & 1) never holds back the descriptors of batch n + 1
    int curSet = 0;
    bool serial = false;                      // ORBX_SERIAL=1: everything on one stream (profiling aid)
    hipStream_t matchStream[2] = {nullptr, nullptr}, outStream[2] = {nullptr, nullptr};  // where evMatch[s] / evOutOfSet[s] were recorded
    hipEvent_t evMatched[2] = {nullptr, nullptr};  // the batch's match tables are final (recorded before the roll of the previous frame)
    size_t partialSlots = 0;                  // (frame, chunk) slots of d_partial
    bool matchPopcount = false;               // ORBX_MATCH_POPCOUNT=1: xor/popcount scan instead of the int8 MFMA scan
    bool blurMfma = false;                    // ORBX_BLUR_MFMA=1: the Gaussian as int8 products on the matrix cores (k_blur_mfma) instead of k_blur
    // device buffers (sized for maxW x maxH x maxB at create)
    Geom* d_geom = nullptr;
    Cell* d_cells = nullptr; size_t cellsCap = 0;
    short4* d_tabs = nullptr; size_t tabsCap = 0;
    ResizeTabs tabs;
    size_t imgFrameBytes = 0; int imgStride = 0;  // one frame of the host path's device staging (HostSlot::d_in)
    uint8_t* d_pyr = nullptr; size_t pyrCapFrame = 0;
    uint8_t* d_blur = nullptr; size_t blurCapFrame = 0;
    uint64_t* d_candRaw = nullptr; uint64_t* d_candA = nullptr; uint64_t* d_candB = nullptr; size_t candCapFrame = 0;
    int32_t* d_candCount = nullptr;
    uint32_t* d_distScratch = nullptr; size_t distScratchBytes = 0; bool distInLds = true;
    int32_t* d_cellCount = nullptr;      // [maxB][cellsCap] survivors per FAST cell
    uint64_t* d_kept = nullptr; size_t keptCapFrame = 0;
    int32_t* d_keptCount = nullptr;
    int32_t* d_err = nullptr;            // [0] error flags of device-resident calls, [1] block counter of k_pack_host, [2] scratch word, [4 + slot] error flags of the host-fed batch in that slot
    int32_t* d_errCur = nullptr;         // where the kernels launched right now report (a host-fed batch: its slot's word, consumed and cleared by its own k_pack_host)
    int maxKp = 0;                       // output slot capacity (fixed at create)
    OrbxKeyPointDev* d_kps = nullptr;    // [maxB+1][maxKp]  slot 0 = previous frame of the stream
    uint8_t* d_desc = nullptr;           // [maxB+1][maxKp][32]
    int32_t* d_count = nullptr;          // [maxB+1]
    int32_t* d_match = nullptr;          // [2][maxB][maxKp]  one table per result set
    uint8_t* d_binOf = nullptr;          // [maxB][maxKp]
    int32_t* d_hist = nullptr;           // [maxB][32]
    int32_t* d_nmatch = nullptr;         // [2][maxB]
    uint2* d_partial = nullptr;          // [maxB][kMatchChunks][maxKp] chunk partials of the brute-force scan
    uint8_t* d_xdesc = nullptr;          // [maxB + 1] slots of +-1 byte descriptors in MFMA tile order (k_expand_desc)
    int64_t xPitch = 0;
    // host-buffer entries: kSlots batches in flight (upload of n+1 | kernels of n | download of n-1), allocated at first use
    static constexpr int kSlots = 3;
    HostSlot slot[kSlots];
    bool slotsReady = false;
    int nextTicket = 0;
    size_t outOffN = 0, outOffNm = 0, outOffKp = 0, outOffDesc = 0, outOffMatch = 0, outBytes = 0;
    hipStream_t streamUp = nullptr, streamDown = nullptr;  // upload / results of a batch submitted while nothing else is in flight
    hipStream_t streamUpQ = nullptr, streamDownQ = nullptr;  // the same for a batch submitted behind others: hardware queues of their own
    hipEvent_t evOutOfSet[2] = {nullptr, nullptr};         // the download that last read result set s
    hipEvent_t evExtReader[2] = {nullptr, nullptr};        // a frame set's build that last read result set s (owned by the frame set)
    CopyPool pool;
    int matchSet = 0;                    // result set the last matching wrote (d_match / d_nmatch half)
    void* d_stereo = nullptr; size_t stereoBytes = 0;  // orbx_compute_stereo_matches: uRight | depth | SAD | count
    int lastB = 0;
    FrameSrc lastSrc{};
    bool havePrev = false;
    Profiler prof;
};

static int match_prev_on(orbx_handle* h, hipStream_t s, float nnratio, int th_low, int check_ori, bool roll);
static int roll_prev_on(orbx_handle* h, hipStream_t s, int set);

// ------------------------------------------------------------------ tables, ref :410-470
static int init_tables(orbx_handle* h)
{
    const OrbxParams& p = h->prm;
    if (p.nlevels < 1 || p.nlevels > ORBX_MAXL || p.nfeatures < 1 || !(p.scaleFactor > 1.0f))
        return fail(ORBX_E_INVALID, "bad ORBextractor parameters");
    const int L = p.nlevels;
    h->nlevels = L;
    const double scaleFactor = (double)p.scaleFactor;  // member is double (ORBextractor.h:93)
    h->mvScaleFactor[0] = 1.0f;
    h->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < L; i++) {
        h->mvScaleFactor[i] = (float)((double)h->mvScaleFactor[i - 1] * scaleFactor);
        h->mvLevelSigma2[i] = h->mvScaleFactor[i] * h->mvScaleFactor[i];
    }
    for (int i = 0; i < L; i++) {
        h->mvInvScaleFactor[i] = 1.0f / h->mvScaleFactor[i];
        h->mvInvLevelSigma2[i] = 1.0f / h->mvLevelSigma2[i];
    }
    const float factor = (float)(1.0 / scaleFactor);
    float nDesired = (float)p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
        h->mnFeaturesPerLevel[l] = cv_round(nDesired);
        sum += h->mnFeaturesPerLevel[l];
        nDesired *= factor;
    }
    h->mnFeaturesPerLevel[L - 1] = std::max(p.nfeatures - sum, 0);

    int v, v0;
    const int vmax = (int)std::floor((double)((float)kHalfPatch * std::sqrt(2.f) / 2 + 1));
    const int vmin = (int)std::ceil((double)((float)kHalfPatch * std::sqrt(2.f) / 2));
    const double hp2 = kHalfPatch * kHalfPatch;
    for (v = 0; v < 16; v++) h->umax[v] = 0;
    for (v = 0; v <= vmax; ++v) h->umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
        h->umax[v] = v0;
        ++v0;
    }
    return ORBX_OK;
}

// geometry for a frame shape; fills geom/cells/tables (host side only)
struct HostGeom {
    Geom g;
    std::vector<Cell> cells;
    std::vector<short4> tabs;          // all x/y tables back to back
    int xoff[ORBX_MAXL], yoff[ORBX_MAXL];
    int tileStrideDw, tileRows, fastListCap, tileRows0, fastListCap0, nodeCap;
    BlurTiles bt;
    int blurTilesTotal;
    KpBlocks kb;
    int kbTotal;
    std::vector<PyrRange> pyrRanges;   // [block][level]
    int pyrBlocks, pyrBufA, pyrBufB, pyrTabCap;
    bool pyrFused;
};

static int build_geometry(const orbx_handle* h, int w, int h0, HostGeom& out)
{
    Geom& g = out.g;
    memset(&g, 0, sizeof g);
    g.nlevels = h->nlevels;
    g.w0 = w; g.h0 = h0;
    g.iniTh = h->prm.iniThFAST; g.minTh = h->prm.minThFAST;
    memcpy(g.umax, h->umax, sizeof g.umax);
    if (w > 8191 || h0 > 8191) return fail(ORBX_E_UNSUPPORTED, "frame larger than 8191 px");
    out.cells.clear();
    out.tabs.clear();
    int pyrOff = 0, blurOff = 0, candOff = 0, keptOff = 0, maxRoiW = 8, maxRoiH = 8, maxRoiW0 = 8, maxRoiH0 = 8, nodeCap = 16, maxCells = 1;
    for (int l = 0; l < g.nlevels; l++) {
        LevelGeom& L = g.lv[l];
        const float scale = h->mvInvScaleFactor[l];
        L.w = cv_round((double)((float)w * scale));   // :1111-1112
        L.h = cv_round((double)((float)h0 * scale));
        if (L.w < 1 || L.h < 1) return fail(ORBX_E_UNSUPPORTED, "pyramid level %d is empty", l);
        L.stride = align_up(L.w, 64);
        L.pyrOff = pyrOff;
        if (l > 0) pyrOff += align_up(L.stride * L.h, 256);
        L.blurStride = align_up(L.w, 64);
        L.blurOff = blurOff;
        blurOff += align_up(L.blurStride * L.h, 256);
        L.scale = h->mvScaleFactor[l];
        L.kpSize = (float)(int)((float)kPatchSize * h->mvScaleFactor[l]);  // :837
        L.nFeat = h->mnFeaturesPerLevel[l];

        // FAST window and cell grid, :773-787
        const int minBX = kMinBorder, minBY = kMinBorder;
        const int maxBX = L.w - kEdgeThreshold + 3, maxBY = L.h - kEdgeThreshold + 3;
        L.winW = maxBX - minBX; L.winH = maxBY - minBY;
        const float W = 30;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        L.nCols = width > 0 ? (int)(width / W) : 0;
        L.nRows = height > 0 ? (int)(height / W) : 0;
        L.cellBase = (int)out.cells.size();
        L.nCells = 0;
        int candCap = 0;
        if (L.nCols >= 1 && L.nRows >= 1) {
            L.wCell = (int)std::ceil((double)(width / L.nCols));
            L.hCell = (int)std::ceil((double)(height / L.nRows));
            uint32_t seq = 0;
            for (int i = 0; i < L.nRows; i++) {  // :789-806
                const float iniY = (float)(minBY + i * L.hCell);
                float maxY = iniY + L.hCell + 6;
                if (iniY >= maxBY - 3) continue;
                if (maxY > maxBY) maxY = (float)maxBY;
                for (int j = 0; j < L.nCols; j++) {
                    const float iniX = (float)(minBX + j * L.wCell);
                    float maxX = iniX + L.wCell + 6;
                    if (iniX >= maxBX - 6) continue;
                    if (maxX > maxBX) maxX = (float)maxBX;
                    Cell c;
                    c.level = (uint16_t)l;
                    c.x0 = (uint16_t)(int)iniX; c.y0 = (uint16_t)(int)iniY;
                    c.w = (uint16_t)((int)maxX - (int)iniX); c.h = (uint16_t)((int)maxY - (int)iniY);
                    c.ci = (uint16_t)i; c.cj = (uint16_t)j;
                    c.seq = seq++;
                    c.candOff = (uint32_t)candCap;
                    if (c.w < 7 || c.h < 7) continue;  // cv::FAST finds nothing in such a ROI
                    if (c.w > 127 || c.h > 127 || c.seq >= 65536u)
                        return fail(ORBX_E_UNSUPPORTED, "cell geometry out of range");
                    out.cells.push_back(c);
                    L.nCells++;
                    candCap += ((c.w - 6 + 1) / 2) * ((c.h - 6 + 1) / 2);
                    maxRoiW = std::max<int>(maxRoiW, c.w);
                    maxRoiH = std::max<int>(maxRoiH, c.h);
                    if (l == 0) { maxRoiW0 = std::max<int>(maxRoiW0, c.w); maxRoiH0 = std::max<int>(maxRoiH0, c.h); }
                }
            }
        }
        maxCells = std::max(maxCells, L.nCells);
        L.candOff = candOff;
        L.candCap = align_up(candCap + 8, 8);
        candOff += L.candCap;
        // quadtree roots, :543-545
        L.nIni = 0; L.hX = 0.f;
        if (L.winW > 0 && L.winH > 0) {
            L.nIni = (int)roundf((float)(maxBX - minBX) / (float)(maxBY - minBY));
            if (L.nIni >= 1) L.hX = (float)(maxBX - minBX) / (float)L.nIni;
            else if (L.nCells > 0)
                return fail(ORBX_E_UNSUPPORTED, "level %d: width/height < 0.5, the reference divides by zero (ORBextractor.cc:543-545)", l);
        }
        L.keptOff = keptOff;
        L.keptCap = align_up(std::max(L.nFeat + 4, 4 * L.nIni) + 4, 4);
        keptOff += L.keptCap;
        nodeCap = std::max(nodeCap, L.keptCap + 8);
    }
    g.totalCells = (int)out.cells.size();
    g.maxCellsPerLevel = maxCells;
    g.pyrFrameBytes = std::max(pyrOff, 256);
    g.blurFrameBytes = blurOff;
    g.candFrameRecs = candOff;
    g.keptFrameRecs = keptOff;
    g.maxKp = keptOff;
    out.tileStrideDw = maxRoiW + 5 <= 48 ? 12 : 20;      // k_fast<48> or k_fast<80> (tile row stride in bytes)
    if (maxRoiW + 5 > 80 || maxRoiW - 6 > 127 || maxRoiH - 6 > 127) return fail(ORBX_E_UNSUPPORTED, "cell larger than the FAST tile");
    out.tileRows = maxRoiH;
    out.fastListCap = ((maxRoiW - 6) * (maxRoiH - 6) + 63) / 64 * 64;  // compacted detection pixels
    // the level-0 launch (a third of the cells, all of one size) gets its own, smaller LDS footprint: more waves per CU
    out.tileRows0 = maxRoiH0;
    out.fastListCap0 = ((maxRoiW0 - 6) * (maxRoiH0 - 6) + 63) / 64 * 64;
    out.nodeCap = align_up(std::max(nodeCap, 360), 4);  // k_distribute reads its u32 arrays as uint4, and parks its sort scratch (3201 words) in 9 * cap of them

    // cv::resize INTER_LINEAR coefficient tables, levels >= 1
    for (int l = 0; l < g.nlevels; l++) { out.xoff[l] = out.yoff[l] = 0; }
    for (int l = 1; l < g.nlevels; l++) {
        const int sw = g.lv[l - 1].w, sh = g.lv[l - 1].h, dw = g.lv[l].w, dh = g.lv[l].h;
        const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
        const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
        if (scale_x == 2.0 && scale_y == 2.0)
            return fail(ORBX_E_UNSUPPORTED, "scale factor 2: cv::resize switches to INTER_AREA (not on this path)");
        out.xoff[l] = (int)out.tabs.size();
        std::vector<short4> xt(dw);
        int xmax = dw;
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)std::floor(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx + 1 >= sw) {
                xmax = std::min(xmax, dx);
                if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            }
            auto sat = [](float v) { long r = lrintf(v); return (short)(r > 32767 ? 32767 : (r < -32768 ? -32768 : r)); };
            xt[dx] = make_short4((short)sx, sat((1.f - fx) * 2048), sat(fx * 2048), 0);
        }
        for (int dx = 0; dx < dw; dx++) xt[dx].w = dx < xmax ? 1 : 0;
        out.tabs.insert(out.tabs.end(), xt.begin(), xt.end());
        out.yoff[l] = (int)out.tabs.size();
        for (int dy = 0; dy < dh; dy++) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = (int)std::floor(fy);
            fy -= sy;
            auto clip = [&](int y) { return y < 0 ? 0 : (y < sh ? y : sh - 1); };
            auto sat = [](float v) { long r = lrintf(v); return (short)(r > 32767 ? 32767 : (r < -32768 ? -32768 : r)); };
            out.tabs.push_back(make_short4((short)clip(sy), (short)clip(sy + 1), sat((1.f - fy) * 2048), sat(fy * 2048)));
        }
    }
    // blur tiles (kBlurTW x kBlurTH) and orient/desc blocks (kKpPerBlock keypoints) per level
    int tb = 0, kb = 0;
    for (int l = 0; l < g.nlevels; l++) {
        out.bt.base[l] = tb;
        out.bt.tilesX[l] = (g.lv[l].w + kBlurTW - 1) / kBlurTW;
        tb += out.bt.tilesX[l] * ((g.lv[l].h + kBlurTH - 1) / kBlurTH);
        out.kb.base[l] = kb;
        kb += (g.lv[l].keptCap + kKpPerBlock - 1) / kKpPerBlock;
    }
    for (int l = g.nlevels; l <= ORBX_MAXL; l++) { out.bt.base[l] = tb; out.kb.base[l] = kb; }
    out.blurTilesTotal = tb;
    out.kbTotal = kb;

    // fused pyramid: every block owns the same fractional rectangle of each level;
    // the computed range of level l = owned range + what level l+1's computed range reads.
    // The block grid is refined until the LDS tiles fit; if the halo chain cannot fit at all
    // (scale factors near 2, huge frames) the per-level kernel is used instead.
    // A latency handle (max_batch <= 2) starts one refinement finer: 128 blocks instead of 32 for the one frame in flight
    out.pyrFused = false;
    int refine0 = h->maxB <= 2 ? 1 : 0;
    if (const char* e = getenv("ORBX_PYR_REFINE")) refine0 = std::max(0, std::min(3, atoi(e)));
    for (int refine = refine0; refine < 4 && !out.pyrFused; refine++) {
        const int nl = g.nlevels;
        const int top = nl - 1;
        int BX = 8 << refine, BY = 4 << refine;
        while (BX > 1 && g.lv[top].w / BX < 8) BX >>= 1;
        while (BY > 1 && g.lv[top].h / BY < 8) BY >>= 1;
        out.pyrBlocks = BX * BY;
        out.pyrRanges.assign((size_t)out.pyrBlocks * nl, PyrRange{0, 0, 0, 0, 0, 0, 0, 0});
        int maxA = 4, maxB = 4, tabCap = 4;
        for (int bj = 0; bj < BY; bj++)
            for (int bi = 0; bi < BX; bi++) {
                PyrRange* R = &out.pyrRanges[(size_t)(bj * BX + bi) * nl];
                int nx0 = 0, nx1 = 0, ny0 = 0, ny1 = 0;  // computed range of the level above (empty)
                for (int l = top; l >= 0; l--) {
                    const int w = g.lv[l].w, hh = g.lv[l].h;
                    // x boundaries are multiples of 4 so that every output dword has one owner
                    int ox0 = (int)((int64_t)bi * w / BX) & ~3, ox1 = bi + 1 == BX ? w : ((int)((int64_t)(bi + 1) * w / BX) & ~3);
                    int oy0 = (int)((int64_t)bj * hh / BY), oy1 = (int)((int64_t)(bj + 1) * hh / BY);
                    if (l == 0) ox0 = ox1 = oy0 = oy1 = 0;  // level 0 is the caller's frame: nothing to write
                    int cx0 = ox0, cx1 = ox1, cy0 = oy0, cy1 = oy1;
                    if (l < top && nx1 > nx0 && ny1 > ny0) {
                        const short4* xt = &out.tabs[out.xoff[l + 1]];
                        const short4* yt = &out.tabs[out.yoff[l + 1]];
                        int sx0 = 1 << 30, sx1 = -1, sy0 = 1 << 30, sy1 = -1;
                        // the kernel computes whole dword groups: cover the rounded-up range
                        const int nx1g = std::min<int>(g.lv[l + 1].w, nx0 + ((nx1 - nx0 + 3) & ~3));
                        for (int dx = nx0; dx < nx1g; dx++) {
                            const int a = (uint16_t)xt[dx].x, b = a + (xt[dx].w ? 2 : 1);
                            sx0 = std::min(sx0, a); sx1 = std::max(sx1, b);
                        }
                        for (int dy = ny0; dy < ny1; dy++) {
                            sy0 = std::min<int>(sy0, yt[dy].x); sy1 = std::max<int>(sy1, yt[dy].y + 1);
                        }
                        sx1 = std::min(sx1, w); sy1 = std::min(sy1, hh);
                        if (cx1 > cx0 && cy1 > cy0) {
                            cx0 = std::min(cx0, sx0); cx1 = std::max(cx1, sx1);
                            cy0 = std::min(cy0, sy0); cy1 = std::max(cy1, sy1);
                        } else { cx0 = sx0; cx1 = sx1; cy0 = sy0; cy1 = sy1; }
                    }
                    cx0 &= ~3;  // dword-aligned tile origin
                    R[l] = PyrRange{(int16_t)ox0, (int16_t)ox1, (int16_t)oy0, (int16_t)oy1,
                                    (int16_t)cx0, (int16_t)cx1, (int16_t)cy0, (int16_t)cy1};
                    if (cx1 > cx0 && cy1 > cy0) {
                        const int rowBytes = (cx1 - cx0 + 3) & ~3;
                        int rows = cy1 - cy0;
                        if (l == 0 && ny1 > ny0) {
                            // the kernel stages level 0 in kPyrStrips strips: the rows the strip's level-1 rows read
                            // (same split as k_pyramid: rows [chh * s / n, chh * (s + 1) / n) of level 1's computed range)
                            const short4* yt = &out.tabs[out.yoff[1]];
                            const int chh1 = ny1 - ny0;
                            rows = 0;
                            for (int sidx = 0; sidx < kPyrStrips; sidx++) {
                                const int ya = (int)((int64_t)chh1 * sidx / kPyrStrips), yb = (int)((int64_t)chh1 * (sidx + 1) / kPyrStrips);
                                if (yb > ya) rows = std::max(rows, std::min<int>(yt[ny0 + yb - 1].y + 1, cy1) - (int)yt[ny0 + ya].x);
                            }
                        }
                        const int words = rowBytes / 4 * rows + 4;
                        if (l & 1) maxB = std::max(maxB, words); else maxA = std::max(maxA, words);
                        if (l > 0) tabCap = std::max(tabCap, std::max(rowBytes, cy1 - cy0));
                    }
                    nx0 = cx0; nx1 = cx1; ny0 = cy0; ny1 = cy1;
                }
            }
        out.pyrBufA = maxA; out.pyrBufB = maxB; out.pyrTabCap = (tabCap + 3) & ~3;
        if (getenv("ORBX_DEBUG_GEOM")) fprintf(stderr, "pyramid: %d blocks, bufA %d B, bufB %d B, tabs %d B\n", out.pyrBlocks, maxA * 4, maxB * 4, out.pyrTabCap * 16);
        const size_t pl = ((size_t)maxA + maxB) * 4 + (size_t)out.pyrTabCap * 16;
        out.pyrFused = pl <= 64 * 1024 || (refine == 3 && pl <= 156 * 1024);
    }
    return ORBX_OK;
}

static constexpr int kMatchChunks = 4;  // train chunks per query block (wave count x4)

static size_t dist_lds_bytes(int cap, int maxCells) { return (size_t)(19 * cap + 8 + 2 * (maxCells + 1)) * 4; }

// ------------------------------------------------------------------ create / destroy
static void free_device(orbx_handle* h)
{
    if (h->device < 0) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int i = 0; i < orbx_handle::kMaxSplit; i++) {
        if (h->streamP[i]) (void)hipStreamSynchronize(h->streamP[i]);
        if (h->streamB[i]) (void)hipStreamSynchronize(h->streamB[i]);
    }
    if (h->stream3) (void)hipStreamSynchronize(h->stream3);
    h->prof.destroy();
    void* ptrs[] = {h->d_distScratch, h->d_pyrRanges, h->d_geom, h->d_cells, h->d_tabs, h->d_pyr, h->d_blur, h->d_candRaw, h->d_candA, h->d_candB,
                    h->d_candCount, h->d_cellCount, h->d_kept, h->d_keptCount, h->d_err, h->d_kps, h->d_desc, h->d_count,
                    h->d_match, h->d_binOf, h->d_hist, h->d_nmatch, h->d_partial, h->d_xdesc};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->d_stereo) (void)hipFree(h->d_stereo);
    h->pool.stop();
    for (hipStream_t st : {h->streamUp, h->streamDown, h->streamUpQ, h->streamDownQ}) if (st) (void)hipStreamSynchronize(st);
    for (auto& sl : h->slot) {
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_in) (void)hipFree(sl.d_in);
        for (hipEvent_t& e : sl.evUp) if (e) (void)hipEventDestroy(e);
        if (sl.evOut) (void)hipEventDestroy(sl.evOut);
    }
    for (hipStream_t st : {h->streamUp, h->streamDown, h->streamUpQ, h->streamDownQ}) if (st) (void)hipStreamDestroy(st);
    for (int i = 0; i < orbx_handle::kMaxSplit; i++) {
        if (h->evPyr[i]) (void)hipEventDestroy(h->evPyr[i]);
        if (h->evBlur[i]) (void)hipEventDestroy(h->evBlur[i]);
        if (h->evPart[i]) (void)hipEventDestroy(h->evPart[i]);
        if (h->evFast0[i]) (void)hipEventDestroy(h->evFast0[i]);
        if (h->streamP[i]) (void)hipStreamDestroy(h->streamP[i]);
        if (h->streamB[i]) (void)hipStreamDestroy(h->streamB[i]);
    }
    if (h->evStart) (void)hipEventDestroy(h->evStart);
    if (h->evDesc) (void)hipEventDestroy(h->evDesc);
    for (int i = 0; i < 2; i++) if (h->evMatch[i]) (void)hipEventDestroy(h->evMatch[i]);
    for (int i = 0; i < 2; i++) if (h->evMatched[i]) (void)hipEventDestroy(h->evMatched[i]);
    if (h->stream3) (void)hipStreamDestroy(h->stream3);
    if (h->stream) (void)hipStreamDestroy(h->stream);
}

extern "C" int orbx_create(const OrbxParams* params, int max_w, int max_h, int max_batch, int device, orbx_t** out)
{
    if (!params || !out) return fail(ORBX_E_INVALID, "null argument");
    *out = nullptr;
    orbx_handle* h = new orbx_handle();
    h->prm = *params;
    int rc = init_tables(h);
    if (rc) { delete h; return rc; }
    h->device = device;
    { const char* e = getenv("ORBX_SERIAL"); h->serial = e && e[0] == '1'; }
    { const char* e = getenv("ORBX_MATCH_POPCOUNT"); h->matchPopcount = e && e[0] == '1'; }
    // the matrix-core form of the Gaussian (k_blur_mfma): same bytes, faster alone, slower beside k_match_mfma
    { const char* e = getenv("ORBX_BLUR_MFMA"); h->blurMfma = e && e[0] == '1'; }
    { const char* e = getenv("ORBX_SPLIT"); if (e && e[0] >= '1' && e[0] <= '4') h->nsplit = e[0] - '0'; }
    h->maxW = max_w; h->maxH = max_h; h->maxB = max_batch;
    if (device < 0) { *out = h; return ORBX_OK; }  // host-only handle: tables and geometry queries
    if (max_w < 1 || max_h < 1 || max_batch < 1) { delete h; return fail(ORBX_E_INVALID, "bad maximum shape"); }
    int ndev = orbx_device_count();
    if (ndev == 0) { delete h; return fail(ORBX_E_NO_DEVICE, "no HIP device visible: the ORB front-end has no CPU fallback"); }
    if (device >= ndev) { delete h; return fail(ORBX_E_INVALID, "device %d out of range (%d visible)", device, ndev); }

    HostGeom hg;
    rc = build_geometry(h, max_w, max_h, hg);
    if (rc) { delete h; return rc; }
#define CRT(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { int r_ = fail(ORBX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); free_device(h); delete h; return r_; } } while (0)
    CRT(hipSetDevice(device));
    // The first four streams of a process get a hardware queue each, later ones share the last (observed with
    // rocprofv3 --kernel-trace), and two busy streams on one queue serialise.  The steady-state pipeline therefore
    // uses exactly four: the host-facing stream (uploads, downloads -- idle while a device-resident stream runs --
    // and the blur kernels), sub-batch 0, sub-batch 1, matching.
    // The sub-batch streams carry the chain pyramid -> FAST -> quadtree -> descriptors whose length IS the step; the blur /
    // level-0 FAST stream and the matcher have slack.  Queue priority (which queue's workgroups the dispatcher places
    // first) for the chain: 142.6 k -> 144.0 k frames/s; raising the matcher instead: 138.2 k, the blur stream: 140.6 k.
    int prLo = 0, prHi = 0;
    CRT(hipDeviceGetStreamPriorityRange(&prLo, &prHi));
    CRT(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CRT(hipStreamCreateWithPriority(&h->streamP[0], hipStreamNonBlocking, prHi));
    CRT(hipStreamCreateWithPriority(&h->streamP[1], hipStreamNonBlocking, prHi));
    CRT(hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking));
    // hardware queues are bound at a stream's first use: use the four once, now, in this order
    {
        void* scratch = nullptr;
        CRT(hipMalloc(&scratch, 256));
        hipStream_t four[4] = {h->stream, h->streamP[0], h->streamP[1], h->stream3};
        for (hipStream_t st : four) { CRT(hipMemsetAsync(scratch, 0, 256, st)); CRT(hipStreamSynchronize(st)); }
        CRT(hipFree(scratch));
    }
    for (int i = 0; i < orbx_handle::kMaxSplit; i++) {
        if (i > 1) CRT(hipStreamCreateWithPriority(&h->streamP[i], hipStreamNonBlocking, prHi));
        CRT(hipEventCreateWithFlags(&h->evPyr[i], hipEventDisableTiming));
        CRT(hipEventCreateWithFlags(&h->evBlur[i], hipEventDisableTiming));
        CRT(hipEventCreateWithFlags(&h->evPart[i], hipEventDisableTiming));
        CRT(hipEventCreateWithFlags(&h->evFast0[i], hipEventDisableTiming));
    }
    CRT(hipEventCreateWithFlags(&h->evStart, hipEventDisableTiming));
    CRT(hipEventCreateWithFlags(&h->evDesc, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) CRT(hipEventCreateWithFlags(&h->evMatch[i], hipEventDisableTiming));
    for (int i = 0; i < 2; i++) CRT(hipEventCreateWithFlags(&h->evMatched[i], hipEventDisableTiming));
    const size_t B = (size_t)max_batch;
    // capacities with head-room so that smaller shapes (different cell layouts) also fit
    h->cellsCap = hg.cells.size() * 2 + 64;
    h->tabsCap = hg.tabs.size() * 2 + 64;
    h->imgStride = align_up(max_w, 64);
    h->imgFrameBytes = (size_t)h->imgStride * max_h;
    h->pyrCapFrame = (size_t)hg.g.pyrFrameBytes + 4096;
    h->blurCapFrame = (size_t)hg.g.blurFrameBytes + 4096;
    h->candCapFrame = (size_t)hg.g.candFrameRecs + 1024;
    h->keptCapFrame = (size_t)hg.g.keptFrameRecs + 64;
    h->maxKp = hg.g.maxKp + 64;
    CRT(hipMalloc(&h->d_geom, sizeof(Geom)));
    CRT(hipMalloc(&h->d_cells, h->cellsCap * sizeof(Cell)));
    CRT(hipMalloc(&h->d_tabs, h->tabsCap * sizeof(short4)));
    h->pyrRangesCap = 64 * ORBX_MAXL;
    CRT(hipMalloc(&h->d_pyrRanges, h->pyrRangesCap * sizeof(PyrRange)));
    CRT(hipMalloc(&h->d_pyr, h->pyrCapFrame * B));
    CRT(hipMalloc(&h->d_blur, h->blurCapFrame * B));
    CRT(hipMalloc(&h->d_candRaw, h->candCapFrame * B * sizeof(uint64_t)));
    CRT(hipMalloc(&h->d_candA, h->candCapFrame * B * sizeof(uint64_t)));
    CRT(hipMalloc(&h->d_candB, h->candCapFrame * B * sizeof(uint64_t)));
    CRT(hipMalloc(&h->d_candCount, B * ORBX_MAXL * sizeof(int32_t)));
    CRT(hipMalloc(&h->d_cellCount, B * h->cellsCap * sizeof(int32_t)));
    CRT(hipMalloc(&h->d_kept, h->keptCapFrame * B * sizeof(uint64_t)));
    CRT(hipMalloc(&h->d_keptCount, B * ORBX_MAXL * sizeof(int32_t)));
    CRT(hipMalloc(&h->d_err, 8 * sizeof(int32_t)));
    CRT(hipMalloc(&h->d_kps, 2 * (B + 1) * h->maxKp * sizeof(OrbxKeyPointDev)));
    CRT(hipMalloc(&h->d_desc, 2 * (B + 1) * (size_t)h->maxKp * 32));
    CRT(hipMalloc(&h->d_count, 2 * (B + 1) * sizeof(int32_t)));
    CRT(hipMalloc(&h->d_match, 2 * B * h->maxKp * sizeof(int32_t)));
    CRT(hipMalloc(&h->d_binOf, B * (size_t)h->maxKp));
    CRT(hipMalloc(&h->d_hist, B * 32 * sizeof(int32_t)));
    CRT(hipMalloc(&h->d_nmatch, 2 * B * sizeof(int32_t)));
    h->partialSlots = std::max<size_t>(B * kMatchChunks, 16);  // few frames: up to 8 train chunks per frame (k_match_mfma)
    CRT(hipMalloc(&h->d_partial, h->partialSlots * h->maxKp * sizeof(uint2)));
    h->xPitch = (int64_t)align_up(h->maxKp, orbm::kMfmaRowsPerBlock) * 256;
    CRT(hipMalloc(&h->d_xdesc, 2 * (B + 1) * (size_t)h->xPitch));
    CRT(hipMemset(h->d_xdesc, 0, 2 * (B + 1) * (size_t)h->xPitch));
    CRT(hipMemset(h->d_err, 0, 8 * sizeof(int32_t)));
    h->d_errCur = h->d_err;
    CRT(hipMemset(h->d_count, 0, 2 * (B + 1) * sizeof(int32_t)));
    CRT(hipMemset(h->d_hist, 0, B * 32 * sizeof(int32_t)));
    CRT(hipMemset(h->d_nmatch, 0, 2 * B * sizeof(int32_t)));
#undef CRT
    live_add(h);
    *out = h;
    return ORBX_OK;
}

extern "C" void orbx_destroy(orbx_t* h)
{
    if (!h) return;
    live_remove(h);
    free_device(h);
    delete h;
}

extern "C" int orbx_levels(const orbx_t* h) { return h ? h->nlevels : 0; }
extern "C" float orbx_scale_factor(const orbx_t* h) { return h ? (float)(double)h->prm.scaleFactor : 0.f; }
extern "C" int orbx_scale_tables(const orbx_t* h, float* s, float* is, float* s2, float* is2)
{
    if (!h) return fail(ORBX_E_INVALID, "null handle");
    for (int i = 0; i < h->nlevels; i++) {
        if (s) s[i] = h->mvScaleFactor[i];
        if (is) is[i] = h->mvInvScaleFactor[i];
        if (s2) s2[i] = h->mvLevelSigma2[i];
        if (is2) is2[i] = h->mvInvLevelSigma2[i];
    }
    return ORBX_OK;
}
extern "C" int orbx_features_per_level(const orbx_t* h, int32_t* out)
{
    if (!h || !out) return fail(ORBX_E_INVALID, "null argument");
    for (int i = 0; i < h->nlevels; i++) out[i] = h->mnFeaturesPerLevel[i];
    return ORBX_OK;
}
extern "C" int orbx_umax(const orbx_t* h, int32_t out[16])
{
    if (!h || !out) return fail(ORBX_E_INVALID, "null argument");
    for (int i = 0; i < 16; i++) out[i] = h->umax[i];
    return ORBX_OK;
}
extern "C" int orbx_max_keypoints(const orbx_t* h)
{
    if (!h) return 0;
    if (h->device >= 0) return h->maxKp;
    HostGeom hg;
    if (h->maxW < 1 || h->maxH < 1 || build_geometry(h, h->maxW, h->maxH, hg)) return 0;
    return hg.g.maxKp + 64;
}

// ------------------------------------------------------------------ result sets
static inline size_t set_slot0(const orbx_handle* h, int set) { return (size_t)set * ((size_t)h->maxB + 1); }
static inline OrbxKeyPointDev* r_kps(orbx_handle* h, int set) { return h->d_kps + set_slot0(h, set) * h->maxKp; }
static inline uint8_t* r_desc(orbx_handle* h, int set) { return h->d_desc + set_slot0(h, set) * h->maxKp * 32; }
static inline int32_t* r_count(orbx_handle* h, int set) { return h->d_count + set_slot0(h, set); }
static inline uint8_t* r_xdesc(orbx_handle* h, int set) { return h->d_xdesc + set_slot0(h, set) * (size_t)h->xPitch; }

// ------------------------------------------------------------------ shape configuration
static int sync_all(orbx_handle* h)
{
    HIPCHK(hipStreamSynchronize(h->stream));
    for (int i = 0; i < orbx_handle::kMaxSplit; i++) {
        if (h->streamP[i]) HIPCHK(hipStreamSynchronize(h->streamP[i]));
        if (h->streamB[i]) HIPCHK(hipStreamSynchronize(h->streamB[i]));
    }
    HIPCHK(hipStreamSynchronize(h->stream3));
    for (hipStream_t st : {h->streamUp, h->streamDown, h->streamUpQ, h->streamDownQ}) if (st) HIPCHK(hipStreamSynchronize(st));
    h->matchPending[0] = h->matchPending[1] = false;
    h->evOutOfSet[0] = h->evOutOfSet[1] = nullptr;
    return ORBX_OK;
}

static int configure_shape(orbx_handle* h, int w, int hh)
{
    if (h->curW == w && h->curH == hh) return ORBX_OK;
    if (w > h->maxW || hh > h->maxH) return fail(ORBX_E_INVALID, "frame %dx%d exceeds the handle's maximum %dx%d", w, hh, h->maxW, h->maxH);
    HostGeom hg;
    int rc = build_geometry(h, w, hh, hg);
    if (rc) return rc;
    if (hg.cells.size() > h->cellsCap || hg.tabs.size() > h->tabsCap || (size_t)hg.g.pyrFrameBytes > h->pyrCapFrame ||
        (size_t)hg.g.blurFrameBytes > h->blurCapFrame || (size_t)hg.g.candFrameRecs > h->candCapFrame ||
        (size_t)hg.g.keptFrameRecs > h->keptCapFrame || hg.g.maxKp > h->maxKp)
        return fail(ORBX_E_INVALID, "frame %dx%d needs more scratch than the handle was created with", w, hh);
    h->distInLds = dist_lds_bytes(hg.nodeCap, hg.g.maxCellsPerLevel) <= 156 * 1024;
    if (h->distInLds) {
        if (dist_lds_bytes(hg.nodeCap, hg.g.maxCellsPerLevel) > 48 * 1024)
            HIPCHK(hipFuncSetAttribute((const void*)k_distribute<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dist_lds_bytes(hg.nodeCap, hg.g.maxCellsPerLevel)));
    } else {
        // per-level target too large for LDS: node list in a global scratch region per (frame, level)
        const size_t need = dist_lds_bytes(hg.nodeCap, hg.g.maxCellsPerLevel) * (size_t)h->maxB * hg.g.nlevels;
        if (need > h->distScratchBytes) {
            if ((rc = sync_all(h))) return rc;
            if (h->d_distScratch) HIPCHK(hipFree(h->d_distScratch));
            h->d_distScratch = nullptr; h->distScratchBytes = 0;
            HIPCHK(hipMalloc(&h->d_distScratch, need));
            h->distScratchBytes = need;
        }
    }
    hg.g.maxKp = h->maxKp;  // output slots keep their create-time pitch
    if ((rc = sync_all(h))) return rc;
    HIPCHK(hipMemcpy(h->d_geom, &hg.g, sizeof(Geom), hipMemcpyHostToDevice));
    if (!hg.cells.empty()) HIPCHK(hipMemcpy(h->d_cells, hg.cells.data(), hg.cells.size() * sizeof(Cell), hipMemcpyHostToDevice));
    if (!hg.tabs.empty()) HIPCHK(hipMemcpy(h->d_tabs, hg.tabs.data(), hg.tabs.size() * sizeof(short4), hipMemcpyHostToDevice));
    h->pyrFused = hg.pyrFused;
    if (hg.pyrFused) {
        if (hg.pyrRanges.size() > h->pyrRangesCap) {
            if (h->d_pyrRanges) HIPCHK(hipFree(h->d_pyrRanges));
            h->d_pyrRanges = nullptr; h->pyrRangesCap = 0;
            HIPCHK(hipMalloc(&h->d_pyrRanges, hg.pyrRanges.size() * sizeof(PyrRange)));
            h->pyrRangesCap = hg.pyrRanges.size();
        }
        HIPCHK(hipMemcpy(h->d_pyrRanges, hg.pyrRanges.data(), hg.pyrRanges.size() * sizeof(PyrRange), hipMemcpyHostToDevice));
        h->pyrBlocks = hg.pyrBlocks; h->pyrBufA = hg.pyrBufA; h->pyrBufB = hg.pyrBufB; h->pyrTabCap = hg.pyrTabCap;
        const size_t pl = ((size_t)h->pyrBufA + h->pyrBufB) * 4 + (size_t)h->pyrTabCap * 16;
        if (pl > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_pyramid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl));
    }
    for (int l = 0; l < ORBX_MAXL; l++) {
        h->tabs.xtab[l] = h->d_tabs + (l < hg.g.nlevels ? hg.xoff[l] : 0);
        h->tabs.ytab[l] = h->d_tabs + (l < hg.g.nlevels ? hg.yoff[l] : 0);
    }
    h->geom = hg.g;
    h->cells = hg.cells;
    h->tileStrideDw = hg.tileStrideDw; h->tileRows = hg.tileRows; h->fastListCap = hg.fastListCap; h->nodeCap = hg.nodeCap;
    h->tileRows0 = hg.tileRows0; h->fastListCap0 = hg.fastListCap0;
    h->blurTiles = hg.bt; h->kpBlocks = hg.kb; h->kpBlocksTotal = hg.kbTotal;
    h->geom.totalCells = hg.g.totalCells;
    h->curW = w; h->curH = hh;
    // a new shape starts a new stream
    h->havePrev = false;
    for (int set = 0; set < 2; set++) HIPCHK(hipMemset(r_count(h, set), 0, sizeof(int32_t)));
    return ORBX_OK;
}

static int check_device(orbx_handle* h)
{
    if (!h) return fail(ORBX_E_INVALID, "null handle");
    if (h->device < 0) return fail(ORBX_E_NO_DEVICE, "host-only handle: no HIP device bound, and there is no CPU fallback");
    HIPCHK(hipSetDevice(h->device));
    return ORBX_OK;
}

// ------------------------------------------------------------------ the pipeline
// grid.y of the launches that map (block, frame) through xcd_block_frame
static inline unsigned xcd_grid_y(int nb) { return nb < 8 ? (unsigned)nb : 8u * (unsigned)((nb + 7) / 8); }

// make `s` wait for every sub-batch of the last extraction
static int join_parts(orbx_handle* h, hipStream_t s)
{
    for (int p = 0; p < h->lastParts; p++) HIPCHK(hipStreamWaitEvent(s, h->evPart[p], 0));
    return ORBX_OK;
}

// kernel launches of one (sub-)batch: frames [f0, f0 + nb) of `src`
struct Launcher {
    orbx_handle* h;
    FrameSrc src;   // src.f0 = first frame
    int nb;
    void fast(hipStream_t fs, int cell0, int ncells) const
    {
        if (ncells <= 0) return;
        const Geom& g = h->geom;
        const bool l0 = cell0 == 0 && ncells == g.lv[0].nCells;  // the level-0 launch has its own, smaller LDS footprint
        const int rows = l0 ? h->tileRows0 : h->tileRows, cap = l0 ? h->fastListCap0 : h->fastListCap;
        const size_t lds = (size_t)2 * (rows * h->tileStrideDw + 4) * 4 + (size_t)cap * 2;
        h->prof.begin(P_FAST, fs);
        if (h->tileStrideDw == 12)
            hipLaunchKernelGGL(k_fast<48>, dim3(ncells, xcd_grid_y(nb)), dim3(64), lds, fs, h->d_geom, h->d_cells, src, h->d_candRaw,
                               h->d_cellCount, h->d_errCur, rows, cap, nb, cell0);
        else
            hipLaunchKernelGGL(k_fast<80>, dim3(ncells, xcd_grid_y(nb)), dim3(64), lds, fs, h->d_geom, h->d_cells, src, h->d_candRaw,
                               h->d_cellCount, h->d_errCur, rows, cap, nb, cell0);
        h->prof.end(fs);
    }
    void dist(hipStream_t ds, int l0, int nl) const  // quadtree of levels [l0, l0 + nl)
    {
        if (nl <= 0) return;
        const Geom& g = h->geom;
        h->prof.begin(P_DISTRIBUTE, ds);
        const size_t dl = dist_lds_bytes(h->nodeCap, g.maxCellsPerLevel);
        if (h->distInLds)
            hipLaunchKernelGGL(k_distribute<true>, dim3(nl, nb), dim3(kDistThreads), dl, ds, h->d_geom, h->d_candRaw,
                               h->d_candA, h->d_candB, h->d_cells, h->d_cellCount, h->d_candCount, h->d_kept, h->d_keptCount,
                               h->d_errCur, h->nodeCap, src.f0, (uint32_t*)nullptr, 0, l0);
        else
            hipLaunchKernelGGL(k_distribute<false>, dim3(nl, nb), dim3(kDistThreads), 0, ds, h->d_geom, h->d_candRaw,
                               h->d_candA, h->d_candB, h->d_cells, h->d_cellCount, h->d_candCount, h->d_kept, h->d_keptCount,
                               h->d_errCur, h->nodeCap, src.f0, h->d_distScratch + (size_t)src.f0 * g.nlevels * (dl / 4), (int)(dl / 4), l0);
        h->prof.end(ds);
    }
    // done != nullptr: the event rides on the kernel's own dispatch packet (hipExtLaunchKernelGGL) -- a separate
    // hipEventRecord between two kernels of a stream costs ~6 us of gap (tools/b1_timeline.sh), which only matters
    // where the chain of kernels IS the latency of a call
    int pyramid(hipStream_t s, hipEvent_t done = nullptr) const
    {
        const Geom& g = h->geom;
        if (g.nlevels > 1 && h->pyrFused) {
            const size_t pl = ((size_t)h->pyrBufA + h->pyrBufB) * 4 + (size_t)h->pyrTabCap * 16;
            h->prof.begin(P_RESIZE, s);
            if (done && !h->prof.cur)
                hipExtLaunchKernelGGL(k_pyramid, dim3(h->pyrBlocks, nb), dim3(256), pl, s, nullptr, done, 0, (const Geom*)h->d_geom, src, h->tabs,
                                      (const PyrRange*)h->d_pyrRanges, h->pyrBufA, h->pyrBufB, h->pyrTabCap);
            else
                hipLaunchKernelGGL(k_pyramid, dim3(h->pyrBlocks, nb), dim3(256), pl, s, h->d_geom, src, h->tabs,
                                   (const PyrRange*)h->d_pyrRanges, h->pyrBufA, h->pyrBufB, h->pyrTabCap);
            h->prof.end(s);
            if (done && h->prof.cur) HIPCHK(hipEventRecord(done, s));
        } else {
            for (int l = 1; l < g.nlevels; l++) {
                h->prof.begin(P_RESIZE, s);
                hipLaunchKernelGGL(k_resize_level, dim3((g.lv[l].w + 255) / 256, (g.lv[l].h + 3) / 4, nb), dim3(64, 4, 1), 0, s,
                                   h->d_geom, src, h->tabs, l);
                h->prof.end(s);
            }
            if (done) HIPCHK(hipEventRecord(done, s));
        }
        return ORBX_OK;
    }
    void blur(hipStream_t s) const
    {
        h->prof.begin(P_BLUR, s);
        if (h->blurMfma)
            hipLaunchKernelGGL(k_blur_mfma, dim3(h->blurTiles.base[h->geom.nlevels], xcd_grid_y(nb)), dim3(256), 0, s, h->d_geom, src, h->blurTiles, nb);
        else
            hipLaunchKernelGGL(k_blur, dim3(h->blurTiles.base[h->geom.nlevels], xcd_grid_y(nb)), dim3(256), 0, s, h->d_geom, src, h->blurTiles, nb);
        h->prof.end(s);
    }
    // The output slots of `set` were read by the matching two batches back and by its download (host path); a wait is
    // only enqueued when that work sits on another stream (every cross-stream wait costs microseconds of latency).
    int desc(hipStream_t s, int set, hipEvent_t done = nullptr) const
    {
        if (h->matchPending[set] && h->matchStream[set] != s) HIPCHK(hipStreamWaitEvent(s, h->evMatch[set], 0));
        if (h->evOutOfSet[set] && h->outStream[set] != s) HIPCHK(hipStreamWaitEvent(s, h->evOutOfSet[set], 0));
        if (h->evExtReader[set]) HIPCHK(hipStreamWaitEvent(s, h->evExtReader[set], 0));
        h->prof.begin(P_ORIENT_DESC, s);
        if (done && !h->prof.cur)
            hipExtLaunchKernelGGL(k_orient_desc, dim3(h->kpBlocksTotal, xcd_grid_y(nb)), dim3(256), 0, s, nullptr, done, 0, (const Geom*)h->d_geom, src, h->kpBlocks,
                                  (const uint64_t*)h->d_kept, (const int32_t*)h->d_keptCount, r_kps(h, set) + h->maxKp, r_desc(h, set) + (size_t)h->maxKp * 32,
                                  r_count(h, set) + 1, nb);
        else
            hipLaunchKernelGGL(k_orient_desc, dim3(h->kpBlocksTotal, xcd_grid_y(nb)), dim3(256), 0, s, h->d_geom, src, h->kpBlocks, h->d_kept,
                               h->d_keptCount, r_kps(h, set) + h->maxKp, r_desc(h, set) + (size_t)h->maxKp * 32, r_count(h, set) + 1, nb);
        h->prof.end(s);
        if (done && h->prof.cur) HIPCHK(hipEventRecord(done, s));
        return ORBX_OK;
    }
};

// Two event graphs.
// THROUGHPUT (B > 2): the batch is cut into sub-batches that run on separate stream groups: the latency-bound kernels
// of one sub-batch (quadtree, descriptors) overlap the throughput-bound ones (FAST, matching) of the other.  Level-0
// FAST runs beside the pyramid on the blur stream; the quadtree of all levels follows FAST on the sub-batch's stream.
// (Moving the level-0 quadtree ahead -- right behind the level-0 FAST, or behind the blur -- was measured at 64 frames
// per step: 128.3 k -> 118.8 k and 111.2 k frames/s, two A/B rounds each: on the blur stream it delays the blur and
// with it the descriptors.  It stays where it was.)
// LATENCY (one or two frames, the per-frame drop-in entry): nothing else keeps the GPU busy, the chain IS the call.
//   main (streamP[0]):  [frames arrive here on the host path] pyramid -> FAST 1.. -> quadtree 1.. -> descriptors
//   aux  (stream):      FAST 0 -> quadtree 0 -> blur
// one cross-stream wait in front of the descriptors; the level-0 quadtree (as long as levels 1.. together: one
// workgroup per level) runs beside pyramid + FAST instead of behind them.
// sIn = the stream on which the frames become available (nullptr: the host-facing stream, where the device-resident
// entry has always taken them from).
static int run_extract(orbx_handle* h, const uint8_t* d_imgs, int B, int w, int hh, int stride, size_t pitch,
                       const hipEvent_t* evUploaded = nullptr, hipStream_t sIn = nullptr)
{
    int rc = configure_shape(h, w, hh);
    if (rc) return rc;
    if (B < 1 || B > h->maxB) return fail(ORBX_E_INVALID, "batch %d outside [1,%d]", B, h->maxB);
    if (((uintptr_t)d_imgs & 3) || (stride & 3) || (pitch & 3) || stride < w)
        return fail(ORBX_E_INVALID, "device frames need 4-byte aligned base/stride/pitch and stride >= width");
    const Geom& g = h->geom;
    FrameSrc src;
    src.img0 = d_imgs; src.stride0 = stride; src.pitch0 = (int64_t)pitch;
    src.pyr = h->d_pyr; src.blur = h->d_blur; src.f0 = 0;
    // the pyramid/blur buffers use the geometry's per-frame sizes as pitch
    hipStream_t s0 = h->stream;
    if (!sIn) sIn = s0;
    const int nsplit = h->serial || B <= 2 ? 1 : std::min(h->nsplit, B);
    const bool lat = !h->serial && B <= 2 && g.nlevels > 1;
    // Sub-batch p owns stream streamP[p] across calls: it follows its own previous work (its frames' scratch
    // buffers) and the upload, nothing else -- the next batch's pyramid of sub-batch 0 starts while this batch's
    // sub-batch 1 is still in its quadtree.  Consumers join through evPart (join_parts).
    // evFrames = "the frames are there", for the streams that did not carry them: the upload's own event (host path,
    // throughput mode: the frames arrive on a copy stream) or one recorded behind the upload on the stream that did
    // (latency mode).  A device-resident call has nothing to announce (the caller's frames are complete, every hazard
    // on the scratch buffers is ordered by the sub-batch chains below).  Never an event recorded on the host-facing
    // stream: it would sit behind the previous step's blur, and the next pyramid would wait for a kernel it does not
    // depend on.
    const bool devCall = !evUploaded && sIn == s0;
    hipEvent_t evFrames = nullptr;
    if (evUploaded) {
        // per sub-batch, below
    } else if (!h->serial && !devCall) {
        HIPCHK(hipEventRecord(h->evStart, sIn));
        evFrames = h->evStart;
    }
    // A frame's scratch (pyramid and blur levels, candidate segments, kept records) is ordered between two calls by
    // the stream of the sub-batch that owns the frame.  When the batch size -- and with it the frame -> sub-batch
    // map -- changes between two calls that the caller did not separate by a sync, a frame can change hands: its new
    // owner's pyramid would overwrite what the old owner's descriptor kernel may still be reading.  On such a call
    // (never in a steady stream) every stream first joins all sub-batches of the previous call.
    if (!h->serial && h->lastParts > 0 && (B != h->prevB || nsplit != h->prevSplit)) {
        for (int p = 0; p < h->lastParts; p++) {
            HIPCHK(hipStreamWaitEvent(s0, h->evPart[p], 0));
            for (int q = 0; q < nsplit; q++) HIPCHK(hipStreamWaitEvent(h->streamP[q], h->evPart[p], 0));
        }
    }
    h->prevB = B; h->prevSplit = nsplit;
    h->lastParts = 0;
    h->curSet ^= 1;
    const int set = h->curSet;
    const int cellsL0 = g.lv[0].nCells;
    if (lat) {
        hipStream_t sm = h->streamP[0], sa = s0;
        Launcher L{h, src, B};
        if (evFrames && sIn != sm) HIPCHK(hipStreamWaitEvent(sm, evFrames, 0));
        if (evFrames && sIn != sa) HIPCHK(hipStreamWaitEvent(sa, evFrames, 0));
        // aux: level 0 needs no pyramid.  (The previous call's quadtree and descriptors, which read what these two
        // overwrite, ran on `sm` in front of the upload / evStart that `sa` has just been made to follow.)
        if (sIn == sa && h->partEverRan[0]) HIPCHK(hipStreamWaitEvent(sa, h->evPart[0], 0));
        L.fast(sa, 0, cellsL0);
        L.dist(sa, 0, 1);
        if ((rc = L.pyramid(sm, h->evPyr[0]))) return rc;
        L.fast(sm, cellsL0, g.totalCells - cellsL0);
        L.dist(sm, 1, g.nlevels - 1);
        HIPCHK(hipStreamWaitEvent(sa, h->evPyr[0], 0));
        L.blur(sa);
        HIPCHK(hipEventRecord(h->evFast0[0], sa));  // the aux chain is through
        HIPCHK(hipStreamWaitEvent(sm, h->evFast0[0], 0));
        if ((rc = L.desc(sm, set, h->evPart[h->lastParts++]))) return rc;
        h->partEverRan[0] = true;
    } else {
        for (int part = 0; part < nsplit; part++) {
            const int f0 = (int)((int64_t)B * part / nsplit), f1 = (int)((int64_t)B * (part + 1) / nsplit);
            const int nb = f1 - f0;
            if (nb <= 0) continue;
            hipStream_t s = h->serial ? s0 : h->streamP[part];
            hipStream_t s2 = h->serial ? s : s0;  // blur: see orbx_create on the choice of streams
            if (evUploaded) {  // host path, throughput mode: this sub-batch's frames arrive on a copy stream
                HIPCHK(hipStreamWaitEvent(s, evUploaded[part], 0));
                if (s2 != s) HIPCHK(hipStreamWaitEvent(s2, evUploaded[part], 0));
            }
            if (evFrames && !h->serial) HIPCHK(hipStreamWaitEvent(s, evFrames, 0));
            if (evFrames && !h->serial && sIn != s0 && part == 0) HIPCHK(hipStreamWaitEvent(s0, evFrames, 0));
            src.f0 = f0;
            Launcher L{h, src, nb};
            // FAST of level 0 needs no pyramid: on the blur stream it runs beside the (latency-bound) pyramid kernel.
            // It overwrites this sub-batch's candidate segments, which the previous batch's quadtree read.
            const bool splitFast = !h->serial && g.nlevels > 1;
            if (splitFast) {
                if (h->partEverRan[part]) HIPCHK(hipStreamWaitEvent(s2, h->evPart[part], 0));
                L.fast(s2, 0, cellsL0);
                HIPCHK(hipEventRecord(h->evFast0[part], s2));
            } else {
                L.fast(s, 0, cellsL0);
            }
            if ((rc = L.pyramid(s))) return rc;
            // blur only needs the pyramid: run it on a second stream beside FAST + quadtree
            HIPCHK(hipEventRecord(h->evPyr[part], s));
            HIPCHK(hipStreamWaitEvent(s2, h->evPyr[part], 0));
            L.blur(s2);
            HIPCHK(hipEventRecord(h->evBlur[part], s2));
            // FAST of levels >= 1 behind the pyramid (level 0 went ahead, see above)
            L.fast(s, cellsL0, g.totalCells - cellsL0);
            if (splitFast) HIPCHK(hipStreamWaitEvent(s, h->evFast0[part], 0));
            L.dist(s, 0, g.nlevels);
            HIPCHK(hipStreamWaitEvent(s, h->evBlur[part], 0));
            if ((rc = L.desc(s, set))) return rc;
            HIPCHK(hipEventRecord(h->evPart[h->lastParts++], s));
            h->partEverRan[part] = true;
        }
    }
    HIPCHK(hipGetLastError());
    h->lastB = B;
    src.f0 = 0;
    h->lastSrc = src;
    return ORBX_OK;
}

extern "C" int orbx_extract_batch_device(orbx_t* h, const uint8_t* d_imgs, int B, int w, int hh, int stride, size_t pitch)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (!d_imgs || w < 1 || hh < 1) return fail(ORBX_E_INVALID, "empty device frame");
    return run_extract(h, d_imgs, B, w, hh, stride, pitch);
}

extern "C" int orbx_device_results(orbx_t* h, OrbxKeyPoint** d_kps, uint8_t** d_desc, int32_t** d_counts, int* cap)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (d_kps) *d_kps = (OrbxKeyPoint*)(r_kps(h, h->curSet) + h->maxKp);
    if (d_desc) *d_desc = r_desc(h, h->curSet) + (size_t)h->maxKp * 32;
    if (d_counts) *d_counts = r_count(h, h->curSet) + 1;
    if (cap) *cap = h->maxKp;
    return ORBX_OK;
}

extern "C" int orbx_sync(orbx_t* h)
{
    int rc = check_device(h);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
#ifdef ORBX_FAST_STATS
    {
        unsigned long long st[16];
        if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_fastStats), sizeof st) == hipSuccess && st[0])
            fprintf(stderr, "FAST stats: pass0 cells %llu visits %llu corners %llu | pass1 cells %llu visits %llu corners %llu | detection px %llu\n",
                    st[0], st[1], st[2], st[4], st[5], st[6], st[8]);
    }
#endif
    int32_t err = 0;
    HIPCHK(hipMemcpy(&err, h->d_err, sizeof err, hipMemcpyDeviceToHost));
    if (err) {
        (void)hipMemset(h->d_err, 0, sizeof err);
        return fail(ORBX_E_CAPACITY, "device scratch overflow (flags 0x%x)", err);
    }
    return ORBX_OK;
}

extern "C" int orbx_device_alloc(orbx_t* h, size_t bytes, void** d_ptr)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (!d_ptr || bytes == 0) return fail(ORBX_E_INVALID, "bad argument");
    HIPCHK(hipMalloc(d_ptr, bytes));
    return ORBX_OK;
}

extern "C" int orbx_device_free(orbx_t* h, void* d_ptr)
{
    int rc = check_device(h);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    if (d_ptr) HIPCHK(hipFree(d_ptr));
    return ORBX_OK;
}

extern "C" int orbx_upload(orbx_t* h, void* d_dst, const void* h_src, size_t bytes)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (!d_dst || !h_src) return fail(ORBX_E_INVALID, "null argument");
    HIPCHK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return ORBX_OK;
}

extern "C" int orbx_download(orbx_t* h, int frame, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out)
{
    int rc = orbx_sync(h);
    if (rc) return rc;
    if (frame < 0 || frame >= h->lastB) return fail(ORBX_E_INVALID, "frame %d not in the last batch", frame);
    int32_t n = 0;
    HIPCHK(hipMemcpy(&n, r_count(h, h->curSet) + 1 + frame, sizeof n, hipMemcpyDeviceToHost));
    if (n_out) *n_out = n;
    if (n > cap) return fail(ORBX_E_CAPACITY, "%d keypoints, caller capacity %d", n, cap);
    if (n > 0) {
        if (kps) HIPCHK(hipMemcpy(kps, r_kps(h, h->curSet) + (size_t)(frame + 1) * h->maxKp, (size_t)n * sizeof(OrbxKeyPoint), hipMemcpyDeviceToHost));
        if (desc) HIPCHK(hipMemcpy(desc, r_desc(h, h->curSet) + (size_t)(frame + 1) * h->maxKp * 32, (size_t)n * 32, hipMemcpyDeviceToHost));
    }
    return ORBX_OK;
}

// ------------------------------------------------------------------ host-buffer entries
// Frame::ExtractORB (src/Frame.cc:247-253) hands the extractor a host image and gets host vectors back.  Behind that
// boundary a batch goes through three stages that overlap across batches: upload on a copy stream | kernels | download
// on a second copy stream; a batch owns one HostSlot from orbx_submit_batch to orbx_release.
static int ensure_slots(orbx_handle* h)
{
    if (h->slotsReady) return ORBX_OK;
    const size_t B = (size_t)h->maxB;
    size_t o = 64;                                                    // [0]: the device error flag
    h->outOffN = o; o += align_up((int)(B * 4), 64);
    h->outOffNm = o; o += align_up((int)(B * 4), 64);
    h->outOffKp = o; o += B * h->maxKp * sizeof(OrbxKeyPointDev); o = (o + 63) & ~(size_t)63;
    h->outOffDesc = o; o += B * (size_t)h->maxKp * 32;
    h->outOffMatch = o; o += B * (size_t)h->maxKp * 4;
    h->outBytes = o;
    // Copy streams.  They are the process's fifth and later streams: the runtime deals its (four) hardware queues out
    // again and they share one with a compute stream.  For a batch submitted while others are in flight that is ruinous
    // -- the wait of the results for the matcher then sits in front of the next batch's first kernels in that queue and
    // the three tickets run one after the other (46 k frames/s where the frames alone allow 120 k).  A stream created
    // with a CU mask (here: all CUs) gets a hardware queue of its own: 46 k -> 69-74 k; but every hand-over through
    // such a queue costs ~0.1 ms, which a lone batch (nothing to overlap with) pays for nothing: B = 8 per call
    // 0.67 -> 0.97 ms.  Hence two pairs; orbx_submit_batch picks by whether another ticket is in flight.
    // (GPU_MAX_HW_QUEUES=6 set from outside gives the plain pair queues of their own as well.)
    {
        HIPCHK(hipStreamCreateWithFlags(&h->streamUp, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&h->streamDown, hipStreamNonBlocking));
        hipDeviceProp_t pr;
        HIPCHK(hipGetDeviceProperties(&pr, h->device));
        std::vector<uint32_t> mask((pr.multiProcessorCount + 31) / 32, 0xFFFFFFFFu);
        // (a runtime that refuses the mask leaves the pipelined batches on plain streams: slower, not wrong)
        if (hipExtStreamCreateWithCUMask(&h->streamUpQ, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(hipStreamCreateWithFlags(&h->streamUpQ, hipStreamNonBlocking));
        }
        // The results' stream gets 8 CUs: its one kernel (k_pack_host) is bound by the link, and its waves -- stalled on
        // writes over PCIe -- are better parked on a few CUs than spread over the wave slots of all of them
        // (pipelined, pinned frames: 77.3 k -> 81.5 k frames/s; 16 CUs 79.9 k, 4 CUs 81.4 k).
        for (size_t w = 0; w < mask.size(); w++) mask[w] = w == 0 ? 0xFFu : 0u;
        if (hipExtStreamCreateWithCUMask(&h->streamDownQ, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(hipStreamCreateWithFlags(&h->streamDownQ, hipStreamNonBlocking));
        }
        // (a queue is made at its stream's first use, tens of milliseconds: here, not in the first batch)
        for (hipStream_t st : {h->streamUp, h->streamDown, h->streamUpQ, h->streamDownQ}) { HIPCHK(hipMemsetAsync(h->d_err + 2, 0, 4, st)); HIPCHK(hipStreamSynchronize(st)); }
    }
    for (auto& sl : h->slot) {
        HIPCHK(hipHostMalloc(&sl.h_in, h->imgFrameBytes * B));
        // polled by the host while the kernel that fills it is still running: explicitly coherent (fine-grained), whatever HIP_HOST_COHERENT says
        HIPCHK(hipHostMalloc(&sl.h_out, h->outBytes, hipHostMallocCoherent));
        HIPCHK(hipMalloc(&sl.d_in, h->imgFrameBytes * B));
        for (hipEvent_t& e : sl.evUp) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sl.evOut, hipEventDisableTiming));
    }
    const unsigned hc = std::thread::hardware_concurrency();
    int nth = hc > 8 ? 6 : (hc > 2 ? (int)hc / 2 : 0);
    if (const char* e = getenv("ORBX_COPY_THREADS")) nth = atoi(e);
    if (nth > 0) h->pool.start(std::min(nth, 16), h->device);
    h->slotsReady = true;
    return ORBX_OK;
}

// memory the DMA engines can read in place: hipHostMalloc'ed or hipHostRegister'ed (orbx_host_alloc_frames / orbx_host_register)
static bool is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

extern "C" int orbx_host_alloc_frames(orbx_t* h, int B, int w, int hh, uint8_t** frames, int* stride, size_t* pitch)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (!frames || B < 1 || w < 1 || hh < 1) return fail(ORBX_E_INVALID, "bad argument");
    const int st = align_up(w, 64);
    void* p = nullptr;
    HIPCHK(hipHostMalloc(&p, (size_t)st * hh * B));
    *frames = (uint8_t*)p;
    if (stride) *stride = st;
    if (pitch) *pitch = (size_t)st * hh;
    return ORBX_OK;
}

extern "C" int orbx_host_alloc(orbx_t* h, size_t bytes, void** p)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (!p || bytes == 0) return fail(ORBX_E_INVALID, "bad argument");
    HIPCHK(hipHostMalloc(p, bytes, hipHostMallocCoherent));  // read by the host behind an event or a flag: coherent
    return ORBX_OK;
}

extern "C" int orbx_host_free(orbx_t* h, void* p)
{
    int rc = check_device(h);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    if (p) HIPCHK(hipHostFree(p));
    return ORBX_OK;
}

extern "C" int orbx_host_register(orbx_t* h, void* p, size_t bytes)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (!p || !bytes) return fail(ORBX_E_INVALID, "bad argument");
    HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return ORBX_OK;
}

extern "C" int orbx_host_unregister(orbx_t* h, void* p)
{
    int rc = check_device(h);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    if (p) HIPCHK(hipHostUnregister(p));
    return ORBX_OK;
}

static int slot_of(orbx_handle* h, int ticket, int want, HostSlot** out);

static int submit_core(orbx_t* h, const uint8_t* const* imgs, int B, int w, int hh, int stride,
                       const OrbxStreamOpts* opts, const OrbxBatchOut* into, int* ticket)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (into) {
        // results straight into the caller's arrays: the result kernel writes them over the link, so they must be
        // memory the device can address (orbx_host_alloc, or any buffer registered once with orbx_host_register)
        if (!into->kps || !into->desc || !into->n || into->cap < 1) return fail(ORBX_E_INVALID, "bad output arrays");
        if (opts && opts->match_prev && (!into->match || !into->nmatch)) return fail(ORBX_E_INVALID, "match tables requested without arrays for them");
        const void* ptrs[] = {into->kps, into->desc, into->n, into->match, into->nmatch};
        for (const void* q : ptrs) if (q && !is_pinned(q)) return fail(ORBX_E_INVALID, "output arrays must be pinned: allocate them with orbx_host_alloc or register them once with orbx_host_register");
    }
    if (!imgs || B < 1 || !ticket) return fail(ORBX_E_INVALID, "no frames");
    if (B > h->maxB) return fail(ORBX_E_INVALID, "batch %d exceeds %d", B, h->maxB);
    if (w < 1 || hh < 1) return fail(ORBX_E_INVALID, "empty frame");
    if (w > h->maxW || hh > h->maxH) return fail(ORBX_E_INVALID, "frame exceeds the handle's maximum");
    if (stride < w) return fail(ORBX_E_INVALID, "stride < width");
    for (int f = 0; f < B; f++) if (!imgs[f]) return fail(ORBX_E_INVALID, "null frame %d", f);
    if ((rc = ensure_slots(h))) return rc;
    HostSlot& sl = h->slot[h->nextTicket % orbx_handle::kSlots];
    if (sl.state != 0) return fail(ORBX_E_INVALID, "%d batches in flight: collect / release ticket %d first", orbx_handle::kSlots, sl.ticket);
    if ((rc = configure_shape(h, w, hh))) return rc;  // a new shape drains the streams before anything is overwritten

    const int dstride = align_up(w, 64);
    const size_t dpitch = (size_t)dstride * hh;
    // One or two frames per call is the latency mode: the chain upload -> kernels -> results IS the call, so the frame
    // goes up on the stream its first kernel runs on (no event hop) and the results come back through one kernel that
    // writes the pinned host buffer (k_pack_host) instead of six copies.  Larger batches are the throughput mode: copy
    // streams of their own, so that the DMA of neighbouring batches runs beside the kernels.
    const bool lat = B <= 2 && !h->serial;
    bool behind = false;  // another ticket is in flight: this batch's copies must not share a hardware queue with kernels
    for (const HostSlot& o : h->slot) behind |= o.state == 1;
    hipStream_t up = lat ? h->streamP[0] : (behind ? h->streamUpQ : h->streamUp);  // latency mode: the stream of the pyramid, the chain's first kernel
    const bool pinned = is_pinned(imgs[0]);
    bool contiguous = true;  // frames back to back at a constant pitch of whole rows
    for (int f = 1; f < B && contiguous; f++) contiguous = imgs[f] == imgs[0] + (size_t)f * stride * hh;
    for (int f = 1; f < B; f++)
        if (is_pinned(imgs[f]) != pinned) return fail(ORBX_E_INVALID, "frame %d is %s, frame 0 %s: one kind per batch", f, pinned ? "pageable" : "pinned", pinned ? "pinned" : "pageable");
    h->prof.begin(P_H2D, up);
    // Throughput mode uploads the batch in the parts run_extract cuts it into (same frame ranges), an event behind each:
    // sub-batch 0's kernels start when its half is there, and for pageable frames the staging of part p + 1 (host
    // threads) runs beside the DMA of part p.  Latency mode: one part, no event (the frames ride on the kernels' stream).
    const int nparts = lat || h->serial ? 1 : std::min(h->nsplit, B);
    for (int part = 0; part < nparts; part++) {
        const int F0 = (int)((int64_t)B * part / nparts), F1 = (int)((int64_t)B * (part + 1) / nparts), nb = F1 - F0;
        if (nb <= 0) continue;
        uint8_t* const dst = sl.d_in + (size_t)F0 * dpitch;
        if (pinned && contiguous && stride == dstride) {
            // frames from orbx_host_alloc_frames: already in the device layout, the DMA reads the caller's memory
            HIPCHK(hipMemcpyAsync(dst, imgs[F0], dpitch * nb, hipMemcpyHostToDevice, up));
        } else if (pinned && contiguous) {
            HIPCHK(hipMemcpy2DAsync(dst, dstride, imgs[F0], stride, w, (size_t)hh * nb, hipMemcpyHostToDevice, up));
        } else if (pinned) {
            for (int f = F0; f < F1; f++)
                HIPCHK(hipMemcpy2DAsync(sl.d_in + f * dpitch, dstride, imgs[f], stride, w, hh, hipMemcpyHostToDevice, up));
        } else {
            // pageable frames: row-band jobs over the copy threads into the pinned staging, one DMA for the part
            // (one 1241x376 frame: 34 us on one core -- 376 row copies -- against ~15 us over four)
            const int bands = 4, jobs = nb * bands;
            uint8_t* const hin = sl.h_in;
            h->pool.run(jobs, [=](int j) {
                const int f = F0 + j / bands, c = j % bands;
                const int y0 = (int)((int64_t)hh * c / bands), y1 = (int)((int64_t)hh * (c + 1) / bands);
                for (int y = y0; y < y1; y++) memcpy(hin + f * dpitch + (size_t)y * dstride, imgs[f] + (size_t)y * stride, (size_t)w);
            });
            // (one DMA also in the latency mode: four band DMAs sent off by the workers as they finish were slower --
            // a copy of this size is mostly its fixed cost, ~7 of 16 us)
            HIPCHK(hipMemcpyAsync(dst, sl.h_in + (size_t)F0 * dpitch, dpitch * nb, hipMemcpyHostToDevice, up));
        }
        if (!lat) HIPCHK(hipEventRecord(sl.evUp[part], up));
    }
    h->prof.end(up);

    // this batch's kernels report scratch overflows in the slot's own word; its k_pack_host hands the word to the host
    // and clears it (one sticky word for all batches would report an overflow of ticket n for n+1 and n+2 as well)
    int32_t* const errWord = h->d_err + 4 + h->nextTicket % orbx_handle::kSlots;
    h->d_errCur = errWord;
    rc = run_extract(h, sl.d_in, B, w, hh, dstride, dpitch, lat ? nullptr : sl.evUp, lat ? up : nullptr);
    h->d_errCur = h->d_err;
    if (rc) return rc;
    const bool match = opts && opts->match_prev;
    const int set = h->curSet;
    int32_t* const dm = h->d_match + (size_t)set * h->maxB * h->maxKp;
    int32_t* const dnm = h->d_nmatch + (size_t)set * h->maxB;
    hipStream_t outS = nullptr;
    if (lat) {
        // One chain on one stream: descriptors -> matching -> the result kernel, which writes the pinned buffer directly
        // and raises the flag the caller polls -> only then the roll of the previous-frame slot.
        hipStream_t ps = h->streamP[0];
        if (match && (rc = match_prev_on(h, ps, opts->nnratio, opts->th_low, opts->check_ori, false))) return rc;
        PackArgs pa;
        pa.kps = (const uint32_t*)(r_kps(h, set) + h->maxKp); pa.desc = (const uint32_t*)(r_desc(h, set) + (size_t)h->maxKp * 32);
        pa.count = r_count(h, set) + 1; pa.match = match ? dm : nullptr; pa.nmatch = match ? dnm : nullptr; pa.err = errWord;
        pa.hKps = (uint32_t*)(sl.h_out + h->outOffKp); pa.hDesc = (uint32_t*)(sl.h_out + h->outOffDesc);
        pa.hN = (int32_t*)(sl.h_out + h->outOffN); pa.hMatch = (int32_t*)(sl.h_out + h->outOffMatch);
        pa.hNmatch = (int32_t*)(sl.h_out + h->outOffNm); pa.hErr = (int32_t*)sl.h_out; pa.maxKp = h->maxKp; pa.hostPitch = h->maxKp;
        if (into) { pa.hKps = (uint32_t*)into->kps; pa.hDesc = (uint32_t*)into->desc; pa.hN = into->n; pa.hMatch = into->match; pa.hNmatch = into->nmatch; pa.hostPitch = into->cap; }
        pa.hFlag = (int32_t*)sl.h_out + 1; pa.flagValue = h->nextTicket + 1; pa.blocksDone = h->d_err + 1;
        ((volatile int32_t*)sl.h_out)[1] = 0;
        h->prof.begin(P_D2H, ps);
        hipLaunchKernelGGL(k_pack_host, dim3(16, B), dim3(256), 0, ps, pa);
        h->prof.end(ps);
        HIPCHK(hipEventRecord(sl.evOut, ps));
        outS = ps;
        if (match && (rc = roll_prev_on(h, ps, set))) return rc;
    } else {
        // download behind the batch's kernels on the second copy stream: full-capacity slots, one pass, no host sync
        hipStream_t dn = behind ? h->streamDownQ : h->streamDown;
        if ((rc = join_parts(h, dn))) return rc;
        if (match && (rc = orbx_match_prev_batch_device(h, opts->nnratio, opts->th_low, opts->check_ori))) return rc;
        if (match) HIPCHK(hipStreamWaitEvent(dn, h->evMatched[set], 0));
        // One kernel writes the exact n keypoints / descriptors / match entries of every frame into the pinned buffer.
        // (Six device-to-host copies of full-capacity slots went through the runtime's blit path with ~110 us between
        // consecutive copies: 0.55 ms per 64-frame batch, as long as the upload.  No flag here: the consumer waits for evOut.)
        PackArgs pa;
        pa.kps = (const uint32_t*)(r_kps(h, set) + h->maxKp); pa.desc = (const uint32_t*)(r_desc(h, set) + (size_t)h->maxKp * 32);
        pa.count = r_count(h, set) + 1; pa.match = match ? dm : nullptr; pa.nmatch = match ? dnm : nullptr; pa.err = errWord;
        pa.hKps = (uint32_t*)(sl.h_out + h->outOffKp); pa.hDesc = (uint32_t*)(sl.h_out + h->outOffDesc);
        pa.hN = (int32_t*)(sl.h_out + h->outOffN); pa.hMatch = (int32_t*)(sl.h_out + h->outOffMatch);
        pa.hNmatch = (int32_t*)(sl.h_out + h->outOffNm); pa.hErr = (int32_t*)sl.h_out; pa.maxKp = h->maxKp; pa.hostPitch = h->maxKp;
        if (into) { pa.hKps = (uint32_t*)into->kps; pa.hDesc = (uint32_t*)into->desc; pa.hN = into->n; pa.hMatch = into->match; pa.hNmatch = into->nmatch; pa.hostPitch = into->cap; }
        pa.hFlag = nullptr; pa.flagValue = 0; pa.blocksDone = nullptr;  // the consumer waits for evOut: the kernel's end publishes
        h->prof.begin(P_D2H, dn);
        // two workgroups per frame: the kernel is bound by the link (8 MB at ~45 GB/s), more waves only sit on the CUs
        hipLaunchKernelGGL(k_pack_host, dim3(2, B), dim3(256), 0, dn, pa);
        h->prof.end(dn);
        HIPCHK(hipEventRecord(sl.evOut, dn));
        outS = dn;
    }
    HIPCHK(hipGetLastError());
    h->evOutOfSet[set] = sl.evOut;
    h->outStream[set] = outS;
    sl.state = 1; sl.B = B; sl.matched = match; sl.ticket = h->nextTicket; sl.lat = lat;
    sl.into = into != nullptr; sl.intoN = into ? into->n : nullptr; sl.intoCap = into ? into->cap : 0;
    *ticket = h->nextTicket++;
    return ORBX_OK;
}

extern "C" int orbx_submit_batch(orbx_t* h, const uint8_t* const* imgs, int B, int w, int hh, int stride,
                                 const OrbxStreamOpts* opts, int* ticket)
{
    return submit_core(h, imgs, B, w, hh, stride, opts, nullptr, ticket);
}

extern "C" int orbx_submit_batch_into(orbx_t* h, const uint8_t* const* imgs, int B, int w, int hh, int stride,
                                      const OrbxStreamOpts* opts, const OrbxBatchOut* out, int* ticket)
{
    if (!out) return fail(ORBX_E_INVALID, "null output");
    return submit_core(h, imgs, B, w, hh, stride, opts, out, ticket);
}

// waits for a ticket of orbx_submit_batch_into: its results are in the caller's arrays; releases the ticket
extern "C" int orbx_collect(orbx_t* h, int ticket)
{
    int rc = check_device(h);
    if (rc) return rc;
    HostSlot* sl;
    if ((rc = slot_of(h, ticket, 1, &sl))) return rc;
    if (!sl->into) return fail(ORBX_E_INVALID, "ticket %d was not submitted with output arrays: use orbx_collect_view / orbx_collect_batch", ticket);
    bool landed = false;
    if (sl->lat) {
        volatile int32_t* flag = (volatile int32_t*)sl->h_out + 1;
        for (int spin = 0; spin < 400000 && !landed; spin++) { landed = *flag == ticket + 1; if (!landed) cpu_relax(); }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    if (!landed) HIPCHK(hipEventSynchronize(sl->evOut));
    const int32_t err = *(const int32_t*)sl->h_out;
    int over = 0;
    for (int f = 0; f < sl->B; f++) over |= sl->intoN[f] > sl->intoCap;
    sl->state = 0;
    if (err) return fail(ORBX_E_CAPACITY, "device scratch overflow (flags 0x%x)", err);
    if (over) return fail(ORBX_E_CAPACITY, "a frame produced more keypoints than cap=%d (n[] holds the true counts, the arrays the first cap)", sl->intoCap);
    return ORBX_OK;
}

static int slot_of(orbx_handle* h, int ticket, int want, HostSlot** out)
{
    if (!h->slotsReady || ticket < 0) return fail(ORBX_E_INVALID, "unknown ticket %d", ticket);
    HostSlot& sl = h->slot[ticket % orbx_handle::kSlots];
    if (sl.ticket != ticket || sl.state == 0 || (want && sl.state != want)) return fail(ORBX_E_INVALID, "ticket %d is not %s", ticket, want == 1 ? "in flight" : "outstanding");
    *out = &sl;
    return ORBX_OK;
}

extern "C" int orbx_collect_view(orbx_t* h, int ticket, OrbxBatchView* view)
{
    int rc = check_device(h);
    if (rc) return rc;
    HostSlot* sl;
    if ((rc = slot_of(h, ticket, 1, &sl))) return rc;
    if (!view) return fail(ORBX_E_INVALID, "null view");
    if (sl->into) return fail(ORBX_E_INVALID, "ticket %d was submitted with output arrays: use orbx_collect", ticket);
    bool landed = false;
    if (sl->lat) {
        // the pack kernel's last block writes ticket + 1 behind the results (system-scope release): polling the pinned
        // word saves the wake-up of an event wait on the one-frame-per-call path
        volatile int32_t* flag = (volatile int32_t*)sl->h_out + 1;
        for (int spin = 0; spin < 400000 && !landed; spin++) { landed = *flag == ticket + 1; if (!landed) cpu_relax(); }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    if (!landed) HIPCHK(hipEventSynchronize(sl->evOut));
    sl->state = 2;
    const int32_t err = *(const int32_t*)sl->h_out;
    view->B = sl->B; view->cap = h->maxKp;
    view->n = (const int32_t*)(sl->h_out + h->outOffN);
    view->kps = (const OrbxKeyPoint*)(sl->h_out + h->outOffKp);
    view->desc = sl->h_out + h->outOffDesc;
    view->match = sl->matched ? (const int32_t*)(sl->h_out + h->outOffMatch) : nullptr;
    view->nmatch = sl->matched ? (const int32_t*)(sl->h_out + h->outOffNm) : nullptr;
    if (err) return fail(ORBX_E_CAPACITY, "device scratch overflow (flags 0x%x)", err);  // (the pack kernel cleared the slot's word)
    return ORBX_OK;
}

extern "C" int orbx_release(orbx_t* h, int ticket)
{
    int rc = check_device(h);
    if (rc) return rc;
    HostSlot* sl;
    if ((rc = slot_of(h, ticket, 0, &sl))) return rc;
    if (sl->state == 1) HIPCHK(hipEventSynchronize(sl->evOut));  // abandoned in flight: its buffers are free once it has drained
    sl->state = 0;
    return ORBX_OK;
}

extern "C" int orbx_collect_batch(orbx_t* h, int ticket, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out,
                                  int32_t* match, int* nmatch)
{
    OrbxBatchView v;
    int rc = orbx_collect_view(h, ticket, &v);
    if (rc) { if (rc != ORBX_E_INVALID) { const std::string keep = g_err; (void)orbx_release(h, ticket); g_err = keep; } return rc; }
    int over = 0;
    for (int f = 0; f < v.B; f++) {
        if (n_out) n_out[f] = v.n[f];
        if (nmatch) nmatch[f] = v.nmatch ? v.nmatch[f] : 0;
        if (v.n[f] > cap) over = 1;
    }
    {   // frames that fit are copied also when another one does not (E_CAPACITY below names the call, not every frame)
        const int mk = h->maxKp;
        h->pool.run(v.B, [=](int f) {
            const size_t n = (size_t)v.n[f];
            if (!n || n > (size_t)cap) return;
            if (kps) memcpy(kps + (size_t)f * cap, v.kps + (size_t)f * mk, n * sizeof(OrbxKeyPoint));
            if (desc) memcpy(desc + (size_t)f * cap * 32, v.desc + (size_t)f * mk * 32, n * 32);
            if (match && v.match) memcpy(match + (size_t)f * cap, v.match + (size_t)f * mk, n * 4);
        });
    }
    if ((rc = orbx_release(h, ticket))) return rc;
    if (over) return fail(ORBX_E_CAPACITY, "a frame produced more keypoints than cap=%d", cap);
    return ORBX_OK;
}

extern "C" int orbx_extract_match_batch(orbx_t* h, const uint8_t* const* imgs, int B, int w, int hh, int stride,
                                        const OrbxStreamOpts* opts, OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out,
                                        int32_t* match, int* nmatch)
{
    int t = -1;
    int rc = orbx_submit_batch(h, imgs, B, w, hh, stride, opts, &t);
    if (rc) return rc;
    return orbx_collect_batch(h, t, kps, desc, cap, n_out, match, nmatch);
}

extern "C" int orbx_extract_batch(orbx_t* h, const uint8_t* const* imgs, int B, int w, int hh, int stride,
                                  OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (!imgs || B < 1) return fail(ORBX_E_INVALID, "no frames");
    if (B > h->maxB) return fail(ORBX_E_INVALID, "batch %d exceeds %d", B, h->maxB);
    if (w < 1 || hh < 1) { for (int f = 0; f < B; f++) if (n_out) n_out[f] = 0; return ORBX_OK; }  // :1046-1047
    return orbx_extract_match_batch(h, imgs, B, w, hh, stride, nullptr, kps, desc, cap, n_out, nullptr, nullptr);
}

extern "C" int orbx_extract(orbx_t* h, const uint8_t* img, int w, int hh, int stride,
                            OrbxKeyPoint* kps, uint8_t* desc, int cap, int* n_out)
{
    if (!img || w < 1 || hh < 1) { if (n_out) *n_out = 0; return ORBX_OK; }  // empty image: silent return (:1046-1047)
    const uint8_t* one[1] = {img};
    return orbx_extract_batch(h, one, 1, w, hh, stride, kps, desc, cap, n_out);
}

// void Frame::ComputeStereoMatches()   src/Frame.cc:466-638
extern "C" int orbx_compute_stereo_matches(orbx_t* left, orbx_t* right, int frame, float mb, float mbf,
                                           float* u_right, float* depth, int cap, int* n_left)
{
    int rc = orbx_sync(right);
    if (rc) return rc;
    if ((rc = orbx_sync(left))) return rc;  // also leaves the device of `left` current
    if (left->device != right->device) return fail(ORBX_E_INVALID, "the two extractors live on different devices");
    if (left->curW == 0 || left->curW != right->curW || left->curH != right->curH || left->nlevels != right->nlevels)
        return fail(ORBX_E_INVALID, "left and right frames differ in shape or pyramid");
    if (frame < 0 || frame >= left->lastB || frame >= right->lastB) return fail(ORBX_E_INVALID, "frame %d not in the last batches", frame);
    if (!(mb > 0.f) || !(mbf > 0.f)) return fail(ORBX_E_INVALID, "stereo baseline (mb, mbf) must be positive");
    if (left->maxKp >= 65536 || right->maxKp >= 65536) return fail(ORBX_E_UNSUPPORTED, "more than 65535 keypoints per frame");
    int32_t n = 0;
    HIPCHK(hipMemcpy(&n, r_count(left, left->curSet) + 1 + frame, sizeof n, hipMemcpyDeviceToHost));
    if (n_left) *n_left = n;
    if (n > cap) return fail(ORBX_E_CAPACITY, "%d keypoints, caller capacity %d", n, cap);
    if (n == 0) return ORBX_OK;
    // scratch: three arrays of maxKp entries, grown on first use
    const size_t need = (size_t)left->maxKp * 12 + 64;
    if (need > left->stereoBytes) {
        if (left->d_stereo) HIPCHK(hipFree(left->d_stereo));
        left->d_stereo = nullptr; left->stereoBytes = 0;
        HIPCHK(hipMalloc(&left->d_stereo, need));
        left->stereoBytes = need;
    }
    StereoArgs a;
    a.kL = r_kps(left, left->curSet) + (size_t)(frame + 1) * left->maxKp; a.dL = r_desc(left, left->curSet) + (size_t)(frame + 1) * left->maxKp * 32;
    a.nL = r_count(left, left->curSet) + 1 + frame;
    a.kR = r_kps(right, right->curSet) + (size_t)(frame + 1) * right->maxKp; a.dR = r_desc(right, right->curSet) + (size_t)(frame + 1) * right->maxKp * 32;
    a.nR = r_count(right, right->curSet) + 1 + frame;
    a.gL = left->d_geom; a.gR = right->d_geom;
    a.srcL = left->lastSrc; a.srcR = right->lastSrc; a.srcL.f0 = a.srcR.f0 = 0;
    a.fL = a.fR = frame;
    for (int l = 0; l < ORBX_MAXL; l++) { a.sf[l] = l < left->nlevels ? left->mvScaleFactor[l] : 1.f; a.isf[l] = l < left->nlevels ? left->mvInvScaleFactor[l] : 1.f; }
    a.mb = mb; a.mbf = mbf;
    a.uRight = (float*)left->d_stereo; a.depth = a.uRight + left->maxKp; a.sad = (int32_t*)(a.depth + left->maxKp);
    int32_t* d_nAcc = a.sad + left->maxKp;
    hipStream_t s = left->stream;
    hipLaunchKernelGGL(k_stereo_match, dim3((n + 3) / 4), dim3(256), 0, s, a);
    const size_t lds = (size_t)left->maxKp * 4;
    if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_stereo_median, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_stereo_median, dim3(1), dim3(1024), lds, s, a.nL, (const int32_t*)a.sad, a.uRight, a.depth, d_nAcc);
    HIPCHK(hipGetLastError());
    if (u_right) HIPCHK(hipMemcpyAsync(u_right, a.uRight, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    if (depth) HIPCHK(hipMemcpyAsync(depth, a.depth, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ORBX_OK;
}

extern "C" int orbx_pyramid_level(orbx_t* h, int frame, int level, int blurred, uint8_t* dst, int* w, int* hh)
{
    int rc = orbx_sync(h);
    if (rc) return rc;
    if (h->curW == 0) return fail(ORBX_E_INVALID, "no frame extracted yet");
    if (level < 0 || level >= h->geom.nlevels || frame < 0 || frame >= h->lastB) return fail(ORBX_E_INVALID, "bad frame/level");
    const LevelGeom& L = h->geom.lv[level];
    if (w) *w = L.w;
    if (hh) *hh = L.h;
    if (!dst) return ORBX_OK;
    const uint8_t* srcp; size_t sp;
    if (blurred) { srcp = h->d_blur + (size_t)frame * h->geom.blurFrameBytes + L.blurOff; sp = L.blurStride; }
    else if (level == 0) { srcp = h->lastSrc.img0 + (size_t)frame * h->lastSrc.pitch0; sp = h->lastSrc.stride0; }
    else { srcp = h->d_pyr + (size_t)frame * h->geom.pyrFrameBytes + L.pyrOff; sp = L.stride; }
    HIPCHK(hipMemcpy2D(dst, L.w, srcp, sp, L.w, L.h, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbx_level_candidates(orbx_t* h, int frame, int level, uint64_t* dst, int cap, int* n)
{
    int rc = orbx_sync(h);
    if (rc) return rc;
    if (h->curW == 0 || level < 0 || level >= h->geom.nlevels || frame < 0 || frame >= h->lastB) return fail(ORBX_E_INVALID, "bad frame/level");
    // FAST leaves every cell's survivors in the cell's own segment: gather them
    const LevelGeom& L = h->geom.lv[level];
    std::vector<int32_t> counts(std::max(L.nCells, 1));
    if (L.nCells) HIPCHK(hipMemcpy(counts.data(), h->d_cellCount + (size_t)frame * h->geom.totalCells + L.cellBase, (size_t)L.nCells * 4, hipMemcpyDeviceToHost));
    int total = 0;
    for (int c = 0; c < L.nCells; c++) total += counts[c];
    if (n) *n = total;
    if (!dst) return ORBX_OK;
    if (total > cap) return fail(ORBX_E_CAPACITY, "%d candidates, capacity %d", total, cap);
    std::vector<uint64_t> seg(L.candCap);
    HIPCHK(hipMemcpy(seg.data(), h->d_candRaw + (size_t)frame * h->geom.candFrameRecs + L.candOff, (size_t)L.candCap * 8, hipMemcpyDeviceToHost));
    int o = 0;
    for (int c = 0; c < L.nCells; c++)
        for (int i = 0; i < counts[c]; i++) dst[o++] = seg[h->cells[L.cellBase + c].candOff + i];
    return ORBX_OK;
}

// ------------------------------------------------------------------ stream matching
static orbm::MatchIO slots_io(orbx_handle* h, int set)
{
    orbm::MatchIO io;
    io.desc = r_desc(h, set); io.descPitch = (int64_t)h->maxKp * 32;
    io.ang = &((const float*)r_kps(h, set))[3]; io.angStride = 7; io.angPitch = (int64_t)h->maxKp * 7;
    io.count = r_count(h, set);
    return io;
}

// last frame of the batch in `set` becomes the stream's previous frame: slot 0 of the set the next extraction fills
static int roll_prev_on(orbx_handle* h, hipStream_t s, int set)
{
    const int B = h->lastB;
    hipLaunchKernelGGL(k_roll_prev, dim3(16), dim3(256), 0, s, (const uint32_t*)(r_kps(h, set) + (size_t)B * h->maxKp),
                       (const uint32_t*)(r_desc(h, set) + (size_t)B * h->maxKp * 32), (const int32_t*)(r_count(h, set) + B),
                       (uint32_t*)r_kps(h, set ^ 1), (uint32_t*)r_desc(h, set ^ 1), r_count(h, set ^ 1));
    HIPCHK(hipEventRecord(h->evMatch[set], s));
    HIPCHK(hipGetLastError());
    h->matchPending[set] = true;
    h->matchStream[set] = s;
    h->havePrev = true;
    return ORBX_OK;
}

// the kernels of the stream matcher for the B frames of result set `set` (no waits, no roll)
static void match_kernels(orbx_handle* h, hipStream_t s, int set, int B, float nnratio, int th_low, int check_ori)
{
    orbm::MatchIO io = slots_io(h, set);
    int32_t* const d_match = h->d_match + (size_t)set * h->maxB * h->maxKp;  // one table per result set: the host path's
    int32_t* const d_nmatch = h->d_nmatch + (size_t)set * h->maxB;           // download of batch n-1 runs beside batch n
    h->matchSet = set;
    h->prof.begin(P_MATCH_BEST2, s);
    bool fused = false;
    // slots 0..B expanded to +-1 bytes, then the Hamming scan as an int8 MFMA product (train slot f, query slot f+1)
    // with the acceptance rule in its epilogue
    const orbm::AcceptArgs aa = {io, io, 1, 0, nnratio, th_low, check_ori, d_match, (int64_t)h->maxKp, h->d_binOf, h->d_hist};
    // the MFMA scan packs the train index into 16 bits of its key: larger frames take the popcount scan (20-bit index)
    if (h->matchPopcount || h->maxKp >= 65536) {  // the literal xor + popcount scan (lane = query, train descriptor wave-uniform), kept for A/B runs
        hipLaunchKernelGGL(orbm::k_match_best2, dim3((h->maxKp + 255) / 256, B, kMatchChunks), dim3(256), 0, s, io, io, 1, 0,
                           kMatchChunks, h->d_partial, (int64_t)h->maxKp);
        hipLaunchKernelGGL(orbm::k_match_accept, dim3((h->maxKp + 255) / 256, B), dim3(256), 0, s, aa, kMatchChunks,
                           (const uint2*)h->d_partial, (int64_t)h->maxKp);
    } else {
        hipLaunchKernelGGL(orbm::k_expand_desc, dim3((unsigned)(h->xPitch / 4096), B + 1), dim3(256), 0, s, io, 0, 0, r_xdesc(h, set), h->xPitch);
        const int nqb = (h->maxKp + orbm::kMfmaRowsPerBlock - 1) / orbm::kMfmaRowsPerBlock;
        // few frames (the one-frame-per-call entry): cut the train side into chunks so that the scan fills more than B * nqb CUs
        int chunks = 1;
        while (chunks < 8 && (size_t)B * nqb * chunks < 64 && (size_t)B * chunks * 2 <= h->partialSlots) chunks *= 2;
        hipLaunchKernelGGL(orbm::k_match_mfma, dim3(8 * ((B + 7) / 8) * nqb, chunks), dim3(256), 0, s, (const uint8_t*)r_xdesc(h, set), h->xPitch, aa, nqb, B,
                           h->d_partial, (int64_t)h->maxKp);
        fused = chunks > 1;
        if (fused)  // merge + acceptance + histogram + pruning of a frame in one workgroup
            hipLaunchKernelGGL(orbm::k_match_accept_prune, dim3(B), dim3(1024), 0, s, aa, chunks, (const uint2*)h->d_partial,
                               (int64_t)h->maxKp, d_nmatch);
    }
    h->prof.end(s);
    if (!fused) {
        h->prof.begin(P_MATCH_PRUNE, s);
        hipLaunchKernelGGL(orbm::k_match_prune, dim3(B), dim3(256), 0, s, io, 1, check_ori, d_match, (int64_t)h->maxKp,
                           h->d_binOf, h->d_hist, d_nmatch);
        h->prof.end(s);
    }
}

// the matching of the last extracted batch on stream s; roll = false leaves the roll of the previous-frame slot to the
// caller (the latency path puts the result kernel in front of it)
static int match_prev_on(orbx_handle* h, hipStream_t s, float nnratio, int th_low, int check_ori, bool roll)
{
    int rc;
    const int B = h->lastB;
    if (B < 1) return fail(ORBX_E_INVALID, "no extracted batch to match");
    const int set = h->curSet;
    // what this stream does not already follow: the batch's descriptors, the previous batch's roll into slot 0 of this
    // set, the download that last read this set's tables
    for (int p = 0; p < h->lastParts; p++)
        if (!(h->lastParts == 1 && h->prevSplit == 1 && !h->serial && s == h->streamP[0])) HIPCHK(hipStreamWaitEvent(s, h->evPart[p], 0));
    if (h->matchPending[set ^ 1] && h->matchStream[set ^ 1] != s) HIPCHK(hipStreamWaitEvent(s, h->evMatch[set ^ 1], 0));
    if (h->evOutOfSet[set] && h->outStream[set] != s) HIPCHK(hipStreamWaitEvent(s, h->evOutOfSet[set], 0));
    match_kernels(h, s, set, B, nnratio, th_low, check_ori);
    HIPCHK(hipGetLastError());
    if (!roll) return ORBX_OK;
    HIPCHK(hipEventRecord(h->evMatched[set], s));  // the tables are final: the host path's download need not wait for the roll
    if ((rc = roll_prev_on(h, s, set))) return rc;
    return ORBX_OK;
}

extern "C" int orbx_match_prev_batch_device(orbx_t* h, float nnratio, int th_low, int check_ori)
{
    int rc = check_device(h);
    if (rc) return rc;
    // matching runs on its own stream so that the next batch's pyramid/FAST can start beside it
    return match_prev_on(h, h->serial ? h->stream : h->stream3, nnratio, th_low, check_ori, true);
}

extern "C" int orbx_device_matches(orbx_t* h, int32_t** d_match, int32_t** d_nmatch)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (d_match) *d_match = h->d_match + (size_t)h->matchSet * h->maxB * h->maxKp;
    if (d_nmatch) *d_nmatch = h->d_nmatch + (size_t)h->matchSet * h->maxB;
    return ORBX_OK;
}

extern "C" int orbx_download_matches(orbx_t* h, int frame, int32_t* match, int cap, int* nmatch)
{
    int rc = orbx_sync(h);
    if (rc) return rc;
    if (frame < 0 || frame >= h->lastB) return fail(ORBX_E_INVALID, "frame %d not in the last batch", frame);
    int32_t n = 0, nm = 0;
    HIPCHK(hipMemcpy(&n, r_count(h, h->curSet) + 1 + frame, sizeof n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&nm, h->d_nmatch + (size_t)h->matchSet * h->maxB + frame, sizeof nm, hipMemcpyDeviceToHost));
    if (nmatch) *nmatch = nm;
    if (n > cap) return fail(ORBX_E_CAPACITY, "%d queries, caller capacity %d", n, cap);
    if (match && n > 0) HIPCHK(hipMemcpy(match, h->d_match + ((size_t)h->matchSet * h->maxB + frame) * h->maxKp, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbx_reset_stream(orbx_t* h)
{
    int rc = check_device(h);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    for (int set = 0; set < 2; set++) HIPCHK(hipMemsetAsync(r_count(h, set), 0, sizeof(int32_t), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->havePrev = false;
    return ORBX_OK;
}

extern "C" int orbx_set_serial(orbx_t* h, int serial)
{
    int rc = check_device(h);
    if (rc) return rc;
    if ((rc = sync_all(h))) return rc;
    h->serial = serial != 0;
    return ORBX_OK;
}

extern "C" int orbx_profile_enable(orbx_t* h, int enable)
{
    int rc = check_device(h);
    if (rc) return rc;
    h->prof.on = enable != 0;
    // events for ~250 steps up front, so that the timed region creates none
    if (enable) while (h->prof.pool.size() < 8192) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; h->prof.pool.push_back(e); }
    return ORBX_OK;
}

extern "C" int orbx_profile_select(orbx_t* h, const char* kernel)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (!kernel) { h->prof.only = -1; return ORBX_OK; }
    for (int i = 0; i < P_COUNT; i++)
        if (!strcmp(kernel, kProfNames[i])) { h->prof.only = i; return ORBX_OK; }
    return fail(ORBX_E_INVALID, "no kernel named %s", kernel);
}

extern "C" int orbx_profile_read(orbx_t* h, OrbxProfile* out, int reset)
{
    int rc = check_device(h);
    if (rc) return rc;
    if (!out) return fail(ORBX_E_INVALID, "null argument");
    if ((rc = sync_all(h))) return rc;
    h->prof.collect();
    out->n = P_COUNT;
    for (int i = 0; i < P_COUNT; i++) {
        out->name[i] = kProfNames[i];
        out->ms[i] = h->prof.ms[i];
        out->launches[i] = h->prof.launches[i];
        if (reset) { h->prof.ms[i] = 0; h->prof.launches[i] = 0; }
    }
    return ORBX_OK;
}

// ------------------------------------------------------------------ co-run experiment (tools/pair_overlap.py)
// Kernel i repeated on one stream while kernel j runs on another for at least three times as long: the per-launch time of
// i beside j against i alone.  Works on the buffers of the last extracted + matched batch (every stage is idempotent on
// its inputs).  PMC counters cannot do this: rocprofv3 serialises dispatches while it collects them.
extern "C" int orbx_debug_pair_overlap(orbx_t* h, int nb, float target_ms, int* n_kernels, const char** names,
                                       float* alone_ms, float* co_ms, int32_t* lds_bytes, int32_t* wg_threads, int32_t* wgs)
{
    int rc = orbx_sync(h);
    if (rc) return rc;
    if (h->lastB < 1 || nb < 1 || nb > h->lastB) return fail(ORBX_E_INVALID, "run a batch of at least %d frames first", nb);
    constexpr int K = 6;
    static const char* kNames[K] = {"k_pyramid", "k_fast", "k_distribute", "k_blur", "k_orient_desc", "k_match_mfma"};
    if (n_kernels) *n_kernels = K;
    const Geom& g = h->geom;
    FrameSrc src = h->lastSrc; src.f0 = 0;
    Launcher L{h, src, nb};
    const int set = h->curSet, cellsL0 = g.lv[0].nCells;
    auto launch = [&](int k, hipStream_t s) {
        switch (k) {
        case 0: (void)L.pyramid(s); break;
        case 1: L.fast(s, 0, cellsL0); L.fast(s, cellsL0, g.totalCells - cellsL0); break;
        case 2: L.dist(s, 0, g.nlevels); break;
        case 3: L.blur(s); break;
        case 4: (void)L.desc(s, set); break;
        default: match_kernels(h, s, set, nb, 0.7f, 50, 1); break;
        }
    };
    if (lds_bytes && wg_threads && wgs) {
        const size_t pl = ((size_t)h->pyrBufA + h->pyrBufB) * 4 + (size_t)h->pyrTabCap * 16;
        const size_t fl = (size_t)2 * (h->tileRows * h->tileStrideDw + 4) * 4 + (size_t)h->fastListCap * 2;
        const int nqb = (h->maxKp + orbm::kMfmaRowsPerBlock - 1) / orbm::kMfmaRowsPerBlock;
        const int32_t l[K] = {(int32_t)pl, (int32_t)fl, (int32_t)(dist_lds_bytes(h->nodeCap, g.maxCellsPerLevel) + 5552), 13464, 31104, 24576};
        const int32_t t[K] = {256, 64, kDistThreads, 256, 256, 256};
        const int32_t w[K] = {h->pyrBlocks * nb, g.totalCells * nb, g.nlevels * nb, h->blurTiles.base[g.nlevels] * nb, h->kpBlocksTotal * nb, nqb * nb};
        for (int k = 0; k < K; k++) { lds_bytes[k] = l[k]; wg_threads[k] = t[k]; wgs[k] = w[k]; }
    }
    for (int k = 0; k < K; k++) if (names) names[k] = kNames[k];
    hipStream_t sa = h->streamP[0], sb = h->streamP[1];
    hipEvent_t e0, e1, eGo;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1)); HIPCHK(hipEventCreateWithFlags(&eGo, hipEventDisableTiming));
    auto timed = [&](int i, int j, int ni, int nj, float& ms) -> int {  // j < 0: alone
        HIPCHK(hipEventRecord(eGo, h->stream));
        HIPCHK(hipStreamWaitEvent(sa, eGo, 0));
        if (j >= 0) { HIPCHK(hipStreamWaitEvent(sb, eGo, 0)); for (int r = 0; r < nj / 4; r++) launch(j, sb); }  // a head start
        HIPCHK(hipEventRecord(e0, sa));
        for (int r = 0; r < ni; r++) { launch(i, sa); if (j >= 0) for (int q = 0; q * ni < nj - nj / 4 && q < 8; q++) launch(j, sb); }
        HIPCHK(hipEventRecord(e1, sa));
        HIPCHK(hipStreamSynchronize(sa)); HIPCHK(hipStreamSynchronize(sb));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
        ms = t / ni;
        return ORBX_OK;
    };
    float alone[K];
    for (int i = 0; i < K; i++) {
        float t;
        if ((rc = timed(i, -1, 4, 0, t))) return rc;                  // warm
        if ((rc = timed(i, -1, 20, 0, t))) return rc;
        alone[i] = t;
        if (alone_ms) alone_ms[i] = t;
    }
    for (int i = 0; i < K; i++)
        for (int j = 0; j < K; j++) {
            const int ni = std::max(4, (int)(target_ms / alone[i]));
            const int nj = std::max(8, (int)(4.f * ni * alone[i] / alone[j]));   // j's stream stays busy ~4x as long as i's alone time
            float t;
            if ((rc = timed(i, j, ni, nj, t))) return rc;
            if (co_ms) co_ms[i * K + j] = t;
        }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(eGo);
    return orbx_sync(h);
}

// ------------------------------------------------------------------ MapSerializer's descriptor text
// `os << pKF->mDescriptors` / `os << pMP->GetDescriptor()` into an XML attribute (src/MapSerializer.cc:344-347, 429-431):
// cv::Mat's stream operator with OpenCV 3.0's default formatter -- "[" rows "]", elements "%3d" separated by ", ", rows by
// ";\n " (restated from the published out.cpp: third-party formatting, unpinned; the reference never reads it back).
// Host-side string formatting of data that is on the host already: not a compute path, works without a device.
extern "C" int orbm_descriptors_to_text(const uint8_t* desc, int n, int cols, char* out, size_t cap, size_t* len)
{
    if (n < 0 || cols < 0 || (n > 0 && cols > 0 && !desc) || !len) return fail(ORBX_E_INVALID, "bad argument");
    const size_t need = 2 + (size_t)n * ((size_t)cols * 3 + (cols > 0 ? (size_t)(cols - 1) * 2 : 0)) + (n > 0 ? (size_t)(n - 1) * 3 : 0);
    *len = need;
    if (!out) return ORBX_OK;                       // size query
    if (cap < need + 1) return fail(ORBX_E_CAPACITY, "text needs %zu bytes", need + 1);
    char* p = out;
    *p++ = '[';
    for (int r = 0; r < n; r++) {
        if (r) { *p++ = ';'; *p++ = '\n'; *p++ = ' '; }
        for (int c = 0; c < cols; c++) {
            if (c) { *p++ = ','; *p++ = ' '; }
            const unsigned v = desc[(size_t)r * cols + c];
            p[0] = v >= 100 ? (char)('0' + v / 100) : ' ';
            p[1] = v >= 10 ? (char)('0' + (v / 10) % 10) : ' ';
            p[2] = (char)('0' + v % 10);
            p += 3;
        }
    }
    *p++ = ']';
    *p = 0;
    return ORBX_OK;
}

// the inverse (a loader the reference does not have: MultiMapper::InitFromFile never reads descriptors back)
extern "C" int orbm_descriptors_from_text(const char* text, uint8_t* desc, int cap_rows, int cols, int* n_rows)
{
    if (!text || cols < 1 || !n_rows) return fail(ORBX_E_INVALID, "bad argument");
    const char* p = text;
    while (*p == ' ' || *p == '\n') p++;
    if (*p != '[') return fail(ORBX_E_INVALID, "descriptor text does not start with '['");
    p++;
    int r = 0, c = 0;
    bool any = false;
    for (;;) {
        while (*p == ' ' || *p == '\n') p++;
        if (*p == ']') break;
        if (*p < '0' || *p > '9') return fail(ORBX_E_INVALID, "unexpected character '%c' in descriptor text", *p);
        unsigned v = 0;
        while (*p >= '0' && *p <= '9') v = v * 10 + (unsigned)(*p++ - '0');
        if (v > 255 || c >= cols) return fail(ORBX_E_INVALID, "descriptor text: value or column out of range");
        if (desc) { if (r >= cap_rows) return fail(ORBX_E_CAPACITY, "more than %d rows", cap_rows); desc[(size_t)r * cols + c] = (uint8_t)v; }
        any = true;
        c++;
        while (*p == ' ') p++;
        if (*p == ',') p++;
        else if (*p == ';') { if (c != cols) return fail(ORBX_E_INVALID, "descriptor text: short row"); p++; r++; c = 0; }
        else if (*p != ']') return fail(ORBX_E_INVALID, "descriptor text: missing separator");
    }
    if (any) { if (c != cols) return fail(ORBX_E_INVALID, "descriptor text: short row"); r++; }
    *n_rows = r;
    return ORBX_OK;
}

// ------------------------------------------------------------------ matcher handle
struct orbm_handle {
    int device = -1;
    hipStream_t stream = nullptr;
    // growable scratch
    void* d_buf[32] = {nullptr};
    size_t d_cap[32] = {0};
    // one packed upload / download per call (orbt_host.inc)
    void* h_stage = nullptr; size_t h_stageCap = 0;
    size_t ldsAttr[8] = {0};             // hipFuncAttributeMaxDynamicSharedMemorySize already raised for kernel i
    int lastRounds = 0, lastCands = 0;   // of the last projection search (orbm_last_search_stats)
};

static int orbm_reserve(orbm_handle* h, int slot, size_t bytes)
{
    if (bytes <= h->d_cap[slot]) return ORBX_OK;
    if (h->d_buf[slot]) HIPCHK(hipFree(h->d_buf[slot]));
    h->d_buf[slot] = nullptr; h->d_cap[slot] = 0;
    const size_t want = std::max<size_t>(bytes * 3 / 2, 4096);
    HIPCHK(hipMalloc(&h->d_buf[slot], want));
    h->d_cap[slot] = want;
    return ORBX_OK;
}

extern "C" int orbm_create(int device, orbm_t** out)
{
    if (!out) return fail(ORBX_E_INVALID, "null argument");
    *out = nullptr;
    int ndev = orbx_device_count();
    if (ndev == 0) return fail(ORBX_E_NO_DEVICE, "no HIP device visible: the ORB matcher has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(ORBX_E_INVALID, "device %d out of range", device);
    orbm_handle* h = new orbm_handle();
    h->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(ORBX_E_HIP, "cannot create stream on device %d", device);
    }
    live_add(h);
    *out = h;
    return ORBX_OK;
}

extern "C" void orbm_destroy(orbm_t* h)
{
    if (!h) return;
    live_remove(h);
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    for (auto p : h->d_buf) if (p) (void)hipFree(p);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    delete h;
}

static int orbm_check(orbm_handle* h)
{
    if (!h) return fail(ORBX_E_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->device));
    return ORBX_OK;
}

extern "C" int orbm_distance_matrix(orbm_t* h, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* dist)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (nq < 0 || nt < 0 || (nq && !q) || (nt && !t) || !dist) return fail(ORBX_E_INVALID, "bad argument");
    if (nq == 0 || nt == 0) return ORBX_OK;
    if ((rc = orbm_reserve(h, 0, (size_t)nq * 32)) || (rc = orbm_reserve(h, 1, (size_t)nt * 32)) ||
        (rc = orbm_reserve(h, 2, (size_t)nq * nt * 4))) return rc;
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->d_buf[0], q, (size_t)nq * 32, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->d_buf[1], t, (size_t)nt * 32, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(orbm::k_distance_matrix, dim3((nq + 255) / 256, std::min(nt, 64)), dim3(256), 0, s,
                       (const uint8_t*)h->d_buf[0], nq, (const uint8_t*)h->d_buf[1], nt, (int32_t*)h->d_buf[2]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(dist, h->d_buf[2], (size_t)nq * nt * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ORBX_OK;
}

extern "C" int orbm_match_bruteforce(orbm_t* h, const uint8_t* qdesc, const float* qangle, int nq,
                                     const uint8_t* tdesc, const float* tangle, int nt,
                                     float nnratio, int th_low, int check_ori, int32_t* match, int* nmatches)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (nq < 0 || nt < 0 || (nq && (!qdesc || !qangle || !match)) || (nt && (!tdesc || !tangle))) return fail(ORBX_E_INVALID, "bad argument");
    if (nmatches) *nmatches = 0;
    if (nq == 0) return ORBX_OK;
    // slots: 0 qdesc, 1 tdesc, 2 qangle, 3 tangle, 4 counts(2), 5 match, 6 binOf, 7 hist(32)+nmatch
    if ((rc = orbm_reserve(h, 0, (size_t)nq * 32)) || (rc = orbm_reserve(h, 1, (size_t)std::max(nt, 1) * 32)) ||
        (rc = orbm_reserve(h, 2, (size_t)nq * 4)) || (rc = orbm_reserve(h, 3, (size_t)std::max(nt, 1) * 4)) ||
        (rc = orbm_reserve(h, 4, 16)) || (rc = orbm_reserve(h, 5, (size_t)nq * 4)) || (rc = orbm_reserve(h, 6, (size_t)nq)) ||
        (rc = orbm_reserve(h, 7, 34 * 4))) return rc;
    hipStream_t s = h->stream;
    const int32_t counts[2] = {nq, nt};
    HIPCHK(hipMemcpyAsync(h->d_buf[0], qdesc, (size_t)nq * 32, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->d_buf[2], qangle, (size_t)nq * 4, hipMemcpyHostToDevice, s));
    if (nt) {
        HIPCHK(hipMemcpyAsync(h->d_buf[1], tdesc, (size_t)nt * 32, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(h->d_buf[3], tangle, (size_t)nt * 4, hipMemcpyHostToDevice, s));
    }
    HIPCHK(hipMemcpyAsync(h->d_buf[4], counts, sizeof counts, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(h->d_buf[7], 0, 34 * 4, s));
    orbm::MatchIO q{(const uint8_t*)h->d_buf[0], 0, (const float*)h->d_buf[2], 0, 1, (const int32_t*)h->d_buf[4]};
    orbm::MatchIO t{(const uint8_t*)h->d_buf[1], 0, (const float*)h->d_buf[3], 0, 1, (const int32_t*)h->d_buf[4] + 1};
    int32_t* d_hist = (int32_t*)h->d_buf[7];
    if ((rc = orbm_reserve(h, 8, (size_t)nq * kMatchChunks * sizeof(uint2)))) return rc;
    const bool mfma = nt < 65536;  // the MFMA scan packs the train index into 16 bits of its key
    orbm::AcceptArgs aa = {q, t, 0, 0, nnratio, th_low, check_ori, (int32_t*)h->d_buf[5], (int64_t)nq, (uint8_t*)h->d_buf[6], d_hist};
    if (mfma) {
        const int64_t xPitch = (int64_t)align_up(std::max(nq, nt), orbm::kMfmaRowsPerBlock) * 256;
        if ((rc = orbm_reserve(h, 9, (size_t)2 * xPitch))) return rc;
        hipLaunchKernelGGL(orbm::k_expand_desc, dim3((unsigned)(xPitch / 4096), 1), dim3(256), 0, s, q, 0, 0, (uint8_t*)h->d_buf[9], xPitch);
        hipLaunchKernelGGL(orbm::k_expand_desc, dim3((unsigned)(xPitch / 4096), 1), dim3(256), 0, s, t, 0, 1, (uint8_t*)h->d_buf[9], xPitch);
        // slot table {nq, nt}: query slot 0, train slot 1 (angles keep their own slot 0 via pitch 0)
        orbm::AcceptArgs am = aa;
        am.tslot0 = 1;
        const int nqb = (nq + orbm::kMfmaRowsPerBlock - 1) / orbm::kMfmaRowsPerBlock;
        hipLaunchKernelGGL(orbm::k_match_mfma, dim3(8 * nqb), dim3(256), 0, s, (const uint8_t*)h->d_buf[9], xPitch, am, nqb, 1, (uint2*)nullptr, (int64_t)0);
    } else {
        hipLaunchKernelGGL(orbm::k_match_best2, dim3((nq + 255) / 256, 1, kMatchChunks), dim3(256), 0, s, q, t, 0, 0, kMatchChunks,
                           (uint2*)h->d_buf[8], (int64_t)nq);
        hipLaunchKernelGGL(orbm::k_match_accept, dim3((nq + 255) / 256, 1), dim3(256), 0, s, aa, kMatchChunks, (const uint2*)h->d_buf[8], (int64_t)nq);
    }
    hipLaunchKernelGGL(orbm::k_match_prune, dim3(1), dim3(256), 0, s, q, 0, check_ori, (int32_t*)h->d_buf[5], (int64_t)nq,
                       (const uint8_t*)h->d_buf[6], d_hist, d_hist + 32);
    HIPCHK(hipGetLastError());
    int32_t nm = 0;
    HIPCHK(hipMemcpyAsync(match, h->d_buf[5], (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&nm, d_hist + 32, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (nmatches) *nmatches = nm;
    return ORBX_OK;
}

// ------------------------------------------------------------------ SearchByBoW
// one side of a BoW search, resident in HBM (node ids stay on the host: the lock-step walk happens there)
struct BowSide {
    const uint8_t* d_desc; const float* d_ang; const int32_t *d_start, *d_idx;
    const uint32_t* node_id; int n_nodes; int n;
};

static int bow_core(orbm_handle* h, const BowSide& q, const uint8_t* qvalid, const BowSide& t, const uint8_t* tvalid,
                    float nnratio, int check_ori, int out_by_train, int32_t* match, int* nmatches)
{
    int rc;
    const int nq = q.n, nt = t.n;
    const int nout = out_by_train ? nt : nq;
    // lock-step walk of the two sorted node-id lists (ORBmatcher.cc:180-266): host side,
    // it only decides WHICH node pairs are searched
    std::vector<int32_t> pq, pt;
    {
        int a = 0, b = 0;
        while (a < q.n_nodes && b < t.n_nodes) {
            if (q.node_id[a] == t.node_id[b]) { pq.push_back(a); pt.push_back(b); a++; b++; }
            else if (q.node_id[a] < t.node_id[b]) a++;
            else b++;
        }
    }
    const int npairs = (int)pq.size();
    if (npairs == 0) return ORBX_OK;
    enum { S_QV = 4, S_TV, S_PQ = 10, S_PT, S_MATCHED, S_MATCH, S_BIN, S_HIST };
    if ((rc = orbm_reserve(h, S_QV, (size_t)nq)) || (rc = orbm_reserve(h, S_TV, (size_t)nt)) || (rc = orbm_reserve(h, S_PQ, (size_t)npairs * 4)) ||
        (rc = orbm_reserve(h, S_PT, (size_t)npairs * 4)) || (rc = orbm_reserve(h, S_MATCHED, (size_t)nt)) ||
        (rc = orbm_reserve(h, S_MATCH, (size_t)nout * 4)) || (rc = orbm_reserve(h, S_BIN, (size_t)nout)) || (rc = orbm_reserve(h, S_HIST, 34 * 4))) return rc;
    hipStream_t s = h->stream;
#define UP(slot, src, bytes) HIPCHK(hipMemcpyAsync(h->d_buf[slot], src, bytes, hipMemcpyHostToDevice, s))
    if (qvalid) UP(S_QV, qvalid, (size_t)nq);
    if (tvalid) UP(S_TV, tvalid, (size_t)nt);
    UP(S_PQ, pq.data(), (size_t)npairs * 4); UP(S_PT, pt.data(), (size_t)npairs * 4);
    HIPCHK(hipMemsetAsync(h->d_buf[S_MATCHED], 0, (size_t)nt, s));
    HIPCHK(hipMemsetAsync(h->d_buf[S_MATCH], 0xFF, (size_t)nout * 4, s));
    HIPCHK(hipMemsetAsync(h->d_buf[S_HIST], 0, 34 * 4, s));
    orbm::BowArgs a{};
    a.qdesc = q.d_desc; a.qang = q.d_ang;
    a.qvalid = qvalid ? (const uint8_t*)h->d_buf[S_QV] : nullptr;
    a.tdesc = t.d_desc; a.tang = t.d_ang;
    a.tvalid = tvalid ? (const uint8_t*)h->d_buf[S_TV] : nullptr;
    a.qstart = q.d_start; a.qidx = q.d_idx;
    a.tstart = t.d_start; a.tidx = t.d_idx;
    a.pairQ = (const int32_t*)h->d_buf[S_PQ]; a.pairT = (const int32_t*)h->d_buf[S_PT];
    a.matched = (uint8_t*)h->d_buf[S_MATCHED];
    a.match = (int32_t*)h->d_buf[S_MATCH]; a.binOf = (uint8_t*)h->d_buf[S_BIN]; a.hist = (int32_t*)h->d_buf[S_HIST];
    a.nnratio = nnratio; a.thLow = 50; a.checkOri = check_ori; a.outByTrain = out_by_train;
    hipLaunchKernelGGL(orbm::k_bow_pairs, dim3(npairs), dim3(64), 0, s, a);
    hipLaunchKernelGGL(orbm::k_prune_flat, dim3(1), dim3(256), 0, s, a.match, nout, check_ori, (const uint8_t*)a.binOf,
                       a.hist, a.hist + 32);
    HIPCHK(hipGetLastError());
    int32_t nm = 0;
    HIPCHK(hipMemcpyAsync(match, a.match, (size_t)nout * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&nm, a.hist + 32, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (nmatches) *nmatches = nm;
    return ORBX_OK;
}

extern "C" int orbm_search_by_bow(orbm_t* h,
                                  const uint8_t* qdesc, const float* qangle, const uint8_t* qvalid, int nq,
                                  const OrbmFeatVec* qfv,
                                  const uint8_t* tdesc, const float* tangle, const uint8_t* tvalid, int nt,
                                  const OrbmFeatVec* tfv,
                                  float nnratio, int check_ori, int out_by_train,
                                  int32_t* match, int* nmatches)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (nq < 0 || nt < 0 || !qfv || !tfv || !match || (nq && (!qdesc || !qangle)) || (nt && (!tdesc || !tangle)))
        return fail(ORBX_E_INVALID, "bad argument");
    const int nout = out_by_train ? nt : nq;
    for (int i = 0; i < nout; i++) match[i] = -1;
    if (nmatches) *nmatches = 0;
    if (nq == 0 || nt == 0) return ORBX_OK;
    const int nqi = qfv->start[qfv->n_nodes], nti = tfv->start[tfv->n_nodes];
    for (int i = 0; i < nqi; i++) if (qfv->idx[i] < 0 || qfv->idx[i] >= nq) return fail(ORBX_E_INVALID, "query feature index out of range");
    for (int i = 0; i < nti; i++) if (tfv->idx[i] < 0 || tfv->idx[i] >= nt) return fail(ORBX_E_INVALID, "train feature index out of range");
    enum { S_QD, S_TD, S_QA, S_TA, S_QS = 6, S_QI, S_TS, S_TI };
    if ((rc = orbm_reserve(h, S_QD, (size_t)nq * 32)) || (rc = orbm_reserve(h, S_TD, (size_t)nt * 32)) || (rc = orbm_reserve(h, S_QA, (size_t)nq * 4)) ||
        (rc = orbm_reserve(h, S_TA, (size_t)nt * 4)) || (rc = orbm_reserve(h, S_QS, (size_t)(qfv->n_nodes + 1) * 4)) ||
        (rc = orbm_reserve(h, S_QI, (size_t)std::max(nqi, 1) * 4)) || (rc = orbm_reserve(h, S_TS, (size_t)(tfv->n_nodes + 1) * 4)) ||
        (rc = orbm_reserve(h, S_TI, (size_t)std::max(nti, 1) * 4))) return rc;
    hipStream_t s = h->stream;
    UP(S_QD, qdesc, (size_t)nq * 32); UP(S_TD, tdesc, (size_t)nt * 32);
    UP(S_QA, qangle, (size_t)nq * 4); UP(S_TA, tangle, (size_t)nt * 4);
    UP(S_QS, qfv->start, (size_t)(qfv->n_nodes + 1) * 4);
    if (nqi) UP(S_QI, qfv->idx, (size_t)nqi * 4);
    UP(S_TS, tfv->start, (size_t)(tfv->n_nodes + 1) * 4);
    if (nti) UP(S_TI, tfv->idx, (size_t)nti * 4);
    const BowSide q = {(const uint8_t*)h->d_buf[S_QD], (const float*)h->d_buf[S_QA], (const int32_t*)h->d_buf[S_QS], (const int32_t*)h->d_buf[S_QI],
                       qfv->node_id, qfv->n_nodes, nq};
    const BowSide t = {(const uint8_t*)h->d_buf[S_TD], (const float*)h->d_buf[S_TA], (const int32_t*)h->d_buf[S_TS], (const int32_t*)h->d_buf[S_TI],
                       tfv->node_id, tfv->n_nodes, nt};
    return bow_core(h, q, qvalid, t, tvalid, nnratio, check_ori, out_by_train, match, nmatches);
}

// ------------------------------------------------------------------ grid + SearchByProjection
enum { G_KEYS = 16, G_CNT, G_START, G_FILL, G_IDX };

// grid of n device-resident keys into (d_cnt, d_start, d_fill: ncell(+1) ints, d_idx: n ints), on stream s
static int grid_build_device(const OrbmGrid* grid, const orbm::KeyDev* dk, int n, int32_t* d_cnt, int32_t* d_start,
                             int32_t* d_fill, int32_t* d_idx, hipStream_t s, orbm::GridDev& gd)
{
    const int ncell = grid->cols * grid->rows;
    gd.minX = grid->minX; gd.minY = grid->minY; gd.invW = grid->invW; gd.invH = grid->invH; gd.cols = grid->cols; gd.rows = grid->rows;
    HIPCHK(hipMemsetAsync(d_cnt, 0, (size_t)ncell * 4, s));
    HIPCHK(hipMemsetAsync(d_fill, 0, (size_t)ncell * 4, s));
    if (n) hipLaunchKernelGGL(orbm::k_grid_count, dim3((n + 255) / 256), dim3(256), 0, s, gd, dk, n, d_cnt);
    hipLaunchKernelGGL(orbm::k_scan_small, dim3(1), dim3(1024), 0, s, (const int32_t*)d_cnt, ncell, d_start);
    if (n) {
        hipLaunchKernelGGL(orbm::k_grid_fill, dim3((n + 255) / 256), dim3(256), 0, s, gd, dk, n, (const int32_t*)d_start, d_fill, d_idx);
        hipLaunchKernelGGL(orbm::k_grid_sort, dim3((ncell + 255) / 256), dim3(256), 0, s, ncell, (const int32_t*)d_start, d_idx);
    }
    HIPCHK(hipGetLastError());
    return ORBX_OK;
}

static int orbm_build_grid(orbm_handle* h, const OrbmGrid* grid, const OrbxKeyPoint* keys, int n, orbm::GridDev& gd)
{
    if (!grid || grid->cols < 1 || grid->rows < 1 || grid->cols * grid->rows > (1 << 20)) return fail(ORBX_E_INVALID, "bad grid");
    int rc;
    const int ncell = grid->cols * grid->rows;
    if ((rc = orbm_reserve(h, G_KEYS, (size_t)std::max(n, 1) * sizeof(OrbxKeyPoint))) || (rc = orbm_reserve(h, G_CNT, (size_t)ncell * 4)) ||
        (rc = orbm_reserve(h, G_START, (size_t)(ncell + 1) * 4)) || (rc = orbm_reserve(h, G_FILL, (size_t)ncell * 4)) ||
        (rc = orbm_reserve(h, G_IDX, (size_t)std::max(n, 1) * 4))) return rc;
    hipStream_t s = h->stream;
    if (n) HIPCHK(hipMemcpyAsync(h->d_buf[G_KEYS], keys, (size_t)n * sizeof(OrbxKeyPoint), hipMemcpyHostToDevice, s));
    return grid_build_device(grid, (const orbm::KeyDev*)h->d_buf[G_KEYS], n, (int32_t*)h->d_buf[G_CNT], (int32_t*)h->d_buf[G_START],
                             (int32_t*)h->d_buf[G_FILL], (int32_t*)h->d_buf[G_IDX], s, gd);
}

extern "C" int orbm_features_in_area(orbm_t* h, const OrbmGrid* grid, const OrbxKeyPoint* keys_un, int n,
                                     float x, float y, float r, int minLevel, int maxLevel,
                                     int32_t* out, int cap, int* n_out)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (n < 0 || (n && !keys_un) || cap < 0 || (cap && !out)) return fail(ORBX_E_INVALID, "bad argument");
    orbm::GridDev gd;
    if ((rc = orbm_build_grid(h, grid, keys_un, n, gd))) return rc;
    if ((rc = orbm_reserve(h, 0, (size_t)std::max(cap, 1) * 4)) || (rc = orbm_reserve(h, 1, 16))) return rc;
    hipStream_t s = h->stream;
    hipLaunchKernelGGL(orbm::k_features_in_area, dim3(1), dim3(1), 0, s, gd, (const orbm::KeyDev*)h->d_buf[G_KEYS],
                       (const int32_t*)h->d_buf[G_START], (const int32_t*)h->d_buf[G_IDX], x, y, r, minLevel, maxLevel,
                       (int32_t*)h->d_buf[0], cap, (int32_t*)h->d_buf[1]);
    HIPCHK(hipGetLastError());
    int32_t cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, h->d_buf[1], 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (n_out) *n_out = cnt;
    if (cnt > cap) return fail(ORBX_E_CAPACITY, "%d features in area, capacity %d", cnt, cap);
    if (cnt) HIPCHK(hipMemcpy(out, h->d_buf[0], (size_t)cnt * 4, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

#include "orbt_host.inc"

static int proj_check(orbm_handle* h, const OrbmProjParams* pp, const float* q_uvr, const int8_t* q_lvl, const uint8_t* qdesc,
                      const float* qangle, int nq, int nt, const uint8_t* t_occ, const int32_t* assign, int* nmatches)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (!pp || pp->mode < 3 || pp->mode > 6) return fail(ORBX_E_INVALID, "mode must be 3..6");
    if (nq < 0 || nt < 0 || (nq && (!q_uvr || !q_lvl || !qdesc)) || (nt && (!t_occ || !assign))) return fail(ORBX_E_INVALID, "bad argument");
    const bool useRot = pp->check_ori && (pp->mode == 4 || pp->mode == 5);
    if (useRot && nq && !qangle) return fail(ORBX_E_INVALID, "angles required for the rotation check");
    if (nmatches) *nmatches = 0;
    return ORBX_OK;
}

extern "C" int orbm_search_by_projection(orbm_t* h, const OrbmProjParams* pp,
                                         const float* q_uvr, const int8_t* q_lvl,
                                         const uint8_t* qdesc, const float* qangle,
                                         const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                                         const OrbmGrid* grid, const OrbxKeyPoint* t_keys_un,
                                         const uint8_t* tdesc, int nt,
                                         uint8_t* t_occ, int32_t* assign, int* nmatches)
{
    return orbm_search_by_projection_stereo(h, pp, q_uvr, nullptr, q_lvl, qdesc, qangle, qvalid, q_obs_pos, nq, grid, t_keys_un,
                                            tdesc, nullptr, nt, t_occ, assign, nmatches);
}

extern "C" int orbm_search_by_projection_stereo(orbm_t* h, const OrbmProjParams* pp,
                                                const float* q_uvr, const float* q_ur, const int8_t* q_lvl,
                                                const uint8_t* qdesc, const float* qangle,
                                                const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                                                const OrbmGrid* grid, const OrbxKeyPoint* t_keys_un,
                                                const uint8_t* tdesc, const float* t_uright, int nt,
                                                uint8_t* t_occ, int32_t* assign, int* nmatches)
{
    int rc = proj_check(h, pp, q_uvr, q_lvl, qdesc, qangle, nq, nt, t_occ, assign, nmatches);
    if (rc) return rc;
    if (nt && (!t_keys_un || !tdesc)) return fail(ORBX_E_INVALID, "bad argument");
    if ((q_ur != nullptr) != (t_uright != nullptr)) return fail(ORBX_E_INVALID, "q_ur and t_uright go together");
    if (q_ur && pp->mode != 3 && pp->mode != 4) return fail(ORBX_E_INVALID, "only modes 3 and 4 have a stereo gate");
    if (nq == 0 || nt == 0) return ORBX_OK;
    if (!grid || grid->cols < 1 || grid->rows < 1) return fail(ORBX_E_INVALID, "bad grid");
    const ProjTrainHost th = {grid, t_keys_un, tdesc};
    return proj_core(h, pp, q_uvr, q_lvl, qdesc, qangle, qvalid, q_obs_pos, nq, nullptr, &th, nt, t_occ, assign, nmatches, q_ur, t_uright);
}

// ------------------------------------------------------------------ the Frame's matcher-side state in HBM
struct orbm_frame {
    orbm_handle* owner = nullptr;
    int n = 0;
    orbm::GridDev gd{};
    orbm::KeyDev* d_keysUn = nullptr;
    uint8_t* d_desc = nullptr;
    int32_t *d_cnt = nullptr, *d_start = nullptr, *d_fill = nullptr, *d_idx = nullptr;
    uint4* d_rec = nullptr;                  // the features in grid order (orbt::FrameSetDev::rec), read by the projection searches
    int32_t* d_n = nullptr;
    float* d_ang = nullptr;                  // mvKeysUn[i].angle, contiguous (the BoW search reads angles by feature index)
    bool hasBow = false;                     // orbm_frame_compute_bow ran: FeatureVector as CSR, node ids on the host
    std::vector<uint32_t> fvNode;
    int fvNodes = 0;
    int32_t *d_fvStart = nullptr, *d_fvIdx = nullptr;
};

extern "C" int orbm_frame_destroy(orbm_frame_t* f)
{
    if (!f) return ORBX_OK;
    if (f->owner && f->owner->device >= 0) {
        (void)hipSetDevice(f->owner->device);
        (void)hipStreamSynchronize(f->owner->stream);
        void* ptrs[] = {f->d_keysUn, f->d_desc, f->d_cnt, f->d_start, f->d_fill, f->d_idx, f->d_ang, f->d_fvStart, f->d_fvIdx, f->d_rec, f->d_n};
        for (void* p : ptrs) if (p) (void)hipFree(p);
    }
    delete f;
    return ORBX_OK;
}

extern "C" int orbm_frame_create(orbm_t* h, const OrbxKeyPoint* d_keys, const uint8_t* d_desc, int n,
                                 const float K[4], const float D[5], const OrbmGrid* grid, orbm_frame_t** out)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (!out || n < 0 || (n && (!d_keys || !d_desc)) || !K || !D) return fail(ORBX_E_INVALID, "bad argument");
    if (!grid || grid->cols < 1 || grid->rows < 1 || grid->cols * grid->rows > (1 << 20)) return fail(ORBX_E_INVALID, "bad grid");
    *out = nullptr;
    orbm_frame* f = new orbm_frame;
    f->owner = h; f->n = n;
    const int ncell = grid->cols * grid->rows, nn = std::max(n, 1);
#define FCR(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { int r_ = fail(ORBX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); orbm_frame_destroy(f); return r_; } } while (0)
    FCR(hipMalloc(&f->d_keysUn, (size_t)nn * sizeof(OrbxKeyPoint)));
    FCR(hipMalloc(&f->d_desc, (size_t)nn * 32));
    FCR(hipMalloc(&f->d_cnt, (size_t)ncell * 4));
    FCR(hipMalloc(&f->d_start, (size_t)(ncell + 1) * 4));
    FCR(hipMalloc(&f->d_fill, (size_t)ncell * 4));
    FCR(hipMalloc(&f->d_idx, (size_t)nn * 4));
    FCR(hipMalloc(&f->d_ang, (size_t)nn * 4));
    hipStream_t s = h->stream;
    FCR(hipMalloc(&f->d_rec, (size_t)nn * 16));
    FCR(hipMalloc(&f->d_n, 16));
    if (ncell <= kProjMaxCells && n <= 16384 && !((uintptr_t)d_desc & 15)) {
        // one launch: mvKeysUn, angles, descriptors, grid, grid-ordered records
        orbt::FrameBuildArgs fa{};
        fa.fs = {f->d_keysUn, f->d_desc, f->d_ang, f->d_start, f->d_idx, f->d_rec, f->d_n, nn, ncell};
        fa.srcKeys = (const orbm::KeyDev*)d_keys; fa.srcDesc = d_desc; fa.srcCount = nullptr; fa.srcCap = nn; fa.srcN = n;
        fa.slot0 = 0; fa.slotMod = 1;
        fa.grid = {grid->minX, grid->minY, grid->invW, grid->invH, grid->cols, grid->rows};
        fa.und = {K[0], K[1], K[2], K[3], D[0], D[1], D[2], D[3], D[4]};
        fa.undistort = D[0] != 0.0f;  // mvKeysUn = mvKeys otherwise (Frame.cc:406-410)
        f->gd = fa.grid;
        if ((rc = frame_build_launch(h, fa, 1))) { orbm_frame_destroy(f); return rc; }
    } else {
        if (n) {
            FCR(hipMemcpyAsync(f->d_desc, d_desc, (size_t)n * 32, hipMemcpyDeviceToDevice, s));
            if (D[0] == 0.0f) {  // mvKeysUn = mvKeys (Frame.cc:406-410)
                FCR(hipMemcpyAsync(f->d_keysUn, d_keys, (size_t)n * sizeof(OrbxKeyPoint), hipMemcpyDeviceToDevice, s));
            } else {
                orbm::UndistArgs a = {K[0], K[1], K[2], K[3], D[0], D[1], D[2], D[3], D[4]};
                hipLaunchKernelGGL(orbm::k_undistort, dim3((n + 255) / 256), dim3(256), 0, s, (const orbm::KeyDev*)d_keys, n, a, f->d_keysUn);
            }
            FCR(hipMemcpy2DAsync(f->d_ang, 4, (const uint8_t*)d_keys + 12, sizeof(OrbxKeyPoint), 4, (size_t)n, hipMemcpyDeviceToDevice, s));
        }
        FCR(hipFree(f->d_rec));  // a grid this large has no LDS-resident form: no projection search on this frame
        f->d_rec = nullptr;
        if ((rc = grid_build_device(grid, f->d_keysUn, n, f->d_cnt, f->d_start, f->d_fill, f->d_idx, s, f->gd))) { orbm_frame_destroy(f); return rc; }
    }
#undef FCR
    if (hipStreamSynchronize(s) != hipSuccess) { orbm_frame_destroy(f); return fail(ORBX_E_HIP, "frame construction failed"); }
    *out = f;
    return ORBX_OK;
}

extern "C" int orbm_frame_size(const orbm_frame_t* f) { return f ? f->n : 0; }

extern "C" int orbm_frame_download_keys_un(orbm_frame_t* f, OrbxKeyPoint* keys_un)
{
    if (!f || !f->owner) return fail(ORBX_E_INVALID, "null frame");
    int rc = orbm_check(f->owner);
    if (rc) return rc;
    if (f->n && !keys_un) return fail(ORBX_E_INVALID, "bad argument");
    if (f->n) HIPCHK(hipMemcpy(keys_un, f->d_keysUn, (size_t)f->n * sizeof(OrbxKeyPoint), hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbm_search_by_projection_frame(orbm_t* h, const OrbmProjParams* pp,
                                               const float* q_uvr, const int8_t* q_lvl,
                                               const uint8_t* qdesc, const float* qangle,
                                               const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                                               orbm_frame_t* train, uint8_t* t_occ, int32_t* assign, int* nmatches)
{
    if (!train || train->owner != h) return fail(ORBX_E_INVALID, "frame does not belong to this matcher handle");
    const int nt = train->n;
    int rc = proj_check(h, pp, q_uvr, q_lvl, qdesc, qangle, nq, nt, t_occ, assign, nmatches);
    if (rc) return rc;
    if (nq == 0 || nt == 0) return ORBX_OK;
    if (!train->d_rec) return fail(ORBX_E_UNSUPPORTED, "the frame's grid is too large for the projection search");
    const ProjTrain tr = {train->gd, train->d_keysUn, train->d_start, train->d_idx, train->d_desc, nt, train->d_rec};
    return proj_core(h, pp, q_uvr, q_lvl, qdesc, qangle, qvalid, q_obs_pos, nq, &tr, nullptr, nt, t_occ, assign, nmatches);
}

// train side of a windowed best search, resident in HBM
static int window_core(orbm_handle* h, const float* q_uvr, const float* q_ur, const int8_t* q_pred, const uint8_t* qdesc,
                       const uint8_t* qvalid, int nq, const ProjTrain& tr, const float* d_turight, const float* inv_sigma2,
                       int nlevels, int chi2, int32_t* best_idx, int32_t* best_dist)
{
    int rc;
    enum { S_UVR, S_UR, S_PRED, S_QD, S_QV, S_TD, S_TUR, S_SIG, S_BI, S_BD };
    const size_t sizes[] = {(size_t)nq * 12, (size_t)nq * 4, (size_t)nq, (size_t)nq * 32, (size_t)nq, 16, 16, 64, (size_t)nq * 4, (size_t)nq * 4};
    for (int i = 0; i < 10; i++) if (i != S_TD && i != S_TUR && (rc = orbm_reserve(h, i, sizes[i]))) return rc;
    hipStream_t s = h->stream;
    UP(S_UVR, q_uvr, (size_t)nq * 12); UP(S_PRED, q_pred, (size_t)nq); UP(S_QD, qdesc, (size_t)nq * 32);
    if (q_ur) UP(S_UR, q_ur, (size_t)nq * 4);
    if (qvalid) UP(S_QV, qvalid, (size_t)nq);
    if (chi2) UP(S_SIG, inv_sigma2, (size_t)nlevels * 4);
    orbm::WinArgs a{};
    a.grid = tr.gd;
    a.tkeys = tr.keys;
    a.cellStart = tr.cellStart; a.cellIdx = tr.cellIdx;
    a.quvr = (const float*)h->d_buf[S_UVR]; a.qur = q_ur ? (const float*)h->d_buf[S_UR] : nullptr;
    a.qpred = (const int8_t*)h->d_buf[S_PRED]; a.qdesc = (const uint8_t*)h->d_buf[S_QD];
    a.qvalid = qvalid ? (const uint8_t*)h->d_buf[S_QV] : nullptr;
    a.tdesc = tr.desc; a.turight = d_turight;
    a.invSigma2 = (const float*)h->d_buf[S_SIG];
    a.nq = nq; a.chi2 = chi2;
    a.bestIdx = (int32_t*)h->d_buf[S_BI]; a.bestDist = (int32_t*)h->d_buf[S_BD];
    hipLaunchKernelGGL(orbm::k_window_best, dim3((nq + 63) / 64), dim3(64), 0, s, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(best_idx, a.bestIdx, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(best_dist, a.bestDist, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ORBX_OK;
}

static int window_check(orbm_handle* h, const float* q_uvr, const float* q_ur, const int8_t* q_pred, const uint8_t* qdesc, int nq, int nt,
                        const float* t_uright, const float* inv_sigma2, int nlevels, int chi2, int32_t* best_idx, int32_t* best_dist)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (nq < 0 || nt < 0 || (nq && (!q_uvr || !q_pred || !qdesc || !best_idx || !best_dist))) return fail(ORBX_E_INVALID, "bad argument");
    if (chi2 && (!inv_sigma2 || nlevels < 1 || nlevels > 16)) return fail(ORBX_E_INVALID, "inv_sigma2 required for the chi-square test");
    if (chi2 && t_uright && !q_ur) return fail(ORBX_E_INVALID, "q_ur required with t_uright");
    for (int i = 0; i < nq; i++) { best_idx[i] = -1; best_dist[i] = 256; }
    return ORBX_OK;
}

// ------------------------------------------------------------------ SURVEY 8(f).1 entry points
extern "C" int orbm_window_best(orbm_t* h, const float* q_uvr, const float* q_ur, const int8_t* q_pred,
                                const uint8_t* qdesc, const uint8_t* qvalid, int nq,
                                const OrbmGrid* grid, const OrbxKeyPoint* t_keys_un, const uint8_t* tdesc,
                                const float* t_uright, int nt, const float* inv_sigma2, int nlevels, int chi2,
                                int32_t* best_idx, int32_t* best_dist)
{
    int rc = window_check(h, q_uvr, q_ur, q_pred, qdesc, nq, nt, t_uright, inv_sigma2, nlevels, chi2, best_idx, best_dist);
    if (rc) return rc;
    if (nt && (!t_keys_un || !tdesc)) return fail(ORBX_E_INVALID, "bad argument");
    if (nq == 0 || nt == 0) return ORBX_OK;
    ProjTrain tr;
    if ((rc = orbm_build_grid(h, grid, t_keys_un, nt, tr.gd))) return rc;
    enum { S_TD = 5, S_TUR = 6 };
    if ((rc = orbm_reserve(h, S_TD, (size_t)nt * 32)) || (rc = orbm_reserve(h, S_TUR, (size_t)nt * 4))) return rc;
    hipStream_t s = h->stream;
    UP(S_TD, tdesc, (size_t)nt * 32);
    if (t_uright) UP(S_TUR, t_uright, (size_t)nt * 4);
    tr.keys = (const orbm::KeyDev*)h->d_buf[G_KEYS];
    tr.cellStart = (const int32_t*)h->d_buf[G_START]; tr.cellIdx = (const int32_t*)h->d_buf[G_IDX];
    tr.desc = (const uint8_t*)h->d_buf[S_TD];
    tr.nt = nt;
    return window_core(h, q_uvr, q_ur, q_pred, qdesc, qvalid, nq, tr, t_uright ? (const float*)h->d_buf[S_TUR] : nullptr, inv_sigma2,
                       nlevels, chi2, best_idx, best_dist);
}

// the same with a device-resident frame as train side (the KeyFrame of Fuse / SearchBySim3; mono: no right coordinates)
extern "C" int orbm_window_best_frame(orbm_t* h, const float* q_uvr, const int8_t* q_pred, const uint8_t* qdesc, const uint8_t* qvalid, int nq,
                                      orbm_frame_t* train, const float* inv_sigma2, int nlevels, int chi2,
                                      int32_t* best_idx, int32_t* best_dist)
{
    if (!train || train->owner != h) return fail(ORBX_E_INVALID, "frame does not belong to this matcher handle");
    int rc = window_check(h, q_uvr, nullptr, q_pred, qdesc, nq, train->n, nullptr, inv_sigma2, nlevels, chi2, best_idx, best_dist);
    if (rc) return rc;
    if (nq == 0 || train->n == 0) return ORBX_OK;
    ProjTrain tr = {train->gd, train->d_keysUn, train->d_start, train->d_idx, train->d_desc, train->n};
    return window_core(h, q_uvr, nullptr, q_pred, qdesc, qvalid, nq, tr, nullptr, inv_sigma2, nlevels, chi2, best_idx, best_dist);
}

// query side (F1) of SearchForInitialization resident in HBM: descriptors, angles, "octave 0" flags
static int init_core(orbm_handle* h, const float* q_xy, float window_size, const uint8_t* d_qdesc, const float* d_qang,
                     const uint8_t* d_qvalid, int nq, const ProjTrain& tr, float nnratio, int check_ori, int32_t* matches12, int* nmatches)
{
    int rc;
    const int nt = tr.nt;
    if ((size_t)nt * 2 > 150 * 1024) return fail(ORBX_E_UNSUPPORTED, "too many train features for the LDS distance table");
    // flatten: window query (x, y, windowSize), levels (0, 0); only octave-0 queries search (:424-426)
    std::vector<float> uvr((size_t)nq * 3);
    std::vector<int8_t> lvl((size_t)nq * 2, 0);
    for (int i = 0; i < nq; i++) { uvr[3 * i] = q_xy[2 * i]; uvr[3 * i + 1] = q_xy[2 * i + 1]; uvr[3 * i + 2] = window_size; }
    enum { S_UVR, S_LVL, S_CNT = 6, S_OFF, S_KEY, S_CIDX, S_M12, S_M21, S_NM, S_PUSHT, S_PUSHB };
    const int slots[] = {S_UVR, S_LVL, S_CNT, S_OFF, S_KEY, S_CIDX, S_M12, S_M21, S_NM, S_PUSHT, S_PUSHB};
    const size_t sizes[] = {(size_t)nq * 12, (size_t)nq * 2, (size_t)nq * 4, (size_t)(nq + 1) * 4, 16, 16, (size_t)nq * 4, (size_t)nt * 4, 16,
                            (size_t)nq * 4, (size_t)nq};
    for (int i = 0; i < 11; i++) if ((rc = orbm_reserve(h, slots[i], sizes[i]))) return rc;
    hipStream_t s = h->stream;
    UP(S_UVR, uvr.data(), (size_t)nq * 12); UP(S_LVL, lvl.data(), (size_t)nq * 2);
    orbm::ProjArgs a{};
    a.grid = tr.gd;
    a.tkeys = tr.keys;
    a.cellStart = tr.cellStart; a.cellIdx = tr.cellIdx;
    a.quvr = (const float*)h->d_buf[S_UVR]; a.qlvl = (const int8_t*)h->d_buf[S_LVL];
    a.qdesc = d_qdesc; a.qang = d_qang;
    a.qvalid = d_qvalid; a.qobs = nullptr;
    a.tdesc = tr.desc;
    a.nq = nq; a.nt = nt;
    a.candCnt = (int32_t*)h->d_buf[S_CNT]; a.candOff = (int32_t*)h->d_buf[S_OFF];
    a.candKey = nullptr; a.candIdx = nullptr;
    a.tocc = nullptr; a.assign = nullptr; a.nmatch = (int32_t*)h->d_buf[S_NM];
    a.pushT = (int32_t*)h->d_buf[S_PUSHT]; a.pushBin = (uint8_t*)h->d_buf[S_PUSHB];
    a.mode = 7; a.nnratio = nnratio; a.checkOri = check_ori; a.thDist = 50;
    hipLaunchKernelGGL(orbm::k_proj_candidates, dim3((nq + 63) / 64), dim3(64), 0, s, a, 0);
    hipLaunchKernelGGL(orbm::k_scan_small, dim3(1), dim3(1024), 0, s, (const int32_t*)a.candCnt, nq, a.candOff);
    HIPCHK(hipGetLastError());
    int32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, a.candOff + nq, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if ((rc = orbm_reserve(h, S_KEY, (size_t)std::max(total, 1) * 4)) || (rc = orbm_reserve(h, S_CIDX, (size_t)std::max(total, 1) * 4))) return rc;
    a.candKey = (uint32_t*)h->d_buf[S_KEY]; a.candIdx = (int32_t*)h->d_buf[S_CIDX];
    hipLaunchKernelGGL(orbm::k_proj_candidates, dim3((nq + 63) / 64), dim3(64), 0, s, a, 1);
    const size_t lds = ((size_t)nt * 2 + 3) & ~(size_t)3;
    if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)orbm::k_init_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(orbm::k_init_resolve, dim3(1), dim3(64), lds, s, a, (int32_t*)h->d_buf[S_M12], (int32_t*)h->d_buf[S_M21]);
    HIPCHK(hipGetLastError());
    int32_t nm = 0;
    HIPCHK(hipMemcpyAsync(matches12, h->d_buf[S_M12], (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&nm, a.nmatch, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (nmatches) *nmatches = nm;
    return ORBX_OK;
}

extern "C" int orbm_search_for_initialization(orbm_t* h, const float* q_xy, float window_size,
                                              const OrbxKeyPoint* q_keys_un, const uint8_t* qdesc, int nq,
                                              const OrbmGrid* grid, const OrbxKeyPoint* t_keys_un, const uint8_t* tdesc, int nt,
                                              float nnratio, int check_ori, int32_t* matches12, int* nmatches)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (nq < 0 || nt < 0 || (nq && (!q_xy || !q_keys_un || !qdesc || !matches12)) || (nt && (!t_keys_un || !tdesc)))
        return fail(ORBX_E_INVALID, "bad argument");
    for (int i = 0; i < nq; i++) matches12[i] = -1;
    if (nmatches) *nmatches = 0;
    if (nq == 0 || nt == 0) return ORBX_OK;
    ProjTrain tr;
    if ((rc = orbm_build_grid(h, grid, t_keys_un, nt, tr.gd))) return rc;
    std::vector<float> ang(nq);
    std::vector<uint8_t> valid(nq);
    for (int i = 0; i < nq; i++) { valid[i] = q_keys_un[i].octave <= 0; ang[i] = q_keys_un[i].angle; }
    enum { S_QD = 2, S_QA, S_QV, S_TD };
    if ((rc = orbm_reserve(h, S_QD, (size_t)nq * 32)) || (rc = orbm_reserve(h, S_QA, (size_t)nq * 4)) || (rc = orbm_reserve(h, S_QV, (size_t)nq)) ||
        (rc = orbm_reserve(h, S_TD, (size_t)nt * 32))) return rc;
    hipStream_t s = h->stream;
    UP(S_QD, qdesc, (size_t)nq * 32); UP(S_QA, ang.data(), (size_t)nq * 4); UP(S_QV, valid.data(), (size_t)nq); UP(S_TD, tdesc, (size_t)nt * 32);
    tr.keys = (const orbm::KeyDev*)h->d_buf[G_KEYS];
    tr.cellStart = (const int32_t*)h->d_buf[G_START]; tr.cellIdx = (const int32_t*)h->d_buf[G_IDX];
    tr.desc = (const uint8_t*)h->d_buf[S_TD];
    tr.nt = nt;
    return init_core(h, q_xy, window_size, (const uint8_t*)h->d_buf[S_QD], (const float*)h->d_buf[S_QA], (const uint8_t*)h->d_buf[S_QV], nq, tr,
                     nnratio, check_ori, matches12, nmatches);
}

/* SearchForInitialization between two device-resident frames (F1 = the initial frame, F2 = the current one) */
extern "C" int orbm_search_for_initialization_frames(orbm_t* h, const float* q_xy, float window_size, orbm_frame_t* f1, orbm_frame_t* f2,
                                                     float nnratio, int check_ori, int32_t* matches12, int* nmatches)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (!f1 || !f2 || f1->owner != h || f2->owner != h) return fail(ORBX_E_INVALID, "frames do not belong to this matcher handle");
    const int nq = f1->n, nt = f2->n;
    if (nq && (!q_xy || !matches12)) return fail(ORBX_E_INVALID, "bad argument");
    for (int i = 0; i < nq; i++) matches12[i] = -1;
    if (nmatches) *nmatches = 0;
    if (nq == 0 || nt == 0) return ORBX_OK;
    if ((rc = orbm_reserve(h, 4, (size_t)nq))) return rc;
    hipLaunchKernelGGL(orbm::k_octave0_flags, dim3((nq + 255) / 256), dim3(256), 0, h->stream, (const orbm::KeyDev*)f1->d_keysUn, nq, (uint8_t*)h->d_buf[4]);
    const ProjTrain tr = {f2->gd, f2->d_keysUn, f2->d_start, f2->d_idx, f2->d_desc, nt};
    return init_core(h, q_xy, window_size, f1->d_desc, f1->d_ang, (const uint8_t*)h->d_buf[4], nq, tr, nnratio, check_ori, matches12, nmatches);
}

// one side of SearchForTriangulation resident in HBM (node ids on the host for the lock-step walk)
struct TriSide {
    const orbm::KeyDev* keys; const uint8_t* desc; int n;
    const int32_t* d_start; const int32_t* d_idx; const uint32_t* node_id; int n_nodes;
    const uint8_t* skip; const float* uright;   // host arrays (uploaded here), may be null
};

static int tri_core(orbm_handle* h, const TriSide& A, const TriSide& B, const float F12[9], float ex, float ey, const float* sf2,
                    const float* sigma2_2, int nlevels, int only_stereo, int check_ori, int32_t* matches12, int* nmatches)
{
    int rc;
    const int n1 = A.n, n2 = B.n;
    std::vector<int32_t> pa, pb;
    {
        int a = 0, b = 0;
        while (a < A.n_nodes && b < B.n_nodes) {
            if (A.node_id[a] == B.node_id[b]) { pa.push_back(a); pb.push_back(b); a++; b++; }
            else if (A.node_id[a] < B.node_id[b]) a++;
            else b++;
        }
    }
    const int npairs = (int)pa.size();
    if (npairs == 0) return ORBX_OK;
    enum { S_S1 = 2, S_U1 = 3, S_S2 = 6, S_U2 = 7, S_PA = 12, S_PB = 13, S_M12 = 14, S_BIN = 15, S_HIST = 23 };
    const int slots[] = {S_S1, S_U1, S_S2, S_U2, S_PA, S_PB, S_M12, S_BIN, S_HIST};
    const size_t sizes[] = {(size_t)n1, (size_t)n1 * 4, (size_t)n2, (size_t)n2 * 4, (size_t)npairs * 4, (size_t)npairs * 4, (size_t)n1 * 4, (size_t)n1, 34 * 4};
    for (int i = 0; i < 9; i++) if ((rc = orbm_reserve(h, slots[i], sizes[i]))) return rc;
    hipStream_t s = h->stream;
    if (A.skip) UP(S_S1, A.skip, (size_t)n1);
    if (B.skip) UP(S_S2, B.skip, (size_t)n2);
    if (A.uright) UP(S_U1, A.uright, (size_t)n1 * 4);
    if (B.uright) UP(S_U2, B.uright, (size_t)n2 * 4);
    UP(S_PA, pa.data(), (size_t)npairs * 4); UP(S_PB, pb.data(), (size_t)npairs * 4);
    HIPCHK(hipMemsetAsync(h->d_buf[S_M12], 0xFF, (size_t)n1 * 4, s));
    HIPCHK(hipMemsetAsync(h->d_buf[S_HIST], 0, 34 * 4, s));
    orbm::TriArgs a{};
    a.k1 = A.keys; a.d1 = A.desc;
    a.skip1 = A.skip ? (const uint8_t*)h->d_buf[S_S1] : nullptr; a.ur1 = A.uright ? (const float*)h->d_buf[S_U1] : nullptr;
    a.k2 = B.keys; a.d2 = B.desc;
    a.skip2 = B.skip ? (const uint8_t*)h->d_buf[S_S2] : nullptr; a.ur2 = B.uright ? (const float*)h->d_buf[S_U2] : nullptr;
    a.start1 = A.d_start; a.idx1 = A.d_idx; a.start2 = B.d_start; a.idx2 = B.d_idx;
    a.pairA = (const int32_t*)h->d_buf[S_PA]; a.pairB = (const int32_t*)h->d_buf[S_PB];
    for (int i = 0; i < 9; i++) a.F[i] = F12[i];
    a.ex = ex; a.ey = ey;
    for (int i = 0; i < 16; i++) { a.sf2[i] = i < nlevels ? sf2[i] : 0.f; a.sigma2[i] = i < nlevels ? sigma2_2[i] : 0.f; }
    a.onlyStereo = only_stereo; a.checkOri = check_ori;
    a.m12 = (int32_t*)h->d_buf[S_M12]; a.binOf = (uint8_t*)h->d_buf[S_BIN]; a.hist = (int32_t*)h->d_buf[S_HIST];
    hipLaunchKernelGGL(orbm::k_triangulation_pairs, dim3(npairs), dim3(64), 0, s, a);
    hipLaunchKernelGGL(orbm::k_prune_flat, dim3(1), dim3(256), 0, s, a.m12, n1, check_ori, (const uint8_t*)a.binOf, a.hist, a.hist + 32);
    HIPCHK(hipGetLastError());
    int32_t nm = 0;
    HIPCHK(hipMemcpyAsync(matches12, a.m12, (size_t)n1 * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&nm, a.hist + 32, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (nmatches) *nmatches = nm;
    return ORBX_OK;
}

extern "C" int orbm_search_for_triangulation(orbm_t* h,
                                             const OrbxKeyPoint* k1, const uint8_t* d1, const uint8_t* skip1, const float* uright1, int n1,
                                             const OrbmFeatVec* fv1,
                                             const OrbxKeyPoint* k2, const uint8_t* d2, const uint8_t* skip2, const float* uright2, int n2,
                                             const OrbmFeatVec* fv2,
                                             const float F12[9], float ex, float ey, const float* sf2, const float* sigma2_2, int nlevels,
                                             int only_stereo, int check_ori, int32_t* matches12, int* nmatches)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (n1 < 0 || n2 < 0 || !fv1 || !fv2 || !F12 || !sf2 || !sigma2_2 || nlevels < 1 || nlevels > 16 ||
        (n1 && (!k1 || !d1 || !matches12)) || (n2 && (!k2 || !d2)))
        return fail(ORBX_E_INVALID, "bad argument");
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (nmatches) *nmatches = 0;
    if (n1 == 0 || n2 == 0) return ORBX_OK;
    const int ni1 = fv1->start[fv1->n_nodes], ni2 = fv2->start[fv2->n_nodes];
    for (int i = 0; i < ni1; i++) if (fv1->idx[i] < 0 || fv1->idx[i] >= n1) return fail(ORBX_E_INVALID, "feature index out of range");
    for (int i = 0; i < ni2; i++) if (fv2->idx[i] < 0 || fv2->idx[i] >= n2) return fail(ORBX_E_INVALID, "feature index out of range");
    enum { S_K1 = 0, S_D1 = 1, S_K2 = 4, S_D2 = 5, S_ST1 = 8, S_I1 = 9, S_ST2 = 10, S_I2 = 11 };
    const int slots[] = {S_K1, S_D1, S_K2, S_D2, S_ST1, S_I1, S_ST2, S_I2};
    const size_t sizes[] = {(size_t)n1 * 28, (size_t)n1 * 32, (size_t)n2 * 28, (size_t)n2 * 32, (size_t)(fv1->n_nodes + 1) * 4,
                            (size_t)std::max(ni1, 1) * 4, (size_t)(fv2->n_nodes + 1) * 4, (size_t)std::max(ni2, 1) * 4};
    for (int i = 0; i < 8; i++) if ((rc = orbm_reserve(h, slots[i], sizes[i]))) return rc;
    hipStream_t s = h->stream;
    UP(S_K1, k1, (size_t)n1 * 28); UP(S_D1, d1, (size_t)n1 * 32); UP(S_K2, k2, (size_t)n2 * 28); UP(S_D2, d2, (size_t)n2 * 32);
    UP(S_ST1, fv1->start, (size_t)(fv1->n_nodes + 1) * 4);
    if (ni1) UP(S_I1, fv1->idx, (size_t)ni1 * 4);
    UP(S_ST2, fv2->start, (size_t)(fv2->n_nodes + 1) * 4);
    if (ni2) UP(S_I2, fv2->idx, (size_t)ni2 * 4);
    const TriSide A = {(const orbm::KeyDev*)h->d_buf[S_K1], (const uint8_t*)h->d_buf[S_D1], n1, (const int32_t*)h->d_buf[S_ST1],
                       (const int32_t*)h->d_buf[S_I1], fv1->node_id, fv1->n_nodes, skip1, uright1};
    const TriSide B = {(const orbm::KeyDev*)h->d_buf[S_K2], (const uint8_t*)h->d_buf[S_D2], n2, (const int32_t*)h->d_buf[S_ST2],
                       (const int32_t*)h->d_buf[S_I2], fv2->node_id, fv2->n_nodes, skip2, uright2};
    return tri_core(h, A, B, F12, ex, ey, sf2, sigma2_2, nlevels, only_stereo, check_ori, matches12, nmatches);
}

// SearchForTriangulation between two device-resident frames that ran orbm_frame_compute_bow (mono: no right coordinates)
extern "C" int orbm_search_for_triangulation_frames(orbm_t* h, orbm_frame_t* f1, const uint8_t* skip1, orbm_frame_t* f2, const uint8_t* skip2,
                                                    const float F12[9], float ex, float ey, const float* sf2, const float* sigma2_2, int nlevels,
                                                    int check_ori, int32_t* matches12, int* nmatches)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (!f1 || !f2 || f1->owner != h || f2->owner != h) return fail(ORBX_E_INVALID, "frames do not belong to this matcher handle");
    if (!f1->hasBow || !f2->hasBow) return fail(ORBX_E_INVALID, "orbm_frame_compute_bow has not run on both frames");
    if (!F12 || !sf2 || !sigma2_2 || nlevels < 1 || nlevels > 16 || (f1->n && !matches12)) return fail(ORBX_E_INVALID, "bad argument");
    for (int i = 0; i < f1->n; i++) matches12[i] = -1;
    if (nmatches) *nmatches = 0;
    if (f1->n == 0 || f2->n == 0 || f1->fvNodes == 0 || f2->fvNodes == 0) return ORBX_OK;
    const TriSide A = {f1->d_keysUn, f1->d_desc, f1->n, f1->d_fvStart, f1->d_fvIdx, f1->fvNode.data(), f1->fvNodes, skip1, nullptr};
    const TriSide B = {f2->d_keysUn, f2->d_desc, f2->n, f2->d_fvStart, f2->d_fvIdx, f2->fvNode.data(), f2->fvNodes, skip2, nullptr};
    return tri_core(h, A, B, F12, ex, ey, sf2, sigma2_2, nlevels, 0, check_ori, matches12, nmatches);
}
#undef UP

extern "C" int orbm_undistort_keypoints(orbm_t* h, const OrbxKeyPoint* keys, int n, const float K[4], const float D[5],
                                        OrbxKeyPoint* keys_un)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (n < 0 || !K || !D || (n && (!keys || !keys_un))) return fail(ORBX_E_INVALID, "bad argument");
    if (n == 0) return ORBX_OK;
    if (D[0] == 0.0f) { memcpy(keys_un, keys, (size_t)n * sizeof(OrbxKeyPoint)); return ORBX_OK; }  // mvKeysUn = mvKeys (Frame.cc:406-410)
    if ((rc = orbm_reserve(h, 0, (size_t)n * sizeof(OrbxKeyPoint))) || (rc = orbm_reserve(h, 1, (size_t)n * sizeof(OrbxKeyPoint)))) return rc;
    hipStream_t s = h->stream;
    HIPCHK(hipMemcpyAsync(h->d_buf[0], keys, (size_t)n * sizeof(OrbxKeyPoint), hipMemcpyHostToDevice, s));
    orbm::UndistArgs a = {K[0], K[1], K[2], K[3], D[0], D[1], D[2], D[3], D[4]};
    hipLaunchKernelGGL(orbm::k_undistort, dim3((n + 255) / 256), dim3(256), 0, s, (const orbm::KeyDev*)h->d_buf[0], n, a,
                       (orbm::KeyDev*)h->d_buf[1]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(keys_un, h->d_buf[1], (size_t)n * sizeof(OrbxKeyPoint), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ORBX_OK;
}

extern "C" int orbm_distinctive_descriptors(orbm_t* h, const uint8_t* desc, const int32_t* start, int npoints, int32_t* best_idx)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (npoints < 0 || !start || (npoints && !best_idx)) return fail(ORBX_E_INVALID, "bad argument");
    if (npoints == 0) return ORBX_OK;
    const int total = start[npoints];
    int maxN = 0;
    for (int p = 0; p < npoints; p++) {
        if (start[p + 1] < start[p]) return fail(ORBX_E_INVALID, "start must be non-decreasing");
        maxN = std::max(maxN, start[p + 1] - start[p]);
    }
    if (total && !desc) return fail(ORBX_E_INVALID, "null descriptors");
    if ((size_t)maxN * 4 > 150 * 1024) return fail(ORBX_E_UNSUPPORTED, "more than 38400 observations of one map point");
    if ((rc = orbm_reserve(h, 0, (size_t)std::max(total, 1) * 32)) || (rc = orbm_reserve(h, 1, (size_t)(npoints + 1) * 4)) ||
        (rc = orbm_reserve(h, 2, (size_t)npoints * 4))) return rc;
    hipStream_t s = h->stream;
    if (total) HIPCHK(hipMemcpyAsync(h->d_buf[0], desc, (size_t)total * 32, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(h->d_buf[1], start, (size_t)(npoints + 1) * 4, hipMemcpyHostToDevice, s));
    const size_t lds = (size_t)std::max(maxN, 1) * 4;
    if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)orbm::k_distinctive, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(orbm::k_distinctive, dim3(npoints), dim3(64), lds, s, (const uint8_t*)h->d_buf[0], (const int32_t*)h->d_buf[1],
                       (int32_t*)h->d_buf[2]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(best_idx, h->d_buf[2], (size_t)npoints * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ORBX_OK;
}

// ------------------------------------------------------------------ vocabulary
struct orbv_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    int k = 0, L = 0, scoring = 0, weighting = 0, nNodes = 0, nWords = 0;
    std::vector<int> nodesAtLevel;   // [0 .. deepest]: how many nodes the tree has at each depth (root = 0)
    int32_t* d_childStart = nullptr; int32_t* d_childIdx = nullptr; uint8_t* d_desc = nullptr;
    int32_t* d_wordId = nullptr; double* d_weight = nullptr;
    void* d_buf[10] = {nullptr}; size_t d_cap[10] = {0};
};

static int orbv_reserve(orbv_handle* h, int slot, size_t bytes)
{
    if (bytes <= h->d_cap[slot]) return ORBX_OK;
    if (h->d_buf[slot]) HIPCHK(hipFree(h->d_buf[slot]));
    h->d_buf[slot] = nullptr; h->d_cap[slot] = 0;
    const size_t want = std::max<size_t>(bytes * 3 / 2, 4096);
    HIPCHK(hipMalloc(&h->d_buf[slot], want));
    h->d_cap[slot] = want;
    return ORBX_OK;
}

extern "C" void orbv_destroy(orbv_t* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
    void* ptrs[] = {h->d_childStart, h->d_childIdx, h->d_desc, h->d_wordId, h->d_weight};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto p : h->d_buf) if (p) (void)hipFree(p);
    delete h;
}

extern "C" int orbv_create(int device, int k, int L, int scoring, int weighting, int n,
                           const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc,
                           const double* weight, orbv_t** out)
{
    if (!out) return fail(ORBX_E_INVALID, "null argument");
    *out = nullptr;
    // same sanity window as loadFromTextFile (:1360)
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3 || n < 0 ||
        (n && (!parent || !is_leaf || !desc || !weight)))
        return fail(ORBX_E_INVALID, "not a correct vocabulary");
    int ndev = orbx_device_count();
    if (ndev == 0) return fail(ORBX_E_NO_DEVICE, "no HIP device visible: the vocabulary transform has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(ORBX_E_INVALID, "device %d out of range", device);
    const int nNodes = n + 1;
    std::vector<int32_t> cnt(nNodes + 1, 0), childStart(nNodes + 1, 0), childIdx(std::max(n, 1)), wordId(nNodes, -1);
    std::vector<double> w(nNodes, 0.0);
    std::vector<uint8_t> d((size_t)nNodes * 32, 0);
    for (int i = 0; i < n; i++) {
        if (parent[i] < 0 || parent[i] > i) return fail(ORBX_E_INVALID, "node %d: parent %d does not precede it", i + 1, parent[i]);
        cnt[parent[i]]++;
    }
    for (int i = 0; i < nNodes; i++) childStart[i + 1] = childStart[i] + cnt[i];
    std::fill(cnt.begin(), cnt.end(), 0);
    int nWords = 0;
    for (int i = 0; i < n; i++) {
        const int nid = i + 1, pid = parent[i];
        childIdx[childStart[pid] + cnt[pid]++] = nid;     // m_nodes[pid].children.push_back(nid)
        memcpy(&d[(size_t)nid * 32], desc + (size_t)i * 32, 32);
        w[nid] = weight[i];
        if (is_leaf[i]) wordId[nid] = nWords++;
    }
    // every inner node must have children, every childless node must be a word (else the descent of :1236-1253 derails)
    for (int i = 1; i < nNodes; i++)
        if ((childStart[i + 1] == childStart[i]) != (wordId[i] >= 0)) return fail(ORBX_E_INVALID, "node %d: leaf flag and children disagree", i);
    orbv_handle* h = new orbv_handle();
    h->device = device; h->k = k; h->L = L; h->scoring = scoring; h->weighting = weighting; h->nNodes = nNodes; h->nWords = nWords;
    {
        std::vector<int> depth(nNodes, 0);
        h->nodesAtLevel.assign(1, 1);
        for (int i = 0; i < n; i++) {
            const int dpt = depth[i + 1] = depth[parent[i]] + 1;
            if ((int)h->nodesAtLevel.size() <= dpt) h->nodesAtLevel.resize(dpt + 1, 0);
            h->nodesAtLevel[dpt]++;
        }
    }
#define VCRT(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { int r_ = fail(ORBX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); orbv_destroy(h); return r_; } } while (0)
    VCRT(hipSetDevice(device));
    VCRT(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    VCRT(hipMalloc(&h->d_childStart, (size_t)(nNodes + 1) * 4));
    VCRT(hipMalloc(&h->d_childIdx, (size_t)std::max(n, 1) * 4));
    VCRT(hipMalloc(&h->d_desc, (size_t)nNodes * 32));
    VCRT(hipMalloc(&h->d_wordId, (size_t)nNodes * 4));
    VCRT(hipMalloc(&h->d_weight, (size_t)nNodes * 8));
    VCRT(hipMemcpy(h->d_childStart, childStart.data(), (size_t)(nNodes + 1) * 4, hipMemcpyHostToDevice));
    VCRT(hipMemcpy(h->d_childIdx, childIdx.data(), (size_t)std::max(n, 1) * 4, hipMemcpyHostToDevice));
    VCRT(hipMemcpy(h->d_desc, d.data(), (size_t)nNodes * 32, hipMemcpyHostToDevice));
    VCRT(hipMemcpy(h->d_wordId, wordId.data(), (size_t)nNodes * 4, hipMemcpyHostToDevice));
    VCRT(hipMemcpy(h->d_weight, w.data(), (size_t)nNodes * 8, hipMemcpyHostToDevice));
#undef VCRT
    *out = h;
    return ORBX_OK;
}

extern "C" int orbv_load_text(int device, const char* path, orbv_t** out)
{
    if (!path || !out) return fail(ORBX_E_INVALID, "null argument");
    std::ifstream f(path);
    if (!f.is_open()) return fail(ORBX_E_INVALID, "cannot open %s", path);
    std::string s;
    std::getline(f, s);
    std::stringstream ss(s);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    ss >> k >> L >> n1 >> n2;
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3)
        return fail(ORBX_E_INVALID, "Vocabulary loading failure: This is not a correct text file!");
    std::vector<int32_t> parent; std::vector<uint8_t> leaf, desc; std::vector<double> weight;
    while (std::getline(f, s)) {
        if (s.find_first_not_of(" \t\r\n") == std::string::npos) continue;  // the reference trips over a trailing newline
        std::stringstream sn(s);
        int pid = -1, isLeaf = 0;
        sn >> pid >> isLeaf;
        parent.push_back(pid); leaf.push_back(isLeaf > 0);
        for (int i = 0; i < 32; i++) { int b = 0; sn >> b; desc.push_back((uint8_t)b); }  // FORB::fromString
        double w = 0; sn >> w; weight.push_back(w);
        if (sn.fail()) return fail(ORBX_E_INVALID, "malformed vocabulary line %zu", parent.size() + 1);
    }
    return orbv_create(device, k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), out);
}

// transform of n device-resident descriptors; results stay in the handle's scratch (slots below), counts = {words, fv nodes}
enum { SV_DESC, SV_WORD, SV_NODE, SV_W, SV_OW, SV_OV, SV_FN, SV_FS, SV_FI, SV_CNT };
static int voc_transform_device(orbv_handle* h, const uint8_t* d_desc, int n, int levelsup, int32_t counts[2])
{
    counts[0] = counts[1] = 0;
    if (n > 8192) return fail(ORBX_E_UNSUPPORTED, "more than 8192 descriptors per transform");
    int P = 2;
    while (P < n) P <<= 1;
    const size_t sizes[] = {(size_t)n * 32, (size_t)n * 4, (size_t)n * 4, (size_t)n * 8, (size_t)n * 4, (size_t)n * 8,
                            (size_t)n * 4, (size_t)(n + 1) * 4, (size_t)n * 4, 16};
    int rc;
    for (int i = 0; i < 10; i++) if ((rc = orbv_reserve(h, i, sizes[i]))) return rc;
    hipStream_t s = h->stream;
    if (!d_desc) d_desc = (const uint8_t*)h->d_buf[SV_DESC];
    orbv::VocDev v{h->d_childStart, h->d_childIdx, h->d_desc, h->d_wordId, h->d_weight, h->L, h->scoring, h->weighting};
    hipLaunchKernelGGL(orbv::k_voc_descend, dim3((n + 63) / 64), dim3(64), 0, s, v, d_desc, n, levelsup,
                       (uint32_t*)h->d_buf[SV_WORD], (uint32_t*)h->d_buf[SV_NODE], (double*)h->d_buf[SV_W]);
    const bool accLds = P <= 4096;
    const size_t lds = (size_t)P * (accLds ? 36 : 12);
    auto kern = accLds ? orbv::k_voc_aggregate<true> : orbv::k_voc_aggregate<false>;
    if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(1), dim3(orbv::kAggThreads), lds, s, v, n, P, (const uint32_t*)h->d_buf[SV_WORD],
                       (const uint32_t*)h->d_buf[SV_NODE], (const double*)h->d_buf[SV_W], (uint32_t*)h->d_buf[SV_OW],
                       (double*)h->d_buf[SV_OV], (uint32_t*)h->d_buf[SV_FN], (int32_t*)h->d_buf[SV_FS], (int32_t*)h->d_buf[SV_FI],
                       (int32_t*)h->d_buf[SV_CNT]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(counts, h->d_buf[SV_CNT], 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ORBX_OK;
}

extern "C" int orbv_transform(orbv_t* h, const uint8_t* desc, int n, int levelsup,
                              uint32_t* word_id, double* word_value, int* n_words,
                              uint32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int* n_fv_nodes)
{
    if (!h) return fail(ORBX_E_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (n < 0 || (n && (!desc || !word_id || !word_value || !fv_node || !fv_idx)) || !fv_start || !n_words || !n_fv_nodes)
        return fail(ORBX_E_INVALID, "bad argument");
    *n_words = 0; *n_fv_nodes = 0; fv_start[0] = 0;
    if (h->nWords == 0 || n == 0) return ORBX_OK;  // empty(): v and fv stay cleared (:1133-1136)
    if (n > 8192) return fail(ORBX_E_UNSUPPORTED, "more than 8192 descriptors per transform");
    int rc;
    if ((rc = orbv_reserve(h, SV_DESC, (size_t)n * 32))) return rc;
    HIPCHK(hipMemcpyAsync(h->d_buf[SV_DESC], desc, (size_t)n * 32, hipMemcpyHostToDevice, h->stream));
    int32_t counts[2];
    if ((rc = voc_transform_device(h, nullptr, n, levelsup, counts))) return rc;
    *n_words = counts[0]; *n_fv_nodes = counts[1];
    if (counts[0]) {
        HIPCHK(hipMemcpy(word_id, h->d_buf[SV_OW], (size_t)counts[0] * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(word_value, h->d_buf[SV_OV], (size_t)counts[0] * 8, hipMemcpyDeviceToHost));
    }
    HIPCHK(hipMemcpy(fv_start, h->d_buf[SV_FS], (size_t)(counts[1] + 1) * 4, hipMemcpyDeviceToHost));
    if (counts[1]) {
        HIPCHK(hipMemcpy(fv_node, h->d_buf[SV_FN], (size_t)counts[1] * 4, hipMemcpyDeviceToHost));
        const int m = fv_start[counts[1]];
        if (m) HIPCHK(hipMemcpy(fv_idx, h->d_buf[SV_FI], (size_t)m * 4, hipMemcpyDeviceToHost));
    }
    return ORBX_OK;
}

// ------------------------------------------------------------------ SURVEY 8(f).2+3: BoW on a device-resident Frame
// Frame::ComputeBoW (src/Frame.cc:394-402) on the frame's descriptors in HBM: the BowVector goes to the host (the
// KeyFrame database and the relocaliser read it), the FeatureVector stays with the frame for SearchByBoW.
extern "C" int orbm_frame_compute_bow(orbm_frame_t* f, orbv_t* voc, int levelsup,
                                      uint32_t* word_id, double* word_value, int* n_words)
{
    if (!f || !f->owner || !voc) return fail(ORBX_E_INVALID, "null argument");
    int rc = orbm_check(f->owner);
    if (rc) return rc;
    if (voc->device != f->owner->device) return fail(ORBX_E_INVALID, "vocabulary and frame live on different devices");
    if (!n_words || (f->n && (!word_id || !word_value))) return fail(ORBX_E_INVALID, "bad argument");
    *n_words = 0;
    f->fvNode.clear(); f->fvNodes = 0; f->hasBow = true;
    if (voc->nWords == 0 || f->n == 0) return ORBX_OK;
    int32_t counts[2];
    if ((rc = voc_transform_device(voc, f->d_desc, f->n, levelsup, counts))) return rc;
    *n_words = counts[0];
    if (counts[0]) {
        HIPCHK(hipMemcpy(word_id, voc->d_buf[SV_OW], (size_t)counts[0] * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(word_value, voc->d_buf[SV_OV], (size_t)counts[0] * 8, hipMemcpyDeviceToHost));
    }
    f->fvNodes = counts[1];
    f->fvNode.resize((size_t)counts[1]);
    if (!f->d_fvStart) {
        HIPCHK(hipMalloc(&f->d_fvStart, (size_t)(f->n + 1) * 4));
        HIPCHK(hipMalloc(&f->d_fvIdx, (size_t)f->n * 4));
    }
    HIPCHK(hipMemcpy(f->d_fvStart, voc->d_buf[SV_FS], (size_t)(counts[1] + 1) * 4, hipMemcpyDeviceToDevice));
    if (counts[1]) {
        HIPCHK(hipMemcpy(f->fvNode.data(), voc->d_buf[SV_FN], (size_t)counts[1] * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(f->d_fvIdx, voc->d_buf[SV_FI], (size_t)f->n * 4, hipMemcpyDeviceToDevice));
    }
    return ORBX_OK;
}

/* SearchByBoW between two device-resident frames (query = the KeyFrame / pKF1, train = the Frame / pKF2) */
extern "C" int orbm_search_by_bow_frames(orbm_t* h, orbm_frame_t* q, const uint8_t* qvalid, orbm_frame_t* t, const uint8_t* tvalid,
                                         float nnratio, int check_ori, int out_by_train, int32_t* match, int* nmatches)
{
    int rc = orbm_check(h);
    if (rc) return rc;
    if (!q || !t || q->owner != h || t->owner != h) return fail(ORBX_E_INVALID, "frames do not belong to this matcher handle");
    if (!q->hasBow || !t->hasBow) return fail(ORBX_E_INVALID, "orbm_frame_compute_bow has not run on both frames");
    if (!match) return fail(ORBX_E_INVALID, "bad argument");
    const int nout = out_by_train ? t->n : q->n;
    for (int i = 0; i < nout; i++) match[i] = -1;
    if (nmatches) *nmatches = 0;
    if (q->n == 0 || t->n == 0 || q->fvNodes == 0 || t->fvNodes == 0) return ORBX_OK;
    const BowSide qs = {q->d_desc, q->d_ang, q->d_fvStart, q->d_fvIdx, q->fvNode.data(), q->fvNodes, q->n};
    const BowSide ts = {t->d_desc, t->d_ang, t->d_fvStart, t->d_fvIdx, t->fvNode.data(), t->fvNodes, t->n};
    return bow_core(h, qs, qvalid, ts, tvalid, nnratio, check_ori, out_by_train, match, nmatches);
}

#include "orbt_bow_host.inc"
