// multi_robot.cpp -- the reference's multi-robot main loop on the C ABI of liborbslamm_hip.so.
//
// Reference: /root/reference/MultipleRobotsScenario/Examples/Monocular/mono_kitti.cc:83-125 -- one System / tracking
// thread per robot, each feeding ITS camera stream one frame per iteration (:88-101), per-frame times collected and
// printed as median / mean at the end (SingleRobotScenario/Examples/Monocular/mono_tum.cc:113-122).  Here: one host
// thread per robot, robot r on GPU r % N, one extractor handle (ORBextractor) + one matcher handle and frame set
// (what Frame::Frame's tail and ORBmatcher::SearchByProjection need) per robot, nothing shared between robots.
// Per frame a robot does what Tracking::GrabImageMonocular does on the hot path (SingleRobotScenario/src/Tracking.cc:240-267):
//   Frame::Frame        ExtractORB -> UndistortKeyPoints -> AssignFeaturesToGrid   (src/Frame.cc:175-210)
//   TrackWithMotionModel SearchByProjection(CurrentFrame, LastFrame, th = 15, mono)  (src/Tracking.cc:925-936)
// with keypoints + descriptors and the match table back on the host (--mode track), or extract + brute-force match
// against the previous frame (--mode bf, BASELINE.json's headline pair), or extraction alone (--mode extract), or the whole
// front-end of a Tracking iteration (--mode full: track + SearchByProjection(Frame, 3000 local MapPoints), Tracking.cc:1242-1249).
// The GPUs exchange nothing on the data path; once per reporting interval the robots' counters (a 64-byte record per
// GPU) are gathered with ONE ncclAllGather over RCCL -- the only collective there is.
//
// Synthetic camera: a static scene of rectangles and discs on a canvas, a slowly panning viewport, additive noise
// (the shape SURVEY.md 8d describes; SplitMix64, no Python).
//
// build:  hipcc -O2 -std=c++17 examples/multi_robot.cpp -Iinclude -Lorbslamm_amd -lorbslamm_hip -lrccl -Wl,-rpath,'$ORIGIN/../orbslamm_amd' -o examples/multi_robot
// usage:  multi_robot [--gpus N] [--robots R] [--frames F] [--warmup W] [--mode track|bf|extract|full] [--depth 1|2]
//                     [--hub P | -1 = four hubs per GPU (robots in groups of P share one orbslamm::CameraHub: still one thread per robot, the frames that wait together go through one chain) --hub-wait US]
//                     [--mode batch --batch 64 --pool 8 --steps 200 (the offline-sequence mode: bench.py's step through the device-resident entries, one thread + one handle per GPU)]
//                     [--per-call 1..8 (cameras whose frames one thread puts through the chain together)] [--attach 0|1] [--pinned 0|1] [--w 1241 --h 376 --nfeat 2000] [--interval 200] [--json] [--dump FILE]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "orbslamm_hip.h"
#include "orbslamm_hub.hpp"

namespace {

struct Args {
    int gpus = 1, robots = 1, per_call = 1, hub = 0, hub_wait = 40, frames = 600, warmup = 30, depth = 1, attach = 1, pinned = 1, w = 1241, h = 376, nfeat = 2000, interval = 200;
    int batch = 64, pool = 8, steps = 200;   // --mode batch: frames per step, batches resident in HBM, timed steps
    std::string mode = "track", dump;
    bool json = false;
};

// ---------------------------------------------------------------- synthetic camera
struct SplitMix { uint64_t s; uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
                  int below(int n) { return (int)(next() % (uint64_t)n); } };

static int tri(int a, int m) { a %= 2 * m; return a <= m ? a : 2 * m - a; }

static std::vector<uint8_t> make_scene(int w, int h, int robot, int& cw, int& ch)
{
    cw = w + 64; ch = h + 16;
    std::vector<uint8_t> c((size_t)cw * ch, 128);
    SplitMix r{0x0B5A4000ull + (uint64_t)robot};
    const int nshapes = std::max(200, (int)(4000.0 * w * h / (1241.0 * 376.0)));
    for (int i = 0; i < nshapes; i++) {
        const int x0 = r.below(cw), y0 = r.below(ch), sx = 4 + r.below(37), sy = 4 + r.below(37), val = r.below(256), kind = r.below(2);
        if (kind == 0) {
            for (int y = std::max(0, y0 - sy / 2); y <= std::min(ch - 1, y0 + sy / 2); y++)
                for (int x = std::max(0, x0 - sx / 2); x <= std::min(cw - 1, x0 + sx / 2); x++) c[(size_t)y * cw + x] = (uint8_t)val;
        } else {
            const int rad = sx / 2;
            for (int y = std::max(0, y0 - rad); y <= std::min(ch - 1, y0 + rad); y++)
                for (int x = std::max(0, x0 - rad); x <= std::min(cw - 1, x0 + rad); x++)
                    if ((y - y0) * (y - y0) + (x - x0) * (x - x0) <= rad * rad) c[(size_t)y * cw + x] = (uint8_t)val;
        }
    }
    return c;
}

static void make_frame(const std::vector<uint8_t>& scene, int cw, int w, int h, int robot, int t, uint8_t* dst, int stride)
{
    const int ox = tri(2 * t, 64), oy = tri(t, 16);
    SplitMix r{(0x5EED0000ull + (uint64_t)robot) * 100003ull + (uint64_t)t};
    for (int y = 0; y < h; y++) {
        const uint8_t* s = &scene[(size_t)(oy + y) * cw + ox];
        uint8_t* d = dst + (size_t)y * stride;
        for (int x = 0; x < w; x += 8) {
            uint64_t bits = r.next();
            for (int k = 0; k < 8 && x + k < w; k++, bits >>= 8) {
                const int v = (int)s[x + k] + (int)((bits & 0xFF) % 9) - 4;
                d[x + k] = (uint8_t)std::min(255, std::max(0, v));
            }
        }
    }
}

// ---------------------------------------------------------------- per-robot state and loop
struct Stats {   // the record the GPUs gather (64 bytes)
    int64_t frames, keypoints, matches, ns_busy;
    uint64_t checksum;
    int64_t pad[3];
};
static_assert(sizeof(Stats) == 64, "stats record");

#define OX(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "robot %d: %s failed (%d): %s\n", robot, #call, rc_, orbx_last_error()); failed = true; return; } } while (0)

struct Robot {
    int robot = 0, device = 0;
    const Args* a = nullptr;
    std::vector<double> lat_ms;
    Stats st{};
    bool failed = false;
    std::atomic<int64_t> live_frames{0}, live_kps{0}, live_matches{0};
    orbslamm::CameraHub* hub = nullptr; int cam = 0;   // --hub: this robot is camera `cam` of a hub shared with its neighbours
    double batch_mean = 0;
    double us_submit = 0, us_enqueue = 0, us_wait = 0;   // host time in orbx_submit_batch | build + track calls | collect + results (waiting included)

    // --hub: the robot's loop is what it was -- one blocking call per frame -- but the call goes to a hub that puts the frames
    // of the robots waiting at that moment through ONE chain (include/orbslamm_hub.hpp)
    void run_hub(std::atomic<int>& ready, std::atomic<bool>& go)
    {
        const Args& A = *a;
        const int nring = 8, cap = hub->cap();
        uint8_t* ring = nullptr; int stride = A.w; size_t pitch = (size_t)A.w * A.h;
        std::vector<uint8_t> pageable;
        if (A.pinned) OX(orbx_host_alloc_frames(hub->extractor(), nring, A.w, A.h, &ring, &stride, &pitch));
        else { pageable.resize(pitch * nring); ring = pageable.data(); }
        {
            int cw, ch;
            const std::vector<uint8_t> scene = make_scene(A.w, A.h, robot, cw, ch);
            for (int t = 0; t < nring; t++) make_frame(scene, cw, A.w, A.h, robot, t, ring + (size_t)t * pitch, stride);
        }
        std::vector<OrbxKeyPoint> kps((size_t)cap);
        std::vector<uint8_t> desc((size_t)cap * 32);
        std::vector<int32_t> assign((size_t)cap), assign3((size_t)cap);
        const int total = A.warmup + A.frames;
        lat_ms.reserve(A.frames);
        // --mode full: the local map of ring position r = the keypoints of the two frames before it as projected MapPoints
        // (the same construction as in run() below), built once per ring position during the warm-up
        const bool full = A.mode == "full";
        struct LocalMap { std::vector<float> uvr; std::vector<int8_t> lvl; std::vector<uint8_t> desc; int n = 0; };
        std::vector<LocalMap> lmap(full ? nring : 0);
        std::vector<std::vector<OrbxKeyPoint>> rkps(full ? nring : 0);
        std::vector<std::vector<uint8_t>> rdesc(full ? nring : 0);
        float sfl[ORBX_MAX_LEVELS] = {0};
        if (full) OX(orbx_scale_tables(hub->extractor(), sfl, nullptr, nullptr, nullptr));
        OrbmProjParams pp3{3, 0.8f, 0, 100};
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        using clk = std::chrono::steady_clock;
        clk::time_point tTimed = clk::now();
        int64_t batchSum = 0;
        for (int i = 0; i < total; i++) {
            if (i == A.warmup) tTimed = clk::now();
            const uint8_t* fr0 = ring + (size_t)(i % nring) * pitch;
            const auto t0 = clk::now();
            orbslamm::CameraHub::Result res;
            OX(hub->track(cam, fr0, stride, kps.data(), desc.data(), assign.data(), &res));
            const int n = res.n;
            int nm3 = 0;
            if (full) {
                const int r = i % nring;
                if (rkps[r].empty() && n > 0) {   // (warm-up, first pass over the ring)
                    rkps[r].assign(kps.begin(), kps.begin() + n);
                    rdesc[r].assign(desc.begin(), desc.begin() + (size_t)n * 32);
                    for (int rr = 0; rr < nring; rr++) {
                        const int p1 = (rr + nring - 1) % nring, p2 = (rr + nring - 2) % nring;
                        if (lmap[rr].n || rkps[p1].empty() || rkps[p2].empty()) continue;
                        LocalMap& lm = lmap[rr];
                        for (int src = 0; src < 2 && lm.n < 3000; src++) {
                            const auto& K = rkps[src ? p2 : p1]; const auto& D = rdesc[src ? p2 : p1];
                            for (size_t k = 0; k < K.size() && lm.n < 3000; k++, lm.n++) {
                                const int lv = K[k].octave;
                                lm.uvr.push_back(K[k].x); lm.uvr.push_back(K[k].y); lm.uvr.push_back(4.0f * sfl[lv]);
                                lm.lvl.push_back((int8_t)(lv - 1)); lm.lvl.push_back((int8_t)lv);
                                lm.desc.insert(lm.desc.end(), D.begin() + k * 32, D.begin() + (k + 1) * 32);
                            }
                        }
                    }
                }
                LocalMap& lm = lmap[r];
                if (lm.n > 0) {   // Tracking::SearchLocalPoints (Tracking.cc:1242-1249) against the frame just sent, resident in the hub's set
                    OX(hub->search_local_points(cam, &pp3, lm.uvr.data(), lm.lvl.data(), lm.desc.data(), nullptr, nullptr, lm.n, nullptr, assign3.data(), &nm3));
                }
            }
            const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            uint64_t cs = (uint64_t)n * 1315423911ull;   // the same per-frame word the loop below forms for one camera per call
            if (n > 0) { uint32_t d0; memcpy(&d0, desc.data() + (size_t)(n - 1) * 32, 4); cs ^= d0; }
            if (res.nmatches >= 0 && n > 0) cs ^= (uint64_t)(uint32_t)assign[n - 1] << 32;
            if (full && lmap[i % nring].n > 0) cs ^= (uint64_t)(uint32_t)nm3 * 2654435761ull;
            if (robot == 0 && !A.dump.empty() && i >= A.warmup && i < A.warmup + 8) {
                FILE* df = fopen(A.dump.c_str(), i == A.warmup ? "wb" : "ab");
                if (df) {
                    const int32_t hdr[6] = {i, n, A.w, A.h, 2, 1};
                    fwrite(hdr, sizeof hdr, 1, df);
                    for (int y = 0; y < A.h; y++) fwrite(fr0 + (size_t)y * stride, 1, (size_t)A.w, df);
                    fwrite(kps.data(), sizeof(OrbxKeyPoint), (size_t)n, df);
                    fwrite(desc.data(), 32, (size_t)n, df);
                    if (i > 0) { const int32_t nm = res.nmatches; fwrite(assign.data(), 4, (size_t)n, df); fwrite(&nm, 4, 1, df); }
                    fclose(df);
                }
            }
            if (i >= A.warmup) {
                lat_ms.push_back(ms);
                const int nm = std::max(0, res.nmatches) + nm3;
                st.frames += 1; st.keypoints += n; st.matches += nm; st.checksum = st.checksum * 1099511628211ull ^ cs;
                batchSum += res.batch;
                live_frames.fetch_add(1, std::memory_order_relaxed); live_kps.fetch_add(n, std::memory_order_relaxed); live_matches.fetch_add(nm, std::memory_order_relaxed);
            }
        }
        hub->leave(cam);
        st.frames = (int64_t)lat_ms.size();
        batch_mean = (double)batchSum / std::max<size_t>(1, lat_ms.size());
        st.ns_busy = std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - tTimed).count();
        if (A.pinned) orbx_host_free(hub->extractor(), ring);
    }

    // --mode batch: an offline sequence of this robot's camera through the batched device-resident entries -- what bench.py
    // times as `value` (BASELINE.json configs[3]: 64 frames in flight): --pool batches of --batch frames resident in HBM, per
    // step orbx_extract_batch_device + orbx_match_prev_batch_device, nothing crosses the link inside the timed region.
    // One thread + one handle per GPU; lat_ms holds the per-step times.
    void run_batch(std::atomic<int>& ready, std::atomic<bool>& go)
    {
        const Args& A = *a;
        OrbxParams prm{A.nfeat, 1.2f, 8, 20, 7};
        orbx_t* ex = nullptr;
        const int B = A.batch;
        OX(orbx_create(&prm, A.w, A.h, B, device, &ex));
        const int cap = orbx_max_keypoints(ex);
        const int stride = (A.w + 63) / 64 * 64;
        const size_t pitch = (size_t)stride * A.h;
        std::vector<void*> d_pool((size_t)A.pool, nullptr);
        {
            int cw, ch;
            const std::vector<uint8_t> scene = make_scene(A.w, A.h, robot, cw, ch);
            std::vector<uint8_t> host(pitch * B);
            for (int p = 0; p < A.pool; p++) {
                for (int f = 0; f < B; f++) make_frame(scene, cw, A.w, A.h, robot, p * B + f, host.data() + (size_t)f * pitch, stride);
                OX(orbx_device_alloc(ex, pitch * B, &d_pool[p]));
                OX(orbx_upload(ex, d_pool[p], host.data(), pitch * B));
            }
        }
        auto step = [&](int i) -> int {
            int rc = orbx_extract_batch_device(ex, (const uint8_t*)d_pool[i % A.pool], B, A.w, A.h, stride, pitch);
            return rc ? rc : orbx_match_prev_batch_device(ex, 0.7f, 50, 1);
        };
        // every resident batch once (device error flag read, both result sets in use), then the warm-up, then the timed steps
        for (int i = 0; i < A.pool; i++) OX(step(i));
        OX(orbx_sync(ex));
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        for (int i = 0; i < A.warmup; i++) OX(step(i));
        OX(orbx_sync(ex));
        const auto t0 = std::chrono::steady_clock::now();
        auto tp = t0;
        for (int i = 0; i < A.steps; i++) {
            OX(step(A.warmup + i));
            live_frames.fetch_add(B);
            const auto tn = std::chrono::steady_clock::now();   // (enqueue pace: the device runs behind; the sum is exact after the sync)
            lat_ms.push_back(std::chrono::duration<double, std::milli>(tn - tp).count());
            tp = tn;
        }
        OX(orbx_sync(ex));
        st.ns_busy = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        st.frames = (int64_t)B * A.steps;
        // the last batch's results: counts of every frame, checksum over its last frame's bytes
        std::vector<OrbxKeyPoint> kps((size_t)cap);
        std::vector<uint8_t> desc((size_t)cap * 32);
        std::vector<int32_t> match((size_t)cap);
        uint64_t cs = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; i++) { cs ^= b[i]; cs *= 1099511628211ull; } };
        int64_t kp_sum = 0, m_sum = 0;
        for (int f = 0; f < B; f++) {
            int n = 0, nm = 0;
            OX(orbx_download(ex, f, kps.data(), desc.data(), cap, &n));
            OX(orbx_download_matches(ex, f, match.data(), cap, &nm));
            kp_sum += n; m_sum += nm;
            if (f == B - 1) { mix(kps.data(), (size_t)n * sizeof(OrbxKeyPoint)); mix(desc.data(), (size_t)n * 32); mix(match.data(), (size_t)n * 4); }
        }
        st.keypoints = kp_sum * A.steps; st.matches = m_sum * A.steps;   // (per-frame means of the last batch, scaled to the run)
        live_kps.store(st.keypoints); live_matches.store(st.matches);
        st.checksum = cs;
        for (void* d : d_pool) if (d) orbx_device_free(ex, d);
        orbx_destroy(ex);
    }

    void run(std::atomic<int>& ready, std::atomic<bool>& go)
    {
        const Args& A = *a;
        const bool full = A.mode == "full";   // track + Tracking::SearchLocalPoints' search: the whole front-end of one Tracking iteration
        const bool track = A.mode == "track" || full, bf = A.mode == "bf";
        OrbxParams prm{A.nfeat, 1.2f, 8, 20, 7};
        orbx_t* ex = nullptr;
        const int P = A.per_call;   // cameras this thread feeds per call (their frames go through the chain together)
        OX(orbx_create(&prm, A.w, A.h, P, device, &ex));
        const int cap = orbx_max_keypoints(ex);
        orbm_t* m = nullptr;
        orbm_frameset_t* fs = nullptr;
        if (track) {
            OX(orbm_create(device, &m));
            float sf[ORBX_MAX_LEVELS];
            OX(orbx_scale_tables(ex, sf, nullptr, nullptr, nullptr));
            const float K[4] = {718.856f, 718.856f, 607.1928f, 185.2157f}, D[5] = {0, 0, 0, 0, 0};
            const float bounds[4] = {0.f, (float)A.w, 0.f, (float)A.h};
            OrbmGrid g{0.f, 0.f, 64.f / (float)A.w, 48.f / (float)A.h, 64, 48};
            OX(orbm_frameset_create(m, 4 * P, cap, K, D, &g, bounds, sf, orbx_levels(ex), &fs));
            if (A.attach) OX(orbm_frameset_attach(fs, ex));
        }
        // the camera's ring buffer: 8 frames, pinned in the device layout (what a capture driver would fill) or pageable
        // (camera j of this thread: frames j * nring .. of the buffer, scene of robot `robot * P + j`)
        const int nring = 8;
        uint8_t* ring = nullptr; int stride = A.w; size_t pitch = (size_t)A.w * A.h;
        std::vector<uint8_t> pageable;
        if (A.pinned) OX(orbx_host_alloc_frames(ex, nring * P, A.w, A.h, &ring, &stride, &pitch));
        else { pageable.resize(pitch * nring * P); ring = pageable.data(); }
        for (int j = 0; j < P; j++) {
            int cw, ch;
            const std::vector<uint8_t> scene = make_scene(A.w, A.h, robot * P + j, cw, ch);
            for (int t = 0; t < nring; t++) make_frame(scene, cw, A.w, A.h, robot * P + j, t, ring + (size_t)(j * nring + t) * pitch, stride);
        }
        OrbxStreamOpts so{bf ? 1 : 0, 0.7f, 50, 1};
        OrbmProjParams pp{4, 0.9f, 1, 100};
        const int total = A.warmup + A.frames;
        lat_ms.reserve(A.frames);
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();

        // --mode full: the "local map" of frame i = the keypoints of the two frames before it (what the ring held there), as
        // projected MapPoints: u, v = the keypoint, radius = 4 * th * scale[level] (RadiusByViewingCos, th = 1), window
        // [level - 1, level], its descriptor -- at most 3000.  Built once per ring position during the warm-up (the ring
        // repeats), so the timed frames pass ready arrays: in the reference they come from Frame::isInFrustum on the host.
        struct LocalMap { std::vector<float> uvr; std::vector<int8_t> lvl; std::vector<uint8_t> desc; int n = 0; };
        std::vector<LocalMap> lmap(full ? nring : 0);
        std::vector<std::vector<OrbxKeyPoint>> rkps(full ? nring : 0);
        std::vector<std::vector<uint8_t>> rdesc(full ? nring : 0);
        float sfl[ORBX_MAX_LEVELS] = {0};
        if (full) OX(orbx_scale_tables(ex, sfl, nullptr, nullptr, nullptr));
        OrbmProjParams pp3{3, 0.8f, 0, 100};
        using clk = std::chrono::steady_clock;
        int pendingTicket = -1, pendingIdx = -1;
        clk::time_point pendingT0{};
        auto finish = [&](int idx, int ticket, clk::time_point t0, int back) {   // collect frame idx: keypoints / descriptors, then its match table (`back` searches behind the newest)
            OrbxBatchView v{};
            OX(orbx_collect_view(ex, ticket, &v));
            int nm = 0, n = 0, nLast = 0;
            uint64_t cs = 0;
            for (int j = 0; j < P; j++) {
                if (bf) nm += v.nmatch[j];
                nLast = v.n[j]; n += nLast;
                cs = cs * 31 + (uint64_t)nLast * 1315423911ull;
                if (nLast > 0) { uint32_t d0; memcpy(&d0, v.desc + ((size_t)j * v.cap + (nLast - 1)) * 32, 4); cs ^= d0; }
            }
            // --dump: camera 0 of thread 0 writes its first eight timed frames with everything that came back for them -- the
            // test suite replays them through the CPU oracle (tests/test_gpu_example.py): this program holds no checker
            FILE* df = nullptr;
            if (robot == 0 && !A.dump.empty() && idx >= A.warmup && idx < A.warmup + 8) df = fopen(A.dump.c_str(), idx == A.warmup ? "wb" : "ab");
            const int n0 = v.n[0];
            if (df) {
                const int32_t hdr[6] = {idx, n0, A.w, A.h, bf ? 1 : (track ? 2 : 0), P};
                fwrite(hdr, sizeof hdr, 1, df);
                const uint8_t* fr0 = ring + (size_t)(idx % nring) * pitch;
                for (int y = 0; y < A.h; y++) fwrite(fr0 + (size_t)y * stride, 1, (size_t)A.w, df);
                fwrite(v.kps, sizeof(OrbxKeyPoint), (size_t)n0, df);
                fwrite(v.desc, 32, (size_t)n0, df);
                if (bf) { fwrite(v.match, 4, (size_t)n0, df); fwrite(v.nmatch, 4, 1, df); }
            }
            if (full && rkps[idx % nring].empty() && n0 > 0) {   // (warm-up, first pass over the ring)
                rkps[idx % nring].assign(v.kps, v.kps + n0);
                rdesc[idx % nring].assign(v.desc, v.desc + (size_t)n0 * 32);
                const int r = idx % nring, a1 = (r + nring - 1) % nring, a2 = (r + nring - 2) % nring;
                for (int rr = 0; rr < nring; rr++) {   // any ring position whose two predecessors are known now gets its local map
                    const int p1 = (rr + nring - 1) % nring, p2 = (rr + nring - 2) % nring;
                    if (lmap[rr].n || rkps[p1].empty() || rkps[p2].empty()) continue;
                    LocalMap& lm = lmap[rr];
                    for (int src = 0; src < 2 && lm.n < 3000; src++) {
                        const auto& K = rkps[src ? p2 : p1]; const auto& D = rdesc[src ? p2 : p1];
                        for (size_t k = 0; k < K.size() && lm.n < 3000; k++, lm.n++) {
                            const int lv = K[k].octave;
                            lm.uvr.push_back(K[k].x); lm.uvr.push_back(K[k].y); lm.uvr.push_back(4.0f * sfl[lv]);
                            lm.lvl.push_back((int8_t)(lv - 1)); lm.lvl.push_back((int8_t)lv);
                            lm.desc.insert(lm.desc.end(), D.begin() + k * 32, D.begin() + (k + 1) * 32);
                        }
                    }
                }
                (void)a1; (void)a2;
            }
            OX(orbx_release(ex, ticket));
            if (track && idx > 0) {
                const int32_t *assign, *nmp; int np, c2;
                OX(orbm_track_results(fs, back, &assign, &nmp, &np, &c2));
                for (int j = 0; j < P; j++) nm += nmp[j];
                if (nLast > 0) cs ^= (uint64_t)(uint32_t)assign[(size_t)(P - 1) * c2 + nLast - 1] << 32;
                if (df) { fwrite(assign, 4, (size_t)n0, df); fwrite(nmp, 4, 1, df); }
            }
            if (df) fclose(df);
            if (full) {
                LocalMap& lm = lmap[idx % nring];
                if (lm.n > 0) {   // Tracking::SearchLocalPoints (Tracking.cc:1242-1249): the second search of the iteration, against the same resident frame
                    OX(orbm_track_local_points(fs, (idx & 3) * P, &pp3, lm.uvr.data(), lm.lvl.data(), lm.desc.data(), nullptr, nullptr, lm.n, nullptr));
                    const int32_t *a3, *n3; int np3, c3;
                    OX(orbm_track_results(fs, 0, &a3, &n3, &np3, &c3));
                    nm += n3[0];
                    cs ^= (uint64_t)(uint32_t)n3[0] * 2654435761ull;
                }
            }
            const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            if (idx >= A.warmup) {
                lat_ms.push_back(ms);
                st.frames += P; st.keypoints += n; st.matches += nm; st.checksum = st.checksum * 1099511628211ull ^ cs;
                live_frames.fetch_add(P, std::memory_order_relaxed); live_kps.fetch_add(n, std::memory_order_relaxed); live_matches.fetch_add(nm, std::memory_order_relaxed);
            }
        };
        const auto tStart = clk::now();
        clk::time_point tTimed = tStart;
        for (int i = 0; i < total; i++) {
            if (i == A.warmup) tTimed = clk::now();
            const uint8_t* img[8];
            for (int j = 0; j < P; j++) img[j] = ring + (size_t)(j * nring + i % nring) * pitch;
            const auto t0 = clk::now();
            int ticket = -1;
            OX(orbx_submit_batch(ex, img, P, A.w, A.h, stride, &so, &ticket));
            const auto t1 = clk::now();
            if (track) {
                // camera j keeps its last four frames in slots (i & 3) * P + j: one build for the call's P frames
                int32_t cur[8], last[8];
                for (int j = 0; j < P; j++) { cur[j] = (i & 3) * P + j; last[j] = ((i - 1) & 3) * P + j; }
                OX(orbm_frameset_build_from_extractor(fs, cur[0], ex));
                if (A.depth == 2 && !A.attach) {
                    // the search of the PREVIOUS frame is collected below before this frame's search goes out (two
                    // searches never share a result set here), while this frame's extraction already runs beside it
                    if (pendingTicket >= 0) { finish(pendingIdx, pendingTicket, pendingT0, 0); pendingTicket = -1; if (failed) return; }
                }
                if (i > 0) OX(orbm_track_frames(fs, &pp, 15.0f, cur, last, P));
            }
            const auto t2 = clk::now();
            if (i >= A.warmup) { us_submit += std::chrono::duration<double, std::micro>(t1 - t0).count(); us_enqueue += std::chrono::duration<double, std::micro>(t2 - t1).count(); }
            if (A.depth == 2) {
                if (pendingTicket >= 0) { finish(pendingIdx, pendingTicket, pendingT0, track && i > 0 ? 1 : 0); if (failed) return; }  // (attached: this frame's search is already in the queue behind it)
                pendingTicket = ticket; pendingIdx = i; pendingT0 = t0;
            } else {
                finish(i, ticket, t0, 0);
                if (failed) return;
            }
        }
        if (pendingTicket >= 0) finish(pendingIdx, pendingTicket, pendingT0, 0);
        us_submit /= std::max(1, A.frames); us_enqueue /= std::max(1, A.frames);
        st.frames = (int64_t)lat_ms.size() * P;
        st.ns_busy = std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - tTimed).count();
        if (fs) orbm_frameset_destroy(fs);
        if (m) orbm_destroy(m);
        if (A.pinned) orbx_host_free(ex, ring);
        orbx_destroy(ex);
    }
};

static double median(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

}  // namespace

int main(int argc, char** argv)
{
    Args A;
    for (int i = 1; i < argc; i++) {
        const std::string k = argv[i];
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (k == "--gpus") A.gpus = atoi(val()); else if (k == "--robots") A.robots = atoi(val()); else if (k == "--per-call") A.per_call = atoi(val()); else if (k == "--frames") A.frames = atoi(val());
        else if (k == "--warmup") A.warmup = atoi(val()); else if (k == "--mode") A.mode = val(); else if (k == "--depth") A.depth = atoi(val());
        else if (k == "--attach") A.attach = atoi(val()); else if (k == "--pinned") A.pinned = atoi(val()); else if (k == "--w") A.w = atoi(val());
        else if (k == "--h") A.h = atoi(val()); else if (k == "--nfeat") A.nfeat = atoi(val()); else if (k == "--interval") A.interval = atoi(val());
        else if (k == "--dump") A.dump = val();
        else if (k == "--batch") A.batch = atoi(val()); else if (k == "--pool") A.pool = atoi(val()); else if (k == "--steps") A.steps = atoi(val());
        else if (k == "--hub") A.hub = atoi(val()); else if (k == "--hub-wait") A.hub_wait = atoi(val());
        else if (k == "--json") A.json = true;
        else { fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    const int ndev = orbx_device_count();
    if (ndev < 1) { fprintf(stderr, "no HIP device: the ORB front-end has no CPU fallback\n"); return 3; }
    if (A.gpus < 1 || A.gpus > ndev) { fprintf(stderr, "--gpus %d but %d device(s) visible\n", A.gpus, ndev); return 2; }
    if (A.robots < A.gpus) A.robots = A.gpus;
    if (A.mode != "track" && A.mode != "bf" && A.mode != "extract" && A.mode != "full" && A.mode != "batch") { fprintf(stderr, "--mode track|bf|extract|full|batch\n"); return 2; }
    const bool batchMode = A.mode == "batch";
    if (batchMode && (A.batch < 1 || A.batch > 256 || A.pool < 1 || A.steps < 1 || A.per_call != 1 || A.depth != 1 || A.hub != 0)) { fprintf(stderr, "--mode batch: --batch 1..256 --pool >= 1 --steps >= 1, no --per-call / --depth / --hub\n"); return 2; }
    if (batchMode && A.warmup == 30) A.warmup = 5;
    if (A.mode == "full" && (A.per_call != 1 || A.depth != 1)) { fprintf(stderr, "--mode full: one camera per call, one ticket deep\n"); return 2; }
    if (A.mode == "full" && A.warmup < 12) A.warmup = 12;   // the ring's local maps are built during the warm-up
    if (A.depth != 1 && A.depth != 2) { fprintf(stderr, "--depth 1|2\n"); return 2; }
    if (A.per_call < 1 || A.per_call > 8 || (A.per_call > 1 && A.mode == "bf")) { fprintf(stderr, "--per-call 1..8 (> 1: modes track and extract -- the brute-force match is against the SAME camera's previous frame)\n"); return 2; }

    // RCCL: one communicator per GPU in this one process (the reference is one process, one thread per robot)
    std::vector<ncclComm_t> comms(A.gpus);
    std::vector<int> devs(A.gpus);
    for (int d = 0; d < A.gpus; d++) devs[d] = d;
    if (ncclCommInitAll(comms.data(), A.gpus, devs.data()) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 4; }
    std::vector<hipStream_t> cstream(A.gpus);
    std::vector<Stats*> d_send(A.gpus), d_recv(A.gpus);
    for (int d = 0; d < A.gpus; d++) {
        if (hipSetDevice(d) != hipSuccess || hipStreamCreateWithFlags(&cstream[d], hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(&d_send[d], sizeof(Stats)) != hipSuccess || hipMalloc(&d_recv[d], sizeof(Stats) * A.gpus) != hipSuccess) { fprintf(stderr, "hip setup failed\n"); return 4; }
    }

    if (A.hub == -1) A.hub = std::max(1, ((A.robots + A.gpus - 1) / A.gpus + 3) / 4);   // --hub -1: four hubs per GPU (a queue each)
    if (A.hub < 0 || A.hub > orbslamm::CameraHub::kMaxCameras || (A.hub > 0 && ((A.mode != "track" && A.mode != "full") || A.per_call != 1 || A.depth != 1))) { fprintf(stderr, "--hub 1..8: mode track or full, one camera per robot, one ticket deep\n"); return 2; }
    // --hub P: on every GPU the robots are dealt to hubs of P cameras in the order they come
    std::vector<std::unique_ptr<orbslamm::CameraHub>> hubs;
    std::vector<Robot> robots(A.robots);
    if (A.hub > 0) {
        for (int d = 0; d < A.gpus; d++) {
            int k = 0;
            for (int r = d; r < A.robots; r += A.gpus, k++) {
                if (k % A.hub == 0) {
                    int left = 0; for (int q = r; q < A.robots; q += A.gpus) left++;
                    orbslamm::CameraHub::Config c;
                    c.prm = OrbxParams{A.nfeat, 1.2f, 8, 20, 7}; c.w = A.w; c.h = A.h; c.cameras = std::min(A.hub, left); c.device = d; c.wait_us = A.hub_wait;
                    const float K[4] = {718.856f, 718.856f, 607.1928f, 185.2157f};
                    memcpy(c.K, K, sizeof K);
                    hubs.emplace_back(new orbslamm::CameraHub());
                    const int rc = hubs.back()->open(c);
                    if (rc) { fprintf(stderr, "hub on GPU %d: open failed (%d): %s\n", d, rc, orbx_last_error()); return 4; }
                }
                robots[r].hub = hubs.back().get(); robots[r].cam = k % A.hub;
            }
        }
    }
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<std::thread> th;
    for (int r = 0; r < A.robots; r++) { robots[r].robot = r; robots[r].device = r % A.gpus; robots[r].a = &A; }
    for (int r = 0; r < A.robots; r++) th.emplace_back([&, r] { if (batchMode) robots[r].run_batch(ready, go); else if (robots[r].hub) robots[r].run_hub(ready, go); else robots[r].run(ready, go); if (robots[r].failed) ready.fetch_add(1 << 16); });
    while ((ready.load() & 0xFFFF) + (ready.load() >> 16) < A.robots) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);

    // the reporting loop: every --interval frames (of robot 0) gather the per-GPU counters with one ncclAllGather
    std::vector<Stats> gathered(A.gpus);
    int gathers = 0;
    auto gather = [&]() -> bool {
        for (int d = 0; d < A.gpus; d++) {
            Stats s{};
            for (int r = d; r < A.robots; r += A.gpus) { s.frames += robots[r].live_frames.load(); s.keypoints += robots[r].live_kps.load(); s.matches += robots[r].live_matches.load(); }
            if (hipSetDevice(d) != hipSuccess || hipMemcpyAsync(d_send[d], &s, sizeof s, hipMemcpyHostToDevice, cstream[d]) != hipSuccess) return false;
            if (hipStreamSynchronize(cstream[d]) != hipSuccess) return false;   // `s` is a stack variable
        }
        if (ncclGroupStart() != ncclSuccess) return false;
        for (int d = 0; d < A.gpus; d++)
            if (ncclAllGather(d_send[d], d_recv[d], sizeof(Stats), ncclChar, comms[d], cstream[d]) != ncclSuccess) return false;
        if (ncclGroupEnd() != ncclSuccess) return false;
        for (int d = 0; d < A.gpus; d++) { if (hipSetDevice(d) != hipSuccess || hipStreamSynchronize(cstream[d]) != hipSuccess) return false; }
        if (hipSetDevice(0) != hipSuccess || hipMemcpy(gathered.data(), d_recv[0], sizeof(Stats) * A.gpus, hipMemcpyDeviceToHost) != hipSuccess) return false;
        gathers++;
        return true;
    };
    int64_t nextReport = A.interval;
    bool anyFailed = false;
    for (;;) {
        bool done = true;
        for (auto& r : robots) done &= r.failed || r.live_frames.load() >= (batchMode ? (int64_t)A.batch * A.steps : (int64_t)A.frames * A.per_call);
        if (done) break;
        if (A.interval > 0 && robots[0].live_frames.load() >= nextReport) {
            if (!gather()) { fprintf(stderr, "stats gather failed\n"); anyFailed = true; break; }
            nextReport += A.interval;
            if (!A.json) { int64_t f = 0, mm = 0; for (auto& s : gathered) { f += s.frames; mm += s.matches; } fprintf(stderr, "[gather %d] %lld frames, %lld matches over %d GPU(s)\n", gathers, (long long)f, (long long)mm, A.gpus); }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(500));
    }
    for (auto& t : th) t.join();
    const double wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (auto& r : robots) anyFailed |= r.failed;
    if (!anyFailed && !gather()) { fprintf(stderr, "final stats gather failed\n"); anyFailed = true; }
    for (int d = 0; d < A.gpus; d++) { (void)hipSetDevice(d); (void)hipFree(d_send[d]); (void)hipFree(d_recv[d]); (void)hipStreamDestroy(cstream[d]); ncclCommDestroy(comms[d]); }
    if (anyFailed) return 1;

    std::vector<double> all;
    int64_t frames = 0, kps = 0, matches = 0, ns_max = 0;
    uint64_t cs = 0;
    for (auto& r : robots) { all.insert(all.end(), r.lat_ms.begin(), r.lat_ms.end()); frames += r.st.frames; kps += r.st.keypoints; matches += r.st.matches; ns_max = std::max(ns_max, r.st.ns_busy); cs ^= r.st.checksum; }
    int64_t gframes = 0; for (auto& s : gathered) gframes += s.frames;
    double mean = 0; for (double v : all) mean += v; mean /= std::max<size_t>(1, all.size());
    const double fps = frames / (ns_max * 1e-9);
    double usSub = 0, usEnq = 0, hubBatch = 0;
    for (auto& r : robots) hubBatch += r.batch_mean / A.robots;
    for (auto& r : robots) { usSub += r.us_submit / A.robots; usEnq += r.us_enqueue / A.robots; }
    std::sort(all.begin(), all.end());
    const double p95 = all.empty() ? 0 : all[all.size() * 95 / 100], p99 = all.empty() ? 0 : all[all.size() * 99 / 100], pmax = all.empty() ? 0 : all.back();
    if (A.json) {
        printf("{\"mode\": \"%s\", \"batch\": %d, \"steps\": %d, \"gpus\": %d, \"robots\": %d, \"cameras_per_call\": %d, \"depth\": %d, \"attach\": %d, \"pinned\": %d, \"w\": %d, \"h\": %d, \"nfeat\": %d, \"frames_per_robot\": %d, "
               "\"frames_per_s\": %.1f, \"ms_median\": %.4f, \"ms_mean\": %.4f, \"ms_p95\": %.4f, \"ms_p99\": %.4f, \"ms_max\": %.4f, \"host_us_submit\": %.1f, \"host_us_enqueue\": %.1f, \"keypoints_mean\": %.1f, \"matches_mean\": %.1f, \"wall_s\": %.3f, "
               "\"rccl_allgathers\": %d, \"gathered_frames\": %lld, \"hub\": %d, \"hub_batch_mean\": %.2f, \"checksum\": \"%016llx\"}\n",
               A.mode.c_str(), batchMode ? A.batch : 0, batchMode ? A.steps : 0, A.gpus, A.robots, A.per_call, A.depth, A.attach, A.pinned, A.w, A.h, A.nfeat, A.frames, fps, median(all), mean, p95, p99, pmax, usSub, usEnq,
               (double)kps / std::max<int64_t>(1, frames), (double)matches / std::max<int64_t>(1, frames), wall_s, gathers, (long long)gframes, A.hub, hubBatch, (unsigned long long)cs);
    } else {
        // as the reference's examples end (mono_tum.cc:113-122)
        printf("-------\n\nmedian tracking time: %f ms\nmean tracking time: %f ms\n", median(all), mean);
        printf("%d robot(s) on %d GPU(s), mode %s: %lld frames, %.0f frames/s aggregate, %.1f keypoints and %.1f matches per frame; %d RCCL gathers (%lld frames seen)\n",
               A.robots, A.gpus, A.mode.c_str(), (long long)frames, fps, (double)kps / std::max<int64_t>(1, frames), (double)matches / std::max<int64_t>(1, frames), gathers, (long long)gframes);
    }
    return 0;
}
