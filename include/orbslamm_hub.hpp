// orbslamm_hub.hpp -- the live frames of several robots through ONE latency-mode chain (header only, over the C ABI).
//
// Reference shape: MultipleRobotsScenario/Examples/Monocular/mono_kitti.cc:83-101 -- one tracking thread per robot, each
// handing ITS camera's next frame to Tracking::GrabImageMonocular (SingleRobotScenario/src/Tracking.cc:240-267: Frame::Frame
// -> ExtractORB, UndistortKeyPoints, AssignFeaturesToGrid; TrackWithMotionModel -> SearchByProjection(Cur, Last), :925-936).
// With one extractor handle per robot every robot owns a chain of ten small kernels in a queue of its own; an MI355X
// executes about four such queues at a time (docs/experiments.md, round 4), so beyond four robots per GPU the chains
// queue up behind each other: 8 robots x 1 camera read 19.6 k frames/s, the same as 4.  The kernels of a chain are a
// few workgroups each -- eight frames in ONE chain take twice as long as one, not eight times.  CameraHub does that
// without changing the robots' shape: every robot thread still calls `track(camera, frame, ...)` and gets ITS
// keypoints, descriptors and match table back when the call returns; the frames of the robots that are waiting at that
// moment go through the chain together (orbx_submit_batch with B = cameras present, one frame-set build, one
// orbm_track_frames over the pairs present).  Results are the same bytes a handle per robot produces (the kernels
// treat the frames of a batch independently; tests/test_gpu_example.py replays a hub camera on the CPU byte for byte).
//
// Batching rule: the first thread to find no batch being assembled becomes its leader, waits until every camera that is
// a member of the hub has a frame waiting or `wait_us` microseconds have passed (cameras that free-run fall into
// lockstep after one batch; a 30 Hz camera pays at most wait_us once per frame), takes what is there, runs the chain and
// hands each waiting thread a view of its part; each thread copies its own results out, the next leader returns the ticket.
// One batch is in flight per hub; for more than eight cameras per GPU use two to four hubs (a queue each).
//
// Frames of one batch must be of one kind: all pinned in the device layout (orbx_host_alloc_frames / orbx_host_register,
// `stride` = the handle's) or all pageable.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>

#include "orbslamm_hip.h"

namespace orbslamm {

class CameraHub {
public:
    static constexpr int kMaxCameras = 8;   // orbx_create_live's ceiling

    struct Config {
        OrbxParams prm{2000, 1.2f, 8, 20, 7};
        int w = 0, h = 0, cameras = 1, device = 0;
        bool track = true;                       // false: extraction only
        float K[4] = {0, 0, 0, 0}, D[5] = {0, 0, 0, 0, 0};   // fx, fy, cx, cy; k1, k2, p1, p2, k3 (Frame.cc:120-150)
        float th = 15.0f;                        // SearchByProjection(Cur, Last, th, mono): Tracking.cc:931
        OrbmProjParams pp{4, 0.9f, 1, 100};      // mode 4, nnratio 0.9, rotation check, TH_HIGH
        int wait_us = 40;                        // how long a leader waits for the other members' frames
        int sleep_us = 0;                        // > 0: a thread waiting for its results sleeps this long between two looks instead of
                                                 // spinning (frees its core for the robot's other threads; costs up to that much latency)
    };
    struct Result {
        int n = 0;            // keypoints of this frame
        int nmatches = -1;    // SearchByProjection's return value; -1: no previous frame of this camera in the set (no search)
        int batch = 0;        // how many cameras shared the chain with this frame
    };

    CameraHub() = default;
    CameraHub(const CameraHub&) = delete;
    CameraHub& operator=(const CameraHub&) = delete;
    ~CameraHub() { close(); }

    int open(const Config& c)
    {
        close();
        cfg_ = c;
        if (c.cameras < 1 || c.cameras > kMaxCameras || c.w < 1 || c.h < 1) return ORBX_E_INVALID;
        int rc = orbx_create_live(&c.prm, c.w, c.h, c.cameras, c.device, &ex_);
        if (rc) return rc;
        cap_ = orbx_max_keypoints(ex_);
        if (c.track) {
            if ((rc = orbm_create(c.device, &m_))) { close(); return rc; }
            float sf[ORBX_MAX_LEVELS];
            if ((rc = orbx_scale_tables(ex_, sf, nullptr, nullptr, nullptr))) { close(); return rc; }
            const float bounds[4] = {0.f, (float)c.w, 0.f, (float)c.h};   // undistorted image bounds of an undistorted camera (Frame.cc:430-462)
            OrbmGrid g{0.f, 0.f, 64.f / (float)c.w, 48.f / (float)c.h, 64, 48};
            // C (C + 2) slots: a batch takes a run of B <= C consecutive slots that holds NO camera's newest frame -- with at
            // most C such slots in the ring a free run of C always exists -- so a camera that falls several batches behind the
            // others (its thread descheduled, a long local-map search) still finds its previous frame when it comes back.
            // (Round 4's ring of 4 C slots handed the oldest slot out whatever it held: one frame in ~10^4 lost its previous
            // frame that way under --hub-wait 0, found by tests/soak/fuzz_multi_robot.py.)
            nslots_ = c.cameras * (c.cameras + 2);
            if ((rc = orbm_frameset_create(m_, nslots_, cap_, c.K, c.D, &g, bounds, sf, orbx_levels(ex_), &fs_))) { close(); return rc; }
            if ((rc = orbm_frameset_attach(fs_, ex_))) { close(); return rc; }
        }
        for (int s = 0; s < kMaxSlots; s++) { slotCam_[s] = -1; slotSeq_[s] = 0; }
        for (int j = 0; j < kMaxCameras; j++) { lastSlot_[j] = -1; seq_[j] = 0; lastN_[j] = 0; req_[j].state.store(0); }
        cursor_ = 0; pendingTicket_ = -1;   // (a hub reopened with another camera count starts its ring afresh)
        return ORBX_OK;
    }

    void close()
    {
        if (ex_) release_pending();
        if (fs_) { orbm_frameset_destroy(fs_); fs_ = nullptr; }
        if (m_) { orbm_destroy(m_); m_ = nullptr; }
        if (ex_) { orbx_destroy(ex_); ex_ = nullptr; }
        members_.store(0); waiting_.store(0); leader_.store(false);
    }

    int cap() const { return cap_; }             // rows a caller's arrays need
    orbx_t* extractor() const { return ex_; }    // e.g. for orbx_host_alloc_frames
    // a camera that stops sending frames leaves, so that the others' batches do not wait for it
    void leave(int cam) { if (cam >= 0 && cam < cfg_.cameras) members_.fetch_and(~(1u << cam)); }

    // Called by camera `cam`'s own thread (one call at a time per camera).  Blocks until this frame's results are in the
    // caller's arrays: kps[cap()], desc[cap() * 32], assign[cap()] (assign[t] = index of the PREVIOUS frame's feature whose
    // MapPoint current feature t took, -1 = none; may be NULL).  Returns an ORBX_* code.
    int track(int cam, const uint8_t* frame, int stride, OrbxKeyPoint* kps, uint8_t* desc, int32_t* assign, Result* res)
    {
        if (!ex_ || cam < 0 || cam >= cfg_.cameras || !frame || !kps || !desc || !res) return ORBX_E_INVALID;
        Request& r = req_[cam];
        r.frame = frame; r.stride = stride;
        r.state.store(1, std::memory_order_relaxed);
        members_.fetch_or(1u << cam, std::memory_order_relaxed);
        waiting_.fetch_or(1u << cam, std::memory_order_release);
        for (;;) {
            const int st = r.state.load(std::memory_order_acquire);
            if (st >= 2) break;
            if (!leader_.exchange(true, std::memory_order_acquire)) {
                // (a batch that was being served while this thread won the flag may have completed this request meanwhile)
                if (r.state.load(std::memory_order_acquire) < 2) lead();
                leader_.store(false, std::memory_order_release);
                continue;
            }
            idle();
        }
        // state 2: this thread's part of the batch is published; copy it out (the next leader returns the ticket once all have)
        const int rc = r.rc;
        if (rc == ORBX_OK) {
            res->n = r.n; res->nmatches = r.nmatches; res->batch = r.batch;
            std::memcpy(kps, r.kps, (size_t)r.n * sizeof(OrbxKeyPoint));
            std::memcpy(desc, r.desc, (size_t)r.n * 32);
            if (assign) {
                if (r.assign) std::memcpy(assign, r.assign, (size_t)r.n * 4);
                else for (int t = 0; t < r.n; t++) assign[t] = -1;
            }
        }
        r.state.store(0, std::memory_order_relaxed);
        readers_.fetch_sub(1, std::memory_order_release);
        return rc;
    }

    // Tracking::SearchLocalPoints' search (Tracking.cc:1242-1249 -> ORBmatcher::SearchByProjection(Frame, local MapPoints, th),
    // ORBmatcher.cc:45-129) against the frame camera `cam` sent LAST, still resident in the hub's frame set -- the arguments
    // of orbm_track_local_points; assign[cap()] receives, per feature of that frame, the index of the MapPoint it took or -1.
    // The searches of a hub run one after the other between two batches; ORBX_E_INVALID if the frame has left the set.
    int search_local_points(int cam, const OrbmProjParams* pp, const float* q_uvr, const int8_t* q_lvl, const uint8_t* qdesc,
                            const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq, const uint8_t* t_occ, int32_t* assign, int* nmatches)
    {
        if (!fs_ || cam < 0 || cam >= cfg_.cameras || !pp || !assign) return ORBX_E_INVALID;
        LocalRequest& q = lreq_[cam];
        q.pp = pp; q.uvr = q_uvr; q.lvl = q_lvl; q.desc = qdesc; q.valid = qvalid; q.obs = q_obs_pos; q.nq = nq; q.occ = t_occ; q.assign = assign;
        q.state.store(1, std::memory_order_relaxed);
        lwaiting_.fetch_or(1u << cam, std::memory_order_release);
        // whoever holds the leader's flag serves it -- a leader waiting for the other cameras' frames does (this camera's next
        // frame will not come before its search has returned), else this thread takes the flag for the search alone
        while (q.state.load(std::memory_order_acquire) < 2) {
            if (!leader_.exchange(true, std::memory_order_acquire)) {
                serve_local_searches();
                leader_.store(false, std::memory_order_release);
                continue;
            }
            idle();
        }
        q.state.store(0, std::memory_order_relaxed);
        if (nmatches) *nmatches = q.nmatches;
        return q.rc;
    }

private:
    void idle() const
    {
        if (cfg_.sleep_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(cfg_.sleep_us));
        else std::this_thread::yield();
    }
    struct Request {
        std::atomic<int> state{0};   // 0 idle, 1 waiting for a batch, 2 served (view published)
        const uint8_t* frame = nullptr; int stride = 0;
        const OrbxKeyPoint* kps = nullptr; const uint8_t* desc = nullptr; const int32_t* assign = nullptr;
        int n = 0, nmatches = -1, batch = 0, rc = 0;
    };

    struct LocalRequest {
        std::atomic<int> state{0};   // 0 idle, 1 waiting, 2 served
        const OrbmProjParams* pp = nullptr; const float* uvr = nullptr; const int8_t* lvl = nullptr; const uint8_t* desc = nullptr;
        const uint8_t *valid = nullptr, *obs = nullptr, *occ = nullptr; int nq = 0; int32_t* assign = nullptr;
        int nmatches = 0, rc = 0;
    };

    // (leader only) the local-map searches that wait, one after the other on the chain; the tables go straight into the
    // waiting callers' arrays
    void serve_local_searches()
    {
        const uint32_t take = lwaiting_.exchange(0, std::memory_order_acquire);
        if (!take) return;
        // the previous batch's threads copy their match tables out of a result set that four searches from now is recycled
        while (readers_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
        for (int cam = 0; cam < cfg_.cameras; cam++) {
            if (!(take >> cam & 1)) continue;
            LocalRequest& q = lreq_[cam];
            const int slot = lastSlot_[cam];
            int rc = slot >= 0 && slotCam_[slot] == cam && slotSeq_[slot] == seq_[cam] - 1 ? ORBX_OK : ORBX_E_INVALID;
            const int32_t *a = nullptr, *nm = nullptr;
            int np = 0, c2 = 0;
            if (!rc) rc = orbm_track_local_points(fs_, slot, q.pp, q.uvr, q.lvl, q.desc, q.valid, q.obs, q.nq, q.occ);
            if (!rc) rc = orbm_track_results(fs_, 0, &a, &nm, &np, &c2);
            if (!rc) { std::memcpy(q.assign, a, (size_t)lastN_[cam] * 4); q.nmatches = nm[0]; }
            q.rc = rc;
            q.state.store(2, std::memory_order_release);
        }
    }

    void lead()
    {
        using clk = std::chrono::steady_clock;
        const auto t0 = clk::now();
        // every member has a frame waiting, or wait_us are over
        for (;;) {
            const uint32_t w = waiting_.load(std::memory_order_acquire), mset = members_.load(std::memory_order_relaxed);
            if ((w & mset) == mset) break;
            if (std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count() >= cfg_.wait_us) break;
            serve_local_searches();
            std::this_thread::yield();
        }
        serve_local_searches();
        release_pending();   // the previous batch's readers had the wait above to copy their parts
        const uint32_t take = waiting_.exchange(0, std::memory_order_acquire);
        int cams[kMaxCameras], B = 0;
        for (int j = 0; j < cfg_.cameras; j++) if (take >> j & 1) cams[B++] = j;
        if (B == 0) return;
        const uint8_t* imgs[kMaxCameras];
        for (int p = 0; p < B; p++) imgs[p] = req_[cams[p]].frame;
        int rc = ORBX_OK, ticket = -1, np = 0;
        int32_t cur[kMaxCameras], last[kMaxCameras], pairOf[kMaxCameras];
        bool strideOk = true;
        for (int p = 1; p < B; p++) strideOk &= req_[cams[p]].stride == req_[cams[0]].stride;
        if (!strideOk) rc = ORBX_E_INVALID;
        if (!rc) rc = orbx_submit_batch(ex_, imgs, B, cfg_.w, cfg_.h, req_[cams[0]].stride, nullptr, &ticket);
        if (!rc && fs_) {
            // a ring of slots: this batch's frames take the next run of B consecutive ones that holds no camera's newest frame
            // (the `last` operands of this batch's searches and the frames the absent cameras will come back to)
            auto newest = [&](int s) { return slotCam_[s] >= 0 && lastSlot_[slotCam_[s]] == s; };
            int slot0 = -1;
            for (int s = cursor_, tries = 0; slot0 < 0 && tries <= 2 * nslots_; tries++) {
                if (s + B > nslots_) { s = 0; continue; }
                int bad = -1;
                for (int p = 0; p < B; p++) if (newest(s + p)) bad = s + p;
                if (bad < 0) slot0 = s; else s = bad + 1;
            }
            if (slot0 < 0) slot0 = 0;   // (cannot happen: C newest frames leave a free run of C in C (C + 2) slots)
            cursor_ = slot0 + B;
            rc = orbm_frameset_build_from_extractor(fs_, slot0, ex_);
            for (int p = 0; p < B; p++) pairOf[p] = -1;
            for (int p = 0; p < B && !rc; p++) {
                const int j = cams[p], ls = lastSlot_[j];
                const bool havePrev = ls >= 0 && slotCam_[ls] == j && slotSeq_[ls] == seq_[j] - 1 && (ls < slot0 || ls >= slot0 + B);
                if (havePrev) { cur[np] = slot0 + p; last[np] = ls; pairOf[p] = np++; }
            }
            if (!rc && np > 0) rc = orbm_track_frames(fs_, &cfg_.pp, cfg_.th, cur, last, np);
            // the ring remembers a frame only if its Frame tail was built and its search went out: after a failed batch the
            // cameras' next frames have no previous frame (nmatches = -1) instead of one that was never built
            for (int p = 0; p < B; p++) {
                const int j = cams[p];
                if (!rc) { slotCam_[slot0 + p] = j; slotSeq_[slot0 + p] = seq_[j]; lastSlot_[j] = slot0 + p; }
                else { slotCam_[slot0 + p] = -1; lastSlot_[j] = -1; }
                seq_[j]++;
            }
        }
        OrbxBatchView v{};
        if (!rc) rc = orbx_collect_view(ex_, ticket, &v);
        const int32_t *assign = nullptr, *nmp = nullptr;
        int npOut = 0, c2 = 0;
        if (!rc && np > 0) rc = orbm_track_results(fs_, 0, &assign, &nmp, &npOut, &c2);
        readers_.store(B, std::memory_order_relaxed);
        for (int p = 0; p < B; p++) {
            Request& r = req_[cams[p]];
            r.rc = rc; r.batch = B;
            if (!rc) {
                r.n = v.n[p]; lastN_[cams[p]] = v.n[p]; r.kps = v.kps + (size_t)p * v.cap; r.desc = v.desc + (size_t)p * v.cap * 32;
                const int q = fs_ ? pairOf[p] : -1;
                r.assign = q >= 0 ? assign + (size_t)q * c2 : nullptr;
                r.nmatches = q >= 0 ? nmp[q] : -1;
            }
        }
        for (int p = 0; p < B; p++) req_[cams[p]].state.store(2, std::memory_order_release);
        // The views live in the ticket's pinned block, and the leader itself copies its part only after this returns: the
        // ticket goes back at the start of the NEXT batch (or in close()), once readers_ has fallen to zero.
        pendingTicket_ = ticket;
    }

    void release_pending()
    {
        while (readers_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
        if (pendingTicket_ >= 0) (void)orbx_release(ex_, pendingTicket_);
        pendingTicket_ = -1;
    }

    Config cfg_{};
    orbx_t* ex_ = nullptr; orbm_t* m_ = nullptr; orbm_frameset_t* fs_ = nullptr;
    int cap_ = 0, nslots_ = 0, cursor_ = 0;
    static constexpr int kMaxSlots = kMaxCameras * (kMaxCameras + 2);
    int slotCam_[kMaxSlots]; int64_t slotSeq_[kMaxSlots];
    int lastSlot_[kMaxCameras]; int64_t seq_[kMaxCameras]; int lastN_[kMaxCameras];
    Request req_[kMaxCameras];
    LocalRequest lreq_[kMaxCameras];
    std::atomic<uint32_t> members_{0}, waiting_{0}, lwaiting_{0};
    std::atomic<bool> leader_{false};
    std::atomic<int> readers_{0};
    int pendingTicket_ = -1;
};

}  // namespace orbslamm
