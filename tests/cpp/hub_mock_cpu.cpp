// CPU test of orbslamm::CameraHub's host logic (include/orbslamm_hub.hpp) against a MOCK of the C ABI entries it calls:
// leader election, batching, the frame set's slot ring, ticket lifetime.  Built with -fsanitize=thread by
// tests/test_adapter_cpu.py.  The mock checks what the library requires of its caller and what the hub promises:
//   * one thread at a time inside the (extractor, attached frame set) pair;
//   * at most three tickets outstanding, every ticket released exactly once, never while a reader may still copy from it
//     (a released ticket's block is overwritten with 0xEE, a reader would return wrong rows);
//   * orbm_track_frames pairs (cur, last) are slots holding frame s and frame s - 1 of the SAME camera;
//   * every thread gets the rows of ITS camera and ITS frame.
// A "frame" here is four ints: camera, sequence number, and two spare words.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "orbslamm_hub.hpp"

static std::atomic<int> g_fail{0};
#define MOCK_CHECK(c) do { if (!(c)) { std::fprintf(stderr, "MOCK FAIL %s:%d %s\n", __FILE__, __LINE__, #c); g_fail++; } } while (0)

namespace {
constexpr int kCap = 64, kSlots = 80, kTickets = 3;   // kSlots: C (C + 2) for a hub of eight cameras
struct Guard {   // the library's calls on one (extractor, frame set) pair come from one thread at a time
    static std::atomic<bool> busy;
    Guard() { MOCK_CHECK(!busy.exchange(true)); }
    ~Guard() { busy.store(false); }
};
std::atomic<bool> Guard::busy{false};

struct Ticket {
    int state = 0;   // 0 free, 1 in flight, 2 viewed
    int B = 0, cam[8], seq[8];
    std::vector<int32_t> n; std::vector<OrbxKeyPoint> kps; std::vector<uint8_t> desc;
};
struct Mock {
    int maxB = 0, nextTicket = 0, lastB = 0, lastCam[8], lastSeq[8];
    Ticket tk[kTickets];
    int slotCam[kSlots], slotSeq[kSlots], nslots = 0;
    int np = 0, pairCam[8], pairSeq[8];
    std::vector<int32_t> assign, nm;
    long batches = 0, frames = 0;
} M;
}  // namespace

struct orbx_handle { int dummy; };
struct orbm_handle { int dummy; };
struct orbm_frameset { int dummy; };
static orbx_handle g_ex; static orbm_handle g_m; static orbm_frameset g_fs;

extern "C" {
int orbx_create_live(const OrbxParams*, int, int, int max_cameras, int, orbx_t** out)
{
    M = Mock(); M.maxB = max_cameras; for (int& c : M.slotCam) c = -1;
    *out = &g_ex; return ORBX_OK;
}
void orbx_destroy(orbx_t*) { for (const Ticket& t : M.tk) MOCK_CHECK(t.state == 0); }
int orbx_max_keypoints(const orbx_t*) { return kCap; }
int orbx_levels(const orbx_t*) { return 8; }
int orbx_scale_tables(const orbx_t*, float* s, float*, float*, float*) { for (int i = 0; i < 8; i++) s[i] = 1.f + i; return ORBX_OK; }
int orbm_create(int, orbm_t** out) { *out = &g_m; return ORBX_OK; }
void orbm_destroy(orbm_t*) {}
int orbm_frameset_create(orbm_t*, int slots, int cap, const float*, const float*, const OrbmGrid*, const float*, const float*, int, orbm_frameset_t** out)
{
    MOCK_CHECK(slots <= kSlots && cap == kCap); M.nslots = slots; *out = &g_fs; return ORBX_OK;
}
int orbm_frameset_destroy(orbm_frameset_t*) { return ORBX_OK; }
int orbm_frameset_attach(orbm_frameset_t*, orbx_t*) { return ORBX_OK; }

int orbx_submit_batch(orbx_t*, const uint8_t* const* imgs, int B, int, int, int, const OrbxStreamOpts*, int* ticket)
{
    Guard g;
    MOCK_CHECK(B >= 1 && B <= M.maxB);
    Ticket& t = M.tk[M.nextTicket % kTickets];
    MOCK_CHECK(t.state == 0);   // at most kTickets outstanding
    t.state = 1; t.B = B;
    t.n.assign(B, 0); t.kps.assign((size_t)B * kCap, OrbxKeyPoint{}); t.desc.assign((size_t)B * kCap * 32, 0);
    for (int p = 0; p < B; p++) {
        int32_t hdr[2]; std::memcpy(hdr, imgs[p], sizeof hdr);
        t.cam[p] = M.lastCam[p] = hdr[0]; t.seq[p] = M.lastSeq[p] = hdr[1];
        for (int q = 0; q < p; q++) MOCK_CHECK(t.cam[q] != t.cam[p]);   // a camera once per batch
        const int n = 10 + hdr[0] + hdr[1] % 5;
        t.n[p] = n;
        for (int i = 0; i < n; i++) {
            t.kps[(size_t)p * kCap + i].x = (float)hdr[0]; t.kps[(size_t)p * kCap + i].y = (float)hdr[1]; t.kps[(size_t)p * kCap + i].octave = i;
            std::memset(&t.desc[((size_t)p * kCap + i) * 32], (hdr[0] * 31 + hdr[1] + i) & 0xFF, 32);
        }
    }
    M.lastB = B; M.batches++; M.frames += B;
    *ticket = M.nextTicket++;
    return ORBX_OK;
}
int orbm_frameset_build_from_extractor(orbm_frameset_t*, int slot0, orbx_t*)
{
    Guard g;
    MOCK_CHECK(slot0 >= 0 && slot0 + M.lastB <= M.nslots);
    for (int p = 0; p < M.lastB; p++) { M.slotCam[slot0 + p] = M.lastCam[p]; M.slotSeq[slot0 + p] = M.lastSeq[p]; }
    return ORBX_OK;
}
int orbm_track_frames(orbm_frameset_t*, const OrbmProjParams*, float, const int32_t* cur, const int32_t* last, int np)
{
    Guard g;
    MOCK_CHECK(np >= 1 && np <= M.lastB);
    M.np = np; M.assign.assign((size_t)np * kCap, -1); M.nm.assign(np, 0);
    for (int q = 0; q < np; q++) {
        MOCK_CHECK(cur[q] >= 0 && cur[q] < M.nslots && last[q] >= 0 && last[q] < M.nslots && cur[q] != last[q]);
        MOCK_CHECK(M.slotCam[cur[q]] == M.slotCam[last[q]]);            // the SAME camera's
        MOCK_CHECK(M.slotSeq[last[q]] == M.slotSeq[cur[q]] - 1);        // previous frame
        M.pairCam[q] = M.slotCam[cur[q]]; M.pairSeq[q] = M.slotSeq[cur[q]];
        M.nm[q] = M.pairSeq[q] * 7 + M.pairCam[q];
        for (int t = 0; t < kCap; t++) M.assign[(size_t)q * kCap + t] = M.pairCam[q] * 100000 + M.pairSeq[q] * 10 + t % 10;
    }
    return ORBX_OK;
}
int orbm_track_local_points(orbm_frameset_t*, int slot, const OrbmProjParams* pp, const float* q_uvr, const int8_t*, const uint8_t*, const uint8_t*, const uint8_t*, int nq, const uint8_t*)
{
    Guard g;
    MOCK_CHECK(slot >= 0 && slot < M.nslots && pp->mode == 3 && nq == 5);
    const int cam = (int)q_uvr[0], seq = (int)q_uvr[1];   // the caller says whose frame it expects in the slot
    MOCK_CHECK(M.slotCam[slot] == cam && M.slotSeq[slot] == seq);
    M.np = 1; M.assign.assign(kCap, -1); M.nm.assign(1, cam * 3 + seq);
    for (int t = 0; t < kCap; t++) M.assign[t] = 7000000 + cam * 100000 + seq * 10 + t % 10;
    return ORBX_OK;
}
int orbx_collect_view(orbx_t*, int ticket, OrbxBatchView* v)
{
    Guard g;
    Ticket& t = M.tk[ticket % kTickets];
    MOCK_CHECK(t.state == 1);
    t.state = 2;
    v->B = t.B; v->cap = kCap; v->n = t.n.data(); v->kps = t.kps.data(); v->desc = t.desc.data(); v->match = nullptr; v->nmatch = nullptr;
    return ORBX_OK;
}
int orbm_track_results(orbm_frameset_t*, int back, const int32_t** assign, const int32_t** nm, int* np, int* cap)
{
    Guard g;
    MOCK_CHECK(back == 0);
    *assign = M.assign.data(); *nm = M.nm.data(); if (np) *np = M.np; if (cap) *cap = kCap;
    return ORBX_OK;
}
int orbx_release(orbx_t*, int ticket)
{
    Guard g;
    Ticket& t = M.tk[ticket % kTickets];
    MOCK_CHECK(t.state == 2);
    t.state = 0;
    // what a reader that is still copying would now see
    std::memset(t.kps.data(), 0xEE, t.kps.size() * sizeof(OrbxKeyPoint)); std::memset(t.desc.data(), 0xEE, t.desc.size()); std::memset(t.n.data(), 0xEE, t.n.size() * 4);
    return ORBX_OK;
}
}  // extern "C"

struct Outcome { long frames = 0, searched = 0, lost = 0, batchSum = 0; };

static Outcome run(int cameras, int nframes, int wait_us, int slowCam, int track)
{
    orbslamm::CameraHub hub;
    orbslamm::CameraHub::Config c;
    c.w = 64; c.h = 4; c.cameras = cameras; c.wait_us = wait_us; c.track = track != 0;
    MOCK_CHECK(hub.open(c) == ORBX_OK);
    std::vector<Outcome> out(cameras);
    std::vector<std::thread> th;
    for (int cam = 0; cam < cameras; cam++)
        th.emplace_back([&, cam] {
            std::mt19937 rng(1234 + cam);
            std::vector<int32_t> frame(64);
            std::vector<OrbxKeyPoint> kps(kCap); std::vector<uint8_t> desc(kCap * 32); std::vector<int32_t> assign(kCap);
            for (int s = 0; s < nframes; s++) {
                frame[0] = cam; frame[1] = s;
                if (cam == slowCam) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 3000));
                else if (rng() % 16 == 0) std::this_thread::yield();
                orbslamm::CameraHub::Result r;
                MOCK_CHECK(hub.track(cam, (const uint8_t*)frame.data(), 64, kps.data(), desc.data(), assign.data(), &r) == ORBX_OK);
                MOCK_CHECK(r.n == 10 + cam + s % 5 && r.batch >= 1 && r.batch <= cameras);
                bool rows = true;
                for (int i = 0; i < r.n; i++)
                    rows = rows && kps[i].x == (float)cam && kps[i].y == (float)s && kps[i].octave == i && desc[(size_t)i * 32 + 7] == (uint8_t)((cam * 31 + s + i) & 0xFF);
                MOCK_CHECK(rows);
                if (r.nmatches >= 0) {
                    MOCK_CHECK(track && s > 0 && r.nmatches == s * 7 + cam);
                    bool tab = true;
                    for (int t = 0; t < r.n; t++) tab = tab && assign[t] == cam * 100000 + s * 10 + t % 10;
                    MOCK_CHECK(tab);
                    out[cam].searched++;
                } else {
                    for (int t = 0; t < r.n; t++) MOCK_CHECK(assign[t] == -1);
                    if (s > 0 && track) out[cam].lost++;
                }
                if (track && s % 3 == 0) {   // SearchLocalPoints' search against the frame just sent
                    OrbmProjParams p3{3, 0.8f, 0, 100};
                    const float uvr[15] = {(float)cam, (float)s};
                    const int8_t lvl[10] = {0}; const uint8_t qd[5 * 32] = {0};
                    int nm3 = -1;
                    const int rc3 = hub.search_local_points(cam, &p3, uvr, lvl, qd, nullptr, nullptr, 5, nullptr, assign.data(), &nm3);
                    if (rc3 == ORBX_OK) {
                        MOCK_CHECK(nm3 == cam * 3 + s);
                        bool tab = true;
                        for (int t = 0; t < r.n; t++) tab = tab && assign[t] == 7000000 + cam * 100000 + s * 10 + t % 10;
                        MOCK_CHECK(tab);
                    } else MOCK_CHECK(rc3 == ORBX_OK);   // (the camera's newest frame never leaves the set)
                }
                out[cam].frames++; out[cam].batchSum += r.batch;
            }
            hub.leave(cam);
        });
    for (auto& t : th) t.join();
    hub.close();
    Outcome sum;
    for (const Outcome& o : out) { sum.frames += o.frames; sum.searched += o.searched; sum.lost += o.lost; sum.batchSum += o.batchSum; }
    MOCK_CHECK(sum.frames == (long)cameras * nframes && M.frames == sum.frames);
    return sum;
}

int main(int argc, char** argv)
{
    const int nframes = argc > 1 ? std::atoi(argv[1]) : 400;
    // free-running cameras with a generous wait: lockstep, nobody loses its previous frame
    for (int cams : {1, 2, 3, 8}) {
        const Outcome o = run(cams, nframes, 20000, -1, 1);
        MOCK_CHECK(o.lost == 0 && o.searched == (long)cams * (nframes - 1));
        std::printf("cameras %d wait 20 ms: %ld frames in %ld batches (mean %.2f), %ld searched, %ld lost\n", cams, o.frames, M.batches, (double)o.batchSum / o.frames, o.searched, o.lost);
    }
    // no wait at all / one slow camera (it sleeps up to 3 ms before a frame: dozens of the others' batches pass): batches are
    // whatever happens to wait, and NOBODY loses its previous frame -- a batch never takes a slot that holds a camera's newest
    // frame (round 4's ring of 4 x cameras slots lost the slow camera's whenever it went round in between)
    for (int wait : {0, 50}) {
        const Outcome o = run(6, nframes, wait, 2, 1);
        MOCK_CHECK(o.lost == 0 && o.searched == 6L * (nframes - 1));
        std::printf("cameras 6 wait %d us, camera 2 slow: %ld frames in %ld batches (mean %.2f), %ld searched, %ld lost\n", wait, o.frames, M.batches, (double)o.batchSum / o.frames, o.searched, o.lost);
    }
    {   // extraction only
        const Outcome o = run(4, nframes, 100, -1, 0);
        MOCK_CHECK(o.searched == 0);
        std::printf("cameras 4, extraction only: %ld frames in %ld batches\n", o.frames, M.batches);
    }
    if (g_fail.load()) { std::printf("hub_mock: %d failures\n", g_fail.load()); return 1; }
    std::printf("hub_mock ok\n");
    return 0;
}
