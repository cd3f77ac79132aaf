// GPU run of the two C++ adapters against the C oracle (linked in as the checker):
// extractor (flat operator()) + ORBmatcher::SearchByBoW / SearchForInitialization.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>

#include "ORBextractor_hip.hpp"
#include "ORBmatcher_hip.hpp"
#include "../../oracle/orb_oracle.h"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

static bool l0_is_frame(iORB_SLAM::ORBextractor& ex, const std::vector<uint8_t>& img, int W, int H)
{
    iORB_SLAM::PyramidLevel& P = ex.mvImagePyramid[0];
    for (int y = 0; y < H; y++) if (std::memcmp(P.data + (size_t)y * P.step, &img[(size_t)y * W], (size_t)W)) return false;
    return true;
}

int main()
{
    const int W = 640, H = 480;
    std::mt19937 rng(7);
    std::vector<uint8_t> img((size_t)W * H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) img[(size_t)y * W + x] = (uint8_t)((((x / 13) ^ (y / 11)) * 37 + (rng() & 7)) & 0xFF);
    iORB_SLAM::ORBextractor ex(1000, 1.2f, 8, 20, 7, W, H, 0);
    std::vector<OrbxKeyPoint> kps; std::vector<uint8_t> desc;
    ex(img.data(), W, H, W, kps, desc);
    OrcExtractor oex;
    orc_extractor_init(&oex, 1000, 1.2f, 8, 20, 7);
    std::vector<OrcKeyPoint> okps(2000); std::vector<uint8_t> odesc(2000 * 32);
    const int on = orc_extract(&oex, img.data(), W, H, W, okps.data(), odesc.data(), 2000, nullptr, nullptr, nullptr);
    EXPECT(on > 100 && (size_t)on == kps.size());
    EXPECT(std::memcmp(okps.data(), kps.data(), (size_t)on * 28) == 0);
    EXPECT(std::memcmp(odesc.data(), desc.data(), (size_t)on * 32) == 0);

    // SearchByBoW on a std::map-shaped FeatureVector
    const int n = (int)kps.size();
    std::map<unsigned, std::vector<unsigned> > fvq, fvt;
    std::vector<uint8_t> d2 = desc;
    for (int i = 0; i < n; i++) { for (int b = 0; b < 6; b++) d2[(size_t)i * 32 + (rng() % 32)] ^= (uint8_t)(1u << (rng() % 8)); }
    for (int i = 0; i < n; i++) { fvq[(unsigned)(i % 17) * 3].push_back(i); fvt[(unsigned)(i % 17) * 3].push_back(n - 1 - i < 0 ? 0 : i); }
    std::vector<float> ang(n);
    for (int i = 0; i < n; i++) ang[i] = kps[i].angle;
    iORB_SLAM::FlatFeatVec fq = iORB_SLAM::FlatFeatVec::from(fvq), ft = iORB_SLAM::FlatFeatVec::from(fvt);
    iORB_SLAM::FlatMatcher m(0.75f, true, 0);
    std::vector<int32_t> match;
    const int nm = m.SearchByBoW(desc.data(), ang.data(), nullptr, n, fq, d2.data(), ang.data(), nullptr, n, ft, true, match);
    std::vector<int32_t> omatch(n);
    OrcFeatVec oq = {fq.view().n_nodes, fq.node_id.data(), fq.start.data(), fq.idx.data()};
    OrcFeatVec ot = {ft.view().n_nodes, ft.node_id.data(), ft.start.data(), ft.idx.data()};
    const int onm = orc_search_by_bow(desc.data(), ang.data(), nullptr, n, &oq, d2.data(), ang.data(), nullptr, n, &ot, 0.75f, 1, 1, omatch.data());
    EXPECT(nm == onm && nm > 100);
    EXPECT(std::memcmp(match.data(), omatch.data(), (size_t)n * 4) == 0);
    EXPECT(m.DescriptorDistance(desc.data(), d2.data()) == orc_descriptor_distance(desc.data(), d2.data()));

    // SearchForInitialization against the same frame shifted in descriptor space
    OrbmGrid g = {0.f, 0.f, 64.f / 640.f, 48.f / 480.f, 64, 48};
    OrcGridParams og = {0.f, 0.f, 64.f / 640.f, 48.f / 480.f, 64, 48};
    std::vector<float> xy((size_t)n * 2);
    for (int i = 0; i < n; i++) { xy[2 * i] = kps[i].x; xy[2 * i + 1] = kps[i].y; }
    std::vector<int> m12;
    const int ni = m.SearchForInitialization(xy.data(), 100, kps.data(), desc.data(), n, g, kps.data(), d2.data(), n, m12);
    std::vector<int32_t> cs(64 * 48 + 1), ci(n), om12(n);
    orc_grid_build(&og, (const OrcKeyPoint*)kps.data(), n, cs.data(), ci.data());
    const int oni = orc_search_for_initialization(xy.data(), 100.f, (const OrcKeyPoint*)kps.data(), desc.data(), n, &og,
                                                  (const OrcKeyPoint*)kps.data(), cs.data(), ci.data(), d2.data(), n, 0.75f, 1, om12.data());
    EXPECT(ni == oni && ni > 50);
    EXPECT(std::memcmp(m12.data(), om12.data(), (size_t)n * 4) == 0);
    // device-resident frames: the frame just extracted and a second, noisier one never leave HBM
    {
        const OrbxKeyPoint* dk = nullptr; const uint8_t* dd = nullptr;
        ex.lastOnDevice(dk, dd);
        const float K[4] = {517.3f, 516.5f, 318.6f, 255.3f}, D0[5] = {0, 0, 0, 0, 0};
        auto F1 = m.makeFrame(dk, dd, n, K, D0, g);
        std::vector<uint8_t> img2 = img;
        for (size_t i = 0; i < img2.size(); i += 3) img2[i] = (uint8_t)(img2[i] ^ (rng() & 3));
        std::vector<OrbxKeyPoint> kps2; std::vector<uint8_t> desc2;
        ex(img2.data(), W, H, W, kps2, desc2);
        ex.lastOnDevice(dk, dd);
        const int n2 = (int)kps2.size();
        auto F2 = m.makeFrame(dk, dd, n2, K, D0, g);
        std::vector<OrbxKeyPoint> un1;
        F1->keysUn(un1);
        EXPECT((int)un1.size() == n && std::memcmp(un1.data(), kps.data(), (size_t)n * 28) == 0);  // D = 0: mvKeysUn = mvKeys
        std::vector<int> f12;
        const int nf = m.SearchForInitialization(xy.data(), 100, *F1, *F2, f12);
        std::vector<int32_t> cs2(64 * 48 + 1), ci2(n2), of12(n);
        orc_grid_build(&og, (const OrcKeyPoint*)kps2.data(), n2, cs2.data(), ci2.data());
        const int onf = orc_search_for_initialization(xy.data(), 100.f, (const OrcKeyPoint*)kps.data(), desc.data(), n, &og,
                                                      (const OrcKeyPoint*)kps2.data(), cs2.data(), ci2.data(), desc2.data(), n2, 0.75f, 1, of12.data());
        EXPECT(nf == onf && nf > 50);
        EXPECT(std::memcmp(f12.data(), of12.data(), (size_t)n * 4) == 0);
        std::printf("device frames: %d / %d features, %d init matches\n", n, n2, nf);
    }
    // Ownership: matchers are stack temporaries; the device state is the calling thread's.
    //  * constructing / destroying a matcher makes no handle and no allocation once the thread has searched;
    //  * four threads building temporaries side by side get the oracle's answer each (their handles share the chain streams);
    //  * a device frame made in one thread is searched through another thread's handle and destroyed there, after its
    //    creator's thread -- and with it the creator's thread-local handle -- has ended;
    //  * frame blocks are recycled: a frame per image allocates nothing in steady state.
    {
        orbm_t* mine = nullptr;
        EXPECT(orbm_thread_handle(0, &mine) == ORBX_OK && mine == m.handle());
        int64_t a0 = 0, b0 = 0, c0 = 0, a1 = 0, b1 = 0, c1 = 0;
        std::vector<int32_t> mt;
        { iORB_SLAM::FlatMatcher w(0.75f, true, 0); w.SearchByBoW(desc.data(), ang.data(), nullptr, n, fq, d2.data(), ang.data(), nullptr, n, ft, true, mt); }
        orbm_alloc_stats(mine, &a0, &b0, &c0);
        for (int rep = 0; rep < 20; rep++) {
            iORB_SLAM::FlatMatcher tmp(0.75f, true, 0);   // the reference's `ORBmatcher matcher(0.75, true);` on the stack
            const int k = tmp.SearchByBoW(desc.data(), ang.data(), nullptr, n, fq, d2.data(), ang.data(), nullptr, n, ft, true, mt);
            EXPECT(k == onm && std::memcmp(mt.data(), omatch.data(), (size_t)n * 4) == 0);
        }
        orbm_alloc_stats(mine, &a1, &b1, &c1);
        EXPECT(a0 == a1 && b0 == b1 && c0 == c1);
        // frames: the first few take blocks, later ones recycle them
        const OrbxKeyPoint* dk = nullptr; const uint8_t* dd = nullptr;
        ex(img.data(), W, H, W, kps, desc);
        ex.lastOnDevice(dk, dd);
        const float K[4] = {517.3f, 516.5f, 318.6f, 255.3f}, D0[5] = {0, 0, 0, 0, 0};
        { auto Fa = m.makeFrame(dk, dd, n, K, D0, g); auto Fb = m.makeFrame(dk, dd, n, K, D0, g); }
        orbm_alloc_stats(mine, &a0, &b0, &c0);
        for (int rep = 0; rep < 50; rep++) {
            auto Fa = m.makeFrame(dk, dd, n, K, D0, g);
            auto Fb = m.makeFrame(dk, dd, n - (rep % 7), K, D0, g);
            if (rep % 10 == 0) {
                std::vector<int> f12;
                const int nf = m.SearchForInitialization(xy.data(), 100, *Fa, *Fa, f12);
                EXPECT(nf > 50);
            }
        }
        orbm_alloc_stats(mine, &a1, &b1, &c1);
        EXPECT(a0 == a1 && b0 == b1 && c0 == c1);
        // threads: each its own handle, the same answers; a frame handed from a thread that ends to one that goes on
        std::unique_ptr<iORB_SLAM::DeviceFrame> handed;
        int64_t madeBefore = 0, madeAfter = 0;
        orbm_alloc_stats(nullptr, nullptr, nullptr, &madeBefore);
        std::vector<int> tfails(4, 0);
        std::vector<std::thread> th;
        for (int t = 0; t < 4; t++)
            th.emplace_back([&, t] {
                std::vector<int32_t> mm;
                for (int rep = 0; rep < 10; rep++) {
                    iORB_SLAM::FlatMatcher tmp(0.75f, true, 0);
                    const int k = tmp.SearchByBoW(desc.data(), ang.data(), nullptr, n, fq, d2.data(), ang.data(), nullptr, n, ft, true, mm);
                    if (k != onm || std::memcmp(mm.data(), omatch.data(), (size_t)n * 4)) tfails[(size_t)t]++;
                }
                if (t == 0) { iORB_SLAM::FlatMatcher mk(0.75f, true, 0); handed = mk.makeFrame(dk, dd, n, K, D0, g); }
            });
        for (auto& t : th) t.join();
        orbm_alloc_stats(nullptr, nullptr, nullptr, &madeAfter);
        EXPECT(madeAfter - madeBefore == 4);   // one handle per thread, not one per matcher object
        for (int t = 0; t < 4; t++) EXPECT(tfails[(size_t)t] == 0);
        EXPECT(handed && handed->size() == n);
        {
            std::vector<int> f12;
            auto Fm = m.makeFrame(dk, dd, n, K, D0, g);
            const int nf = m.SearchForInitialization(xy.data(), 100, *handed, *Fm, f12);   // thread 0's frame through this thread's handle
            std::vector<int> g12;
            const int ng = m.SearchForInitialization(xy.data(), 100, *Fm, *Fm, g12);
            EXPECT(nf == ng && f12 == g12);
            std::vector<OrbxKeyPoint> un;
            handed->keysUn(un);
            EXPECT((int)un.size() == n && std::memcmp(un.data(), kps.data(), (size_t)n * 28) == 0);
        }
        handed.reset();   // the last reference to thread 0's handle goes here
        std::printf("ownership: 20 + 40 matcher temporaries, 100 recycled frames, no handle or allocation made in steady state\n");
    }
    // FrameSet: two consecutive extractions into two slots, SearchByProjection(Cur, Last) with the identity pose on the
    // device, against the oracle's sequential loop on the downloaded arrays
    {
        const float K[4] = {517.3f, 516.5f, 318.6f, 255.3f}, D0[5] = {0, 0, 0, 0, 0}, bounds[4] = {0.f, (float)W, 0.f, (float)H};
        iORB_SLAM::FrameSet fs(m.handle(), 2, orbx_max_keypoints(ex.handle()), K, D0, g, bounds, ex.GetScaleFactors());
        std::vector<OrbxKeyPoint> ka, kb; std::vector<uint8_t> da, db;
        ex(img.data(), W, H, W, ka, da);
        fs.build(0, ex.handle());
        std::vector<uint8_t> img2 = img;
        for (size_t i = 1; i < img2.size(); i += 5) img2[i] = (uint8_t)(img2[i] ^ (rng() & 3));
        ex(img2.data(), W, H, W, kb, db);
        fs.build(1, ex.handle());
        std::vector<OrbxKeyPoint> un; std::vector<uint8_t> dd2;
        EXPECT(fs.download(1, un, dd2) == (int)kb.size() && std::memcmp(un.data(), kb.data(), kb.size() * 28) == 0 && dd2 == db);
        fs.track({1}, {0}, 15.f);
        const int32_t* assign = nullptr; const int32_t* nmv = nullptr;
        EXPECT(fs.results(assign, nmv) == 1);
        const int na = (int)ka.size(), nb = (int)kb.size();
        std::vector<float> uvr((size_t)na * 3); std::vector<int8_t> lvl((size_t)na * 2); std::vector<float> ang((size_t)na);
        const std::vector<float> sf = ex.GetScaleFactors();
        for (int i = 0; i < na; i++) {
            uvr[3 * i] = ka[i].x; uvr[3 * i + 1] = ka[i].y; uvr[3 * i + 2] = 15.f * sf[(size_t)ka[i].octave];
            lvl[2 * i] = (int8_t)(ka[i].octave - 1); lvl[2 * i + 1] = (int8_t)(ka[i].octave + 1);
            ang[i] = ka[i].angle;
        }
        std::vector<int32_t> cs3(64 * 48 + 1), ci3(nb), want(nb, -1);
        std::vector<uint8_t> occ(nb, 0);
        orc_grid_build(&og, (const OrcKeyPoint*)kb.data(), nb, cs3.data(), ci3.data());
        OrcProjParams opp = {4, 0.9f, 1, 100};
        const int wn = orc_search_by_projection(&opp, uvr.data(), lvl.data(), da.data(), ang.data(), nullptr, nullptr, na, &og,
                                                (const OrcKeyPoint*)kb.data(), cs3.data(), ci3.data(), db.data(), nb, occ.data(), want.data());
        EXPECT(nmv[0] == wn && wn > 100);
        EXPECT(std::memcmp(assign, want.data(), (size_t)nb * 4) == 0);
        std::printf("frame set: %d / %d features, %d projection matches\n", na, nb, wn);
    }
    // mvImagePyramid (ORBextractor.h:85): a public member again, filled on first use after operator(), each level inside
    // a 19 px BORDER_REFLECT_101 frame like ComputePyramid leaves it (ORBextractor.cc:1107-1132)
    {
        std::vector<OrbxKeyPoint> k3; std::vector<uint8_t> d3;
        ex(img.data(), W, H, W, k3, d3);
        EXPECT(ex.mvImagePyramid.size() == 8);
        std::vector<uint8_t> opyr((size_t)W * H * 4);
        orc_extract(&oex, img.data(), W, H, W, okps.data(), odesc.data(), 2000, opyr.data(), nullptr, nullptr);
        size_t off = 0;
        for (int l = 0; l < 8; l++) {
            iORB_SLAM::PyramidLevel& P = ex.mvImagePyramid[(size_t)l];
            int lw = 0, lh = 0;
            orc_level_size(&oex, W, H, l, &lw, &lh);
            EXPECT(P.cols == lw && P.rows == lh && P.step == (size_t)lw + 38);
            bool same = true, border = true;
            for (int y = 0; y < lh && same; y++) same = std::memcmp(P.data + (size_t)y * P.step, &opyr[off + (size_t)y * lw], (size_t)lw) == 0;
            for (int d = 1; d <= 19; d++) {   // reflect-101: pixel -d mirrors pixel +d, pixel n-1+d mirrors n-1-d
                border = border && P.data[-d] == P.data[d] && P.data[(ptrdiff_t)(lh - 1 + d) * (ptrdiff_t)P.step + 5] == P.data[(size_t)(lh - 1 - d) * P.step + 5];
                border = border && P.data[-(ptrdiff_t)d * (ptrdiff_t)P.step - d] == P.data[(size_t)d * P.step + d];
            }
            EXPECT(same); EXPECT(border);
            off += (size_t)lw * lh;
        }
        EXPECT(l0_is_frame(ex, img, W, H));
    }
    // three robots' extractors on ONE hub (MultipleRobotsScenario: an ORBextractor per robot, a tracking thread each): every
    // robot gets the keypoints and descriptors of ITS frames, whatever went through the GPU beside them
    {
        orbslamm::CameraHub::Config hc;
        hc.prm = OrbxParams{1000, 1.2f, 8, 20, 7}; hc.w = W; hc.h = H; hc.cameras = 3; hc.device = 0; hc.track = false; hc.wait_us = 500;
        std::shared_ptr<orbslamm::CameraHub> hub(new orbslamm::CameraHub());
        EXPECT(hub->open(hc) == ORBX_OK);
        const int R = 3, F = 4;
        std::vector<std::vector<uint8_t> > frames((size_t)R * F, std::vector<uint8_t>((size_t)W * H));
        for (int r = 0; r < R; r++)
            for (int f = 0; f < F; f++)
                for (int y = 0; y < H; y++)
                    for (int x = 0; x < W; x++) frames[(size_t)r * F + f][(size_t)y * W + x] = (uint8_t)(((((x + 5 * f) / (11 + r)) ^ ((y + 3 * r) / (9 + f))) * 41 + (rng() & 7)) & 0xFF);
        std::vector<std::vector<OrbxKeyPoint> > gk((size_t)R * F); std::vector<std::vector<uint8_t> > gd((size_t)R * F);
        std::vector<std::thread> th;
        for (int r = 0; r < R; r++)
            th.emplace_back([&, r] {
                iORB_SLAM::ORBextractor rex(hub, r);
                for (int f = 0; f < F; f++) rex(frames[(size_t)r * F + f].data(), W, H, W, gk[(size_t)r * F + f], gd[(size_t)r * F + f]);
            });
        for (auto& t : th) t.join();
        for (int i = 0; i < R * F; i++) {
            const int n2 = orc_extract(&oex, frames[(size_t)i].data(), W, H, W, okps.data(), odesc.data(), 2000, nullptr, nullptr, nullptr);
            EXPECT(n2 > 100 && (size_t)n2 == gk[(size_t)i].size());
            EXPECT((size_t)n2 == gk[(size_t)i].size() && std::memcmp(okps.data(), gk[(size_t)i].data(), (size_t)n2 * 28) == 0);
            EXPECT((size_t)n2 * 32 == gd[(size_t)i].size() && std::memcmp(odesc.data(), gd[(size_t)i].data(), (size_t)n2 * 32) == 0);
        }
    }
    std::printf(fails ? "adapter_gpu: %d failures\n" : "adapter_gpu ok (%d keypoints, %d BoW matches, %d init matches)\n", fails ? fails : on, nm, ni);
    return fails ? 1 : 0;
}
