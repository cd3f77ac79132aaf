// tracking_loop -- the drop-in CLASSES on the clock, used the way the reference's Tracking thread uses them.
//
// The reference builds its matchers as stack temporaries at every call site; per tracked frame
// (/root/reference/SingleRobotScenario/src/Tracking.cc):
//     Frame::Frame -> (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors)                       Frame.cc:175-210, 247-253
//     TrackWithMotionModel:   { ORBmatcher matcher(0.9,true); matcher.SearchByProjection(Cur, Last, 15, true); }   :905-936
//     TrackReferenceKeyFrame: { Cur.ComputeBoW(); ORBmatcher matcher(0.7,true); matcher.SearchByBoW(pKF, Cur, v); } :800-812
//     SearchLocalPoints:      { ORBmatcher matcher(0.8); matcher.SearchByProjection(Cur, mvpLocalMapPoints, th); } :1242-1249
// and Examples/Monocular/mono_tum.cc:80-122 times the whole call per image and prints the median and the mean.
//
// This program is that loop on include/ORBextractor_hip.hpp + include/ORBmatcher_hip.hpp (the typedef INTEGRATION.md asks the
// maintainer for), over tests/cpp/mock_slam.hpp's stand-ins for Frame / KeyFrame / MapPoint (OpenCV and the reference's map
// are not in this image), at KITTI size: 1241 x 376, 2000 features, ~1600 MapPoints in LastFrame, ~3000 local MapPoints.
// Per member it reports   total (what Tracking waits for)  =  device (inside the C ABI)  +  adapter (the object-graph walk:
// the reference's own loop head and write-back, on the host),   the same arrays through the raw C ABI, and the round-4
// pattern (a device handle created and destroyed per matcher object) -- and it FAILS if the steady-state loop made a device
// handle, a stream, or a device / pinned allocation (orbm_alloc_stats; the rocprofv3 HIP-API trace of this binary is the
// outside evidence: profiles/r05_tracking_loop_hip_api.txt).
//
// usage: tracking_loop [--frames 300] [--warmup 30] [--w 1241 --h 376 --features 2000] [--device 0] [--voc-levels 6]
//                      [--no-old-pattern] [--json]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ORBextractor_hip.hpp"
#include "ORBmatcher_hip.hpp"
#include "mock_slam.hpp"

using namespace mock;
typedef iORB_SLAM::ORBmatcherT<Frame, KeyFrame, MapPoint> ORBmatcher;
typedef std::chrono::steady_clock Clock;
static double us_since(Clock::time_point t0) { return std::chrono::duration<double, std::micro>(Clock::now() - t0).count(); }

struct Args {
    int w = 1241, h = 376, features = 2000, frames = 300, warmup = 30, device = 0, vocLevels = 6;
    bool oldPattern = true, json = false;
};

// ---------------------------------------------------------------- synthetic camera (the stream bench.py and examples/multi_robot use)
struct SplitMix { uint64_t s; uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
                  int below(int n) { return (int)(next() % (uint64_t)n); } double unit() { return (double)(next() >> 11) / 9007199254740992.0; } };
static int tri(int a, int m) { a %= 2 * m; return a <= m ? a : 2 * m - a; }

static std::vector<uint8_t> make_scene(int w, int h, int& cw, int& ch)
{
    cw = w + 64; ch = h + 16;
    std::vector<uint8_t> c((size_t)cw * ch, 128);
    SplitMix r{0x0B5A4000ull};
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
static void view_offset(int t, int& ox, int& oy) { ox = tri(2 * t, 64); oy = tri(t, 16); }
static void make_frame(const std::vector<uint8_t>& scene, int cw, int w, int h, int t, uint8_t* dst, int stride)
{
    int ox, oy; view_offset(t, ox, oy);
    SplitMix r{0x5EED0000ull * 100003ull + (uint64_t)t};
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

// ---------------------------------------------------------------- the world behind the camera
// The scene is a fronto-parallel plane at depth Z; the camera translates in its own image plane, so that a canvas position
// (X, Y) is a world point and the pan of frame t is its pose: exact geometry, no approximation in the projections.
static const float FX = 718.856f, FY = 718.856f, CX = 607.1928f, CY = 185.2157f, DEPTH = 12.f;   // KITTI 00-02 intrinsics
static void pose_of(int t, Mat& Tcw)
{
    int ox, oy; view_offset(t, ox, oy);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tcw.at<float>(r, c) = r == c ? 1.f : 0.f;
    Tcw.at<float>(0, 3) = -(float)ox / FX * DEPTH;
    Tcw.at<float>(1, 3) = -(float)oy / FY * DEPTH;
}

struct Vocabulary {   // a synthetic k-ary tree in loadFromTextFile order (the real ORBvoc.txt is absent from the reference checkout)
    orbv_t* v = nullptr;
    ~Vocabulary() { if (v) orbv_destroy(v); }
    void build(int device, int k, int L)
    {
        std::vector<int32_t> parent; std::vector<uint8_t> leaf, desc; std::vector<double> weight;
        SplitMix r{7};
        std::vector<int> prevIds(1, 0);
        std::vector<uint8_t> prevDesc(32);
        for (auto& b : prevDesc) b = (uint8_t)r.below(256);
        int nextId = 1;
        for (int lvl = 1; lvl <= L; lvl++) {
            std::vector<int> ids; std::vector<uint8_t> descs;
            const int nflip = std::max(4, 60 >> (lvl - 1));
            for (size_t p = 0; p < prevIds.size(); p++)
                for (int c = 0; c < k; c++) {
                    uint8_t d[32];
                    std::memcpy(d, &prevDesc[p * 32], 32);
                    for (int f = 0; f < nflip; f++) { const int bit = r.below(256); d[bit >> 3] ^= (uint8_t)(1u << (bit & 7)); }
                    parent.push_back(prevIds[p]); leaf.push_back(lvl == L);
                    desc.insert(desc.end(), d, d + 32);
                    weight.push_back(lvl == L ? 0.5 + 4.0 * r.unit() : 0.0);
                    ids.push_back(nextId++); descs.insert(descs.end(), d, d + 32);
                }
            prevIds.swap(ids); prevDesc.swap(descs);
        }
        if (orbv_create(device, k, L, /*scoring L1_NORM*/ 0, /*weighting TF_IDF*/ 0, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), &v) != ORBX_OK)
            throw std::runtime_error(std::string("vocabulary: ") + orbx_last_error());
    }
    // Frame::ComputeBoW (Frame.cc:394-402): mFeatVec as the std::map DBoW2 fills (the BowVector is not read by the matchers)
    void computeBoW(const Mat& descriptors, int n, FeatureVector& fv, std::vector<uint32_t>& wid, std::vector<double>& wval,
                    std::vector<uint32_t>& node, std::vector<int32_t>& start, std::vector<int32_t>& idx)
    {
        wid.resize((size_t)n); wval.resize((size_t)n); node.resize((size_t)n); start.resize((size_t)n + 1); idx.resize((size_t)n);
        int nw = 0, nn = 0;
        if (orbv_transform(v, descriptors.ptr<uint8_t>(0), n, 4, wid.data(), wval.data(), &nw, node.data(), start.data(), idx.data(), &nn) != ORBX_OK)
            throw std::runtime_error(std::string("ComputeBoW: ") + orbx_last_error());
        fv.clear();
        FeatureVector::iterator hint = fv.end();
        for (int i = 0; i < nn; i++) {
            hint = fv.insert(hint, std::make_pair((unsigned)node[i], std::vector<unsigned>(idx.begin() + start[i], idx.begin() + start[i + 1])));
        }
    }
};

// one ring position: what the extractor found there + the MapPoints Tracking would hold for it
struct Template {
    std::vector<KeyPoint> keys;
    Mat desc;
    FeatureVector fv;
    std::vector<std::unique_ptr<MapPoint> > owned;
    std::vector<MapPoint*> mps;          // per feature, null for ~20 %
    std::vector<float> canvasX, canvasY; // where the feature sits on the canvas (= its world point)
    Mat Tcw = Mat::f32(4, 4);
    std::unique_ptr<KeyFrame> kf;
};

struct Series { std::vector<double> v; void add(double x) { v.push_back(x); }
                double median() const { if (v.empty()) return 0; std::vector<double> s(v); std::sort(s.begin(), s.end()); return s[s.size() / 2]; }
                double mean() const { double t = 0; for (double x : v) t += x; return v.empty() ? 0 : t / (double)v.size(); } };

int main(int argc, char** argv)
{
    Args A;
    for (int i = 1; i < argc; i++) {
        const std::string k = argv[i];
        auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : "0"; };
        if (k == "--frames") A.frames = atoi(val()); else if (k == "--warmup") A.warmup = atoi(val()); else if (k == "--w") A.w = atoi(val());
        else if (k == "--h") A.h = atoi(val()); else if (k == "--features") A.features = atoi(val()); else if (k == "--device") A.device = atoi(val());
        else if (k == "--voc-levels") A.vocLevels = atoi(val()); else if (k == "--no-old-pattern") A.oldPattern = false; else if (k == "--json") A.json = true;
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 2; }
    }
    if (A.frames < 4 || A.warmup < 0 || A.vocLevels < 2 || A.vocLevels > 6) { std::fprintf(stderr, "bad arguments\n"); return 2; }
    try {
        const int nring = 16;
        int cw, ch;
        const std::vector<uint8_t> scene = make_scene(A.w, A.h, cw, ch);
        std::vector<uint8_t> ring((size_t)nring * A.w * A.h);   // pageable, like the cv::Mat cv::imread hands to TrackMonocular
        for (int t = 0; t < nring; t++) make_frame(scene, cw, A.w, A.h, t, &ring[(size_t)t * A.w * A.h], A.w);

        iORB_SLAM::ORBextractor ex(A.features, 1.2f, 8, 20, 7, A.w, A.h, A.device);
        const std::vector<float> scaleFactors = ex.GetScaleFactors();
        Frame::fx = FX; Frame::fy = FY; Frame::cx = CX; Frame::cy = CY;
        Frame::mnMinX = 0; Frame::mnMaxX = (float)A.w; Frame::mnMinY = 0; Frame::mnMaxY = (float)A.h;   // no distortion (Frame.cc:436-464)
        Frame::mfGridElementWidthInv = 64.f / (Frame::mnMaxX - Frame::mnMinX);
        Frame::mfGridElementHeightInv = 48.f / (Frame::mnMaxY - Frame::mnMinY);

        Vocabulary voc;
        voc.build(A.device, 10, A.vocLevels);
        std::vector<uint32_t> wid, node; std::vector<double> wval; std::vector<int32_t> fstart, fidx;

        // ---- templates: every ring position once through the extractor
        KeyFrame anchor;   // the KeyFrame every MapPoint is observed from (Observations() > 0)
        std::vector<Template> T((size_t)nring);
        std::vector<OrbxKeyPoint> kps; std::vector<uint8_t> desc;
        SplitMix pick{99};
        for (int r = 0; r < nring; r++) {
            Template& t = T[(size_t)r];
            ex(&ring[(size_t)r * A.w * A.h], A.w, A.h, A.w, kps, desc);
            const int n = (int)kps.size();
            t.keys.resize((size_t)n);
            std::memcpy((void*)t.keys.data(), kps.data(), (size_t)n * sizeof(KeyPoint));
            t.desc = Mat::u8(std::max(n, 1), 32);
            std::memcpy(t.desc.ptr<uint8_t>(0), desc.data(), (size_t)n * 32);
            voc.computeBoW(t.desc, n, t.fv, wid, wval, node, fstart, fidx);
            pose_of(r, t.Tcw);
            int ox, oy; view_offset(r, ox, oy);
            t.mps.assign((size_t)n, nullptr); t.canvasX.resize((size_t)n); t.canvasY.resize((size_t)n);
            for (int i = 0; i < n; i++) {
                t.canvasX[i] = t.keys[i].pt.x + (float)ox; t.canvasY[i] = t.keys[i].pt.y + (float)oy;
                if (pick.below(5) == 0) continue;
                std::unique_ptr<MapPoint> p(new MapPoint);
                p->mWorldPos.at<float>(0) = (t.canvasX[i] - CX) / FX * DEPTH;
                p->mWorldPos.at<float>(1) = (t.canvasY[i] - CY) / FY * DEPTH;
                p->mWorldPos.at<float>(2) = DEPTH;
                p->mNormalVector.at<float>(2) = -1.f;
                std::memcpy(p->mDescriptor.ptr<uint8_t>(0), t.desc.ptr<uint8_t>(i), 32);
                p->mfMaxDistance = DEPTH * scaleFactors[t.keys[i].octave]; p->mfMinDistance = p->mfMaxDistance / scaleFactors.back();
                p->mObservations[&anchor] = (size_t)i;
                p->mnTrackScaleLevel = t.keys[i].octave; p->mTrackViewCos = 1.f;
                t.mps[i] = p.get();
                t.owned.push_back(std::move(p));
            }
            t.kf.reset(new KeyFrame);
            KeyFrame& K = *t.kf;
            K.N = n; K.mvKeysUn = t.keys; K.mDescriptors = t.desc; K.mFeatVec = t.fv; K.mvpMapPoints = t.mps;
            K.fx = FX; K.fy = FY; K.cx = CX; K.cy = CY; K.mvScaleFactors = scaleFactors;
            K.mnMaxX = A.w; K.mnMaxY = A.h; K.mfGridElementWidthInv = Frame::mfGridElementWidthInv; K.mfGridElementHeightInv = Frame::mfGridElementHeightInv;
        }

        // ---- the loop
        Series sExtract, sFrameCtor, sBoWCompute, sMM, sKF, sAll;
        struct Member { const char* name; Series total, device, raw, old, walk; int64_t matches; explicit Member(const char* n) : name(n), matches(0) {} };
        Member M4("SearchByProjection(CurrentFrame,LastFrame,15,mono)"), M3("SearchByProjection(CurrentFrame,mvpLocalMapPoints,1)"),
               M1("SearchByBoW(pKF,CurrentFrame,vpMapPointMatches)");
        std::vector<MapPoint*> local, vpMapPointMatches;
        int64_t a0 = 0, b0 = 0, c0 = 0, a1 = 0, b1 = 0, c1 = 0;
        orbm_t* th = nullptr;
        uint64_t checksum = 0, walkSink = 0;
        for (int it = -A.warmup; it < A.frames; it++) {
            const bool timed = it >= 0;
            if (it == 0) { orbm_thread_handle(A.device, &th); orbm_alloc_stats(th, &a0, &b0, &c0); }
            const int t = it + A.warmup + 3, r = t % nring;
            const Template& TL = T[(size_t)((t - 1) % nring)];
            const uint8_t* im = &ring[(size_t)r * A.w * A.h];
            const Clock::time_point f0 = Clock::now();

            // Frame::Frame (Frame.cc:175-210): ExtractORB, then N, mvKeysUn, mvpMapPoints, mvbOutlier (the grid is built on the device per search)
            Clock::time_point s0 = Clock::now();
            ex(im, A.w, A.h, A.w, kps, desc);
            const double usExtract = us_since(s0);
            s0 = Clock::now();
            Frame Cur;
            Cur.N = (int)kps.size();
            Cur.mvKeys.resize(kps.size());
            std::memcpy((void*)Cur.mvKeys.data(), kps.data(), kps.size() * sizeof(KeyPoint));
            Cur.mvKeysUn = Cur.mvKeys;
            Cur.mDescriptors = Mat::u8(std::max(Cur.N, 1), 32);
            std::memcpy(Cur.mDescriptors.ptr<uint8_t>(0), desc.data(), (size_t)Cur.N * 32);
            Cur.mvpMapPoints.assign((size_t)Cur.N, static_cast<MapPoint*>(NULL));
            Cur.mvbOutlier.assign((size_t)Cur.N, false);
            Cur.mvScaleFactors = scaleFactors; Cur.mfLogScaleFactor = std::log(1.2f);
            pose_of(r, Cur.mTcw);   // mCurrentFrame.SetPose(mVelocity * mLastFrame.mTcw): the motion model's prediction
            const double usCtor = us_since(s0);

            // LastFrame as Tracking keeps it: the previous image's features with their MapPoints
            Frame Last;
            Last.N = (int)TL.keys.size(); Last.mvKeys = TL.keys; Last.mvKeysUn = TL.keys; Last.mDescriptors = TL.desc;
            Last.mvpMapPoints = TL.mps; Last.mvbOutlier.assign((size_t)Last.N, false); Last.mvScaleFactors = scaleFactors; Last.mTcw = TL.Tcw;

            // TrackWithMotionModel (Tracking.cc:905-936)
            s0 = Clock::now();
            int nm4;
            {
                ORBmatcher matcher(0.9f, true);
                std::fill(Cur.mvpMapPoints.begin(), Cur.mvpMapPoints.end(), static_cast<MapPoint*>(NULL));
                nm4 = matcher.SearchByProjection(Cur, Last, 15, true);
            }
            const double us4 = us_since(s0), dev4 = iORB_SLAM::FlatMatcher::lastDeviceUs();

            // TrackReferenceKeyFrame (Tracking.cc:800-812); its matches go to a vector of their own, Cur keeps the motion model's
            s0 = Clock::now();
            voc.computeBoW(Cur.mDescriptors, Cur.N, Cur.mFeatVec, wid, wval, node, fstart, fidx);
            const double usBow = us_since(s0);
            s0 = Clock::now();
            int nm1;
            {
                ORBmatcher matcher(0.7f, true);
                nm1 = matcher.SearchByBoW(TL.kf.get(), Cur, vpMapPointMatches);
            }
            const double us1 = us_since(s0), dev1 = iORB_SLAM::FlatMatcher::lastDeviceUs();

            // SearchLocalPoints (Tracking.cc:1196-1249): the points of the two frames before LastFrame, "in frustum" (isInFrustum,
            // Frame.cc:255-325, is the caller's: it fills mbTrackInView, mTrackProjX/Y, mnTrackScaleLevel, mTrackViewCos)
            local.clear();
            int ox, oy; view_offset(r, ox, oy);
            for (int back = 2; back <= 3; back++) {
                const Template& TB = T[(size_t)((t - back) % nring)];
                for (size_t i = 0; i < TB.mps.size(); i++) {
                    MapPoint* p = TB.mps[i];
                    if (!p) continue;
                    const float u = TB.canvasX[i] - (float)ox, v = TB.canvasY[i] - (float)oy;
                    p->mbTrackInView = u >= 0 && u < (float)A.w && v >= 0 && v < (float)A.h;
                    p->mTrackProjX = u; p->mTrackProjY = v;
                    local.push_back(p);
                }
            }
            s0 = Clock::now();
            int nm3;
            {
                ORBmatcher matcher(0.8f);
                nm3 = matcher.SearchByProjection(Cur, local, 1);
            }
            const double us3 = us_since(s0), dev3 = iORB_SLAM::FlatMatcher::lastDeviceUs();
            const double usFrame = us_since(f0);
            checksum = checksum * 1000003ull + (uint64_t)(nm4 * 7 + nm1 * 13 + nm3 * 31 + Cur.N);
            if (!timed) continue;
            sExtract.add(usExtract); sFrameCtor.add(usCtor); sBoWCompute.add(usBow);
            M4.total.add(us4); M4.device.add(dev4); M4.matches += nm4;
            M1.total.add(us1); M1.device.add(dev1); M1.matches += nm1;
            M3.total.add(us3); M3.device.add(dev3); M3.matches += nm3;
            sMM.add(usExtract + usCtor + us4 + us3);
            sKF.add(usExtract + usCtor + usBow + us1 + us3);
            sAll.add(usFrame);
        }
        orbm_alloc_stats(th, &a1, &b1, &c1);
        const bool steady = a0 == a1 && b0 == b1 && c0 == c1;

        // ---- the same arrays through the raw C ABI (what the adapter adds), and through the round-4 ownership pattern
        // (orbm_create + search + orbm_destroy per matcher object: a stream, cold scratch and pinned block, device-wide frees)
        {
            const int reps = std::min(A.frames, 60);
            for (int it = 0; it < reps; it++) {
                const int t = it + 5, r = t % nring;
                const Template& TL = T[(size_t)((t - 1) % nring)];
                ex(&ring[(size_t)r * A.w * A.h], A.w, A.h, A.w, kps, desc);
                Frame Cur, Last;
                Cur.N = (int)kps.size(); Cur.mvKeys.resize(kps.size());
                std::memcpy((void*)Cur.mvKeys.data(), kps.data(), kps.size() * sizeof(KeyPoint));
                Cur.mvKeysUn = Cur.mvKeys; Cur.mDescriptors = Mat::u8(std::max(Cur.N, 1), 32);
                std::memcpy(Cur.mDescriptors.ptr<uint8_t>(0), desc.data(), (size_t)Cur.N * 32);
                Cur.mvpMapPoints.assign((size_t)Cur.N, static_cast<MapPoint*>(NULL)); Cur.mvbOutlier.assign((size_t)Cur.N, false);
                Cur.mvScaleFactors = scaleFactors; pose_of(r, Cur.mTcw);
                Last.N = (int)TL.keys.size(); Last.mvKeys = TL.keys; Last.mvKeysUn = TL.keys; Last.mDescriptors = TL.desc;
                Last.mvpMapPoints = TL.mps; Last.mvbOutlier.assign((size_t)Last.N, false); Last.mvScaleFactors = scaleFactors; Last.mTcw = TL.Tcw;
                voc.computeBoW(Cur.mDescriptors, Cur.N, Cur.mFeatVec, wid, wval, node, fstart, fidx);
                auto raw_proj = [&](ORBmatcher& m, Member& mem, int expect) {
                    ORBmatcher::FlatCall& c = m.last();
                    OrbmProjParams pp = {c.mode, m.flat().mfNNratio, m.flat().mbCheckOrientation ? 1 : 0, c.thDist};
                    std::vector<uint8_t> occ; std::vector<int32_t> assign;
                    for (int variant = 0; variant < (A.oldPattern ? 2 : 1); variant++) {
                        occ = c.tocc_in; assign.assign((size_t)c.nt, -2);
                        int n = 0;
                        const Clock::time_point s0 = Clock::now();
                        orbm_t* h = th;
                        if (variant == 1 && orbm_create(A.device, &h) != ORBX_OK) throw std::runtime_error(orbx_last_error());
                        const int rc = orbm_search_by_projection(h, &pp, c.q_uvr.data(), c.q_lvl.data(), c.qdesc.data(), c.qangle.data(), nullptr, c.qobs.data(), c.nq,
                                                                 &c.grid, c.tkeys, c.tdesc, c.nt, occ.data(), assign.data(), &n);
                        if (variant == 1) orbm_destroy(h);
                        const double us = us_since(s0);
                        if (rc != ORBX_OK || n != expect) throw std::runtime_error(std::string("raw C ABI call disagrees with the member: ") + orbx_last_error());
                        (variant ? mem.old : mem.raw).add(us);
                    }
                };
                {
                    ORBmatcher m(0.9f, true);
                    const int n = m.SearchByProjection(Cur, Last, 15, true);
                    raw_proj(m, M4, n);
                    // the getters the reference's own loop calls per MapPoint before it searches (ORBmatcher.cc:1355-1392: GetWorldPos,
                    // GetDescriptor -- clones taken under the point's mutex): what ANY implementation of the member pays on the host
                    const Clock::time_point s0 = Clock::now();
                    for (int i = 0; i < Last.N; i++) {
                        MapPoint* p = Last.mvpMapPoints[i];
                        if (!p || Last.mvbOutlier[i]) continue;
                        const Mat x = p->GetWorldPos(); const Mat d = p->GetDescriptor();
                        walkSink += (uint64_t)x.at<float>(2) + d.ptr<uint8_t>(0)[3] + (uint64_t)p->Observations();
                    }
                    M4.walk.add(us_since(s0));
                }
                {
                    ORBmatcher m(0.7f, true);
                    const int n = m.SearchByBoW(TL.kf.get(), Cur, vpMapPointMatches);
                    ORBmatcher::FlatCall& c = m.last();
                    OrbmFeatVec qa = c.qfv.view(), ta = c.tfv.view();
                    std::vector<int32_t> match((size_t)c.nt);
                    for (int variant = 0; variant < (A.oldPattern ? 2 : 1); variant++) {
                        int k = 0;
                        const Clock::time_point s0 = Clock::now();
                        orbm_t* h = th;
                        if (variant == 1 && orbm_create(A.device, &h) != ORBX_OK) throw std::runtime_error(orbx_last_error());
                        const int rc = orbm_search_by_bow(h, c.qdescBlock, c.qangle.data(), c.qvalid.data(), c.nq, &qa, c.tdesc, c.tangle.data(), nullptr, c.nt, &ta,
                                                          0.7f, 1, 1, match.data(), &k);
                        if (variant == 1) orbm_destroy(h);
                        const double us = us_since(s0);
                        if (rc != ORBX_OK || k != n) throw std::runtime_error(std::string("raw C ABI call disagrees with the member: ") + orbx_last_error());
                        (variant ? M1.old : M1.raw).add(us);
                    }
                }
                {
                    local.clear();
                    int ox, oy; view_offset(r, ox, oy);
                    for (int back = 2; back <= 3; back++) {
                        const Template& TB = T[(size_t)((t - back) % nring)];
                        for (size_t i = 0; i < TB.mps.size(); i++) {
                            MapPoint* p = TB.mps[i];
                            if (!p) continue;
                            const float u = TB.canvasX[i] - (float)ox, v = TB.canvasY[i] - (float)oy;
                            p->mbTrackInView = u >= 0 && u < (float)A.w && v >= 0 && v < (float)A.h;
                            p->mTrackProjX = u; p->mTrackProjY = v;
                            local.push_back(p);
                        }
                    }
                    ORBmatcher m(0.8f);
                    const int n = m.SearchByProjection(Cur, local, 1);
                    raw_proj(m, M3, n);
                    const Clock::time_point s0 = Clock::now();   // ORBmatcher.cc:53-76: mbTrackInView, isBad, GetDescriptor per point
                    for (size_t i = 0; i < local.size(); i++) {
                        MapPoint* p = local[i];
                        if (!p->mbTrackInView || p->isBad()) continue;
                        const Mat d = p->GetDescriptor();
                        walkSink += d.ptr<uint8_t>(0)[3] + (uint64_t)p->Observations();
                    }
                    M3.walk.add(us_since(s0));
                }
            }
        }

        // ---- report (mono_tum.cc:113-122 prints the first two lines)
        const int F = A.frames;
        std::printf("-------\n\nmedian tracking time: %.6f\nmean tracking time: %.6f\n", sAll.median() * 1e-6, sAll.mean() * 1e-6);
        std::printf("(front-end only, %d frames %dx%d, %d features; every member of the list below runs on every frame)\n", F, A.w, A.h, A.features);
        std::printf("  ORBextractor::operator()           median %8.1f us  mean %8.1f us\n", sExtract.median(), sExtract.mean());
        std::printf("  Frame members (host copies)        median %8.1f us\n  ComputeBoW (orbv_transform + map)  median %8.1f us\n", sFrameCtor.median(), sBoWCompute.median());
        Member* mem[3] = {&M4, &M1, &M3};
        for (Member* m : mem)
            std::printf("  %-52s total %7.1f us = device %7.1f + adapter %6.1f (of which the reference's own per-MapPoint getters %6.1f) | raw C ABI %7.1f | handle-per-object (round 4) %8.1f | %.0f matches/frame\n",
                        m->name, m->total.median(), m->device.median(), m->total.median() - m->device.median(), m->walk.median(), m->raw.median(), m->old.median(), (double)m->matches / F);
        std::printf("  frame, motion-model path (extract + Cur/Last search + local points)   median %.1f us  mean %.1f us\n", sMM.median(), sMM.mean());
        std::printf("  frame, reference-KF path (extract + ComputeBoW + BoW search + local)   median %.1f us  mean %.1f us\n", sKF.median(), sKF.mean());
        std::printf("  steady state: device allocations %lld -> %lld, pinned allocations %lld -> %lld, matcher handles %lld -> %lld: %s\n",
                    (long long)a0, (long long)a1, (long long)b0, (long long)b1, (long long)c0, (long long)c1, steady ? "none made" : "NOT steady");
        if (A.json) {
            std::printf("{\"frames\": %d, \"w\": %d, \"h\": %d, \"features\": %d, \"median_tracking_ms\": %.4f, \"mean_tracking_ms\": %.4f, "
                        "\"median_motion_model_frame_ms\": %.4f, \"median_reference_kf_frame_ms\": %.4f, \"extract_us\": %.1f, \"frame_members_us\": %.1f, \"compute_bow_us\": %.1f, \"members\": {",
                        F, A.w, A.h, A.features, sAll.median() * 1e-3, sAll.mean() * 1e-3, sMM.median() * 1e-3, sKF.median() * 1e-3, sExtract.median(), sFrameCtor.median(), sBoWCompute.median());
            for (int i = 0; i < 3; i++)
                std::printf("%s\"%s\": {\"total_us\": %.1f, \"mean_total_us\": %.1f, \"device_us\": %.1f, \"adapter_us\": %.1f, \"reference_getters_us\": %.1f, \"raw_c_abi_us\": %.1f, \"handle_per_object_us\": %.1f, \"matches_per_frame\": %.1f}",
                            i ? ", " : "", mem[i]->name, mem[i]->total.median(), mem[i]->total.mean(), mem[i]->device.median(), mem[i]->total.median() - mem[i]->device.median(), mem[i]->walk.median(),
                            mem[i]->raw.median(), mem[i]->old.median(), (double)mem[i]->matches / F);
            std::printf("}, \"steady_state\": {\"device_allocs\": %lld, \"pinned_allocs\": %lld, \"matcher_handles\": %lld, \"none_made\": %s}, \"checksum\": \"%016llx\"}\n",
                        (long long)(a1 - a0), (long long)(b1 - b0), (long long)(c1 - c0), steady ? "true" : "false", (unsigned long long)(checksum + (walkSink & 0)));
        }
        if (!steady) { std::fprintf(stderr, "tracking_loop: the steady-state loop allocated or made a handle\n"); return 3; }
        std::printf("tracking_loop ok\n");
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "tracking_loop: %s\n", e.what());
        return 1;
    }
}
