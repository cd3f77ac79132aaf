// GPU test of the ORBmatcher drop-in (include/ORBmatcher_hip.hpp: ORBmatcherT<Frame, KeyFrame, MapPoint>): every one of
// the reference's eleven member signatures (ORBmatcher.h:48-83) is instantiated on mock objects carrying the reference's
// member names (mock_slam.hpp), run on the GPU, and checked three ways:
//   (1) the flattened arrays the member handed to the device are fed to the C checker (oracle/, linked in): identical
//       device results (assign / match / best tables, counts);
//   (2) the write-back into the object graph (mvpMapPoints, vpMatched, vnMatches12, Replace / AddObservation ...) equals
//       what the reference's loop does with those results -- replayed here on a deep copy of the world;
//   (3) the flattening itself: query counts, projections and radii recomputed independently (double precision).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "ORBmatcher_hip.hpp"
#include "mock_slam.hpp"
#include "../../oracle/orb_oracle.h"

using namespace mock;
typedef iORB_SLAM::ORBmatcherT<Frame, KeyFrame, MapPoint> ORBmatcher;  // the typedef INTEGRATION.md asks the maintainer for

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

static const int W = 640, H = 480, NLEVELS = 8;
static const float FX = 520.f, FY = 515.f, CX = 318.f, CY = 243.f, SCALE = 1.2f;
static std::mt19937 rng(20260929);
static bool g_fuzz = false;   // other seeds than the committed one: the "scenario is meaningful" thresholds do not apply
#define SC(cond) ((cond) || g_fuzz)
static float urand(float a, float b) { return a + (b - a) * (float)(rng() & 0xFFFFFF) / 16777216.f; }

struct Pose { double R[9], t[3]; };
static Pose make_pose(double yawDeg, double pitchDeg, double tx, double ty, double tz)
{
    const double y = yawDeg * M_PI / 180, p = pitchDeg * M_PI / 180;
    const double Ry[9] = {cos(y), 0, sin(y), 0, 1, 0, -sin(y), 0, cos(y)}, Rx[9] = {1, 0, 0, 0, cos(p), -sin(p), 0, sin(p), cos(p)};
    Pose P;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { P.R[3 * r + c] = 0; for (int k = 0; k < 3; k++) P.R[3 * r + c] += Rx[3 * r + k] * Ry[3 * k + c]; }
    P.t[0] = tx; P.t[1] = ty; P.t[2] = tz;
    return P;
}
static void to_mat(const Pose& P, Mat& T)
{
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T.at<float>(r, c) = r == c ? 1.f : 0.f;
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T.at<float>(r, c) = (float)P.R[3 * r + c]; T.at<float>(r, 3) = (float)P.t[r]; }
}
static void cam_center(const Pose& P, double O[3]) { for (int i = 0; i < 3; i++) { O[i] = 0; for (int k = 0; k < 3; k++) O[i] -= P.R[3 * k + i] * P.t[k]; } }
static bool project(const Pose& P, const double X[3], double& u, double& v, double& z)
{
    double c[3];
    for (int i = 0; i < 3; i++) c[i] = P.R[3 * i] * X[0] + P.R[3 * i + 1] * X[1] + P.R[3 * i + 2] * X[2] + P.t[i];
    z = c[2];
    if (z <= 0.1) return false;
    u = FX * c[0] / z + CX; v = FY * c[1] / z + CY;
    return u >= 8 && u < W - 8 && v >= 8 && v < H - 8;
}

struct World {
    std::vector<std::unique_ptr<MapPoint> > mps;
    std::vector<std::unique_ptr<KeyFrame> > kfs;
    std::vector<double> X;               // 3 per point
    std::vector<uint8_t> trueDesc;       // 32 per point
    std::vector<float> baseAngle;
    std::vector<int> refLevel, origin;   // origin: the physical point behind a (twin) map point
    std::vector<float> scaleFactors, sigma2, invSigma2;
};

static void flip_bits(uint8_t* d, int n) { for (int i = 0; i < n; i++) d[rng() % 32] ^= (uint8_t)(1u << (rng() % 8)); }

// features of one view: (keypoints, descriptors, owning point or -1, feature vector)
struct View { std::vector<KeyPoint> keys; Mat desc; std::vector<int> owner; FeatureVector fv; };
static View make_view(const World& w, const Pose& P, float angleOffset, int nDistract, double keepProb)
{
    struct F { KeyPoint k; uint8_t d[32]; int owner; unsigned node; };
    std::vector<F> fs;
    double O[3]; cam_center(P, O);
    const int np = (int)w.X.size() / 3;
    for (int j = 0; j < np; j++) {
        double u, v, z;
        if (!project(P, &w.X[3 * j], u, v, z)) continue;
        if (urand(0, 1) > keepProb) continue;
        F f;
        const double dx = w.X[3 * j] - O[0], dy = w.X[3 * j + 1] - O[1], dz = w.X[3 * j + 2] - O[2];
        const double dist = std::sqrt(dx * dx + dy * dy + dz * dz);
        int lvl = (int)std::ceil(std::log(w.mps[j]->mfMaxDistance / dist) / std::log((double)SCALE));
        const int jit = (int)(rng() % 10);
        if (jit == 0) lvl--; else if (jit == 1) lvl++;
        lvl = std::max(0, std::min(NLEVELS - 1, lvl));
        f.k.pt.x = (float)u + urand(-0.8f, 0.8f); f.k.pt.y = (float)v + urand(-0.8f, 0.8f);
        f.k.size = 31.f * w.scaleFactors[lvl];
        float a = w.baseAngle[j] + angleOffset + urand(-4.f, 4.f);
        if (rng() % 25 == 0) a += 90.f;  // a few rotation outliers for the histogram to prune
        while (a < 0) a += 360.f;
        while (a >= 360.f) a -= 360.f;
        f.k.angle = a; f.k.response = 50.f + (float)(rng() % 100); f.k.octave = lvl; f.k.class_id = -1;
        std::memcpy(f.d, &w.trueDesc[(size_t)j * 32], 32);
        flip_bits(f.d, (int)(rng() % 22));
        f.owner = j; f.node = (unsigned)(j % 37) * 5u + 3u;
        fs.push_back(f);
    }
    for (int i = 0; i < nDistract; i++) {
        F f;
        f.k.pt.x = urand(8, W - 8); f.k.pt.y = urand(8, H - 8);
        f.k.octave = (int)(rng() % NLEVELS); f.k.size = 31.f * w.scaleFactors[f.k.octave];
        f.k.angle = urand(0, 359.9f); f.k.response = 30.f; f.k.class_id = -1;
        for (int b = 0; b < 32; b++) f.d[b] = (uint8_t)rng();
        if (rng() % 3 == 0) {  // near-duplicates of real descriptors: second-best pressure on the ratio tests
            const int j = (int)(rng() % np);
            std::memcpy(f.d, &w.trueDesc[(size_t)j * 32], 32);
            flip_bits(f.d, 25 + (int)(rng() % 30));
        }
        f.owner = -1; f.node = (unsigned)(rng() % 37) * 5u + 3u;
        fs.push_back(f);
    }
    std::shuffle(fs.begin(), fs.end(), rng);
    View V;
    V.desc = Mat::u8((int)fs.size(), 32);
    for (size_t i = 0; i < fs.size(); i++) {
        V.keys.push_back(fs[i].k);
        std::memcpy(V.desc.ptr<uint8_t>((int)i), fs[i].d, 32);
        V.owner.push_back(fs[i].owner);
        V.fv[fs[i].node].push_back((unsigned)i);
    }
    return V;
}

static KeyFrame* add_keyframe(World& w, const Pose& P, float angleOffset, double mpProb, bool stereo = false)
{
    std::unique_ptr<KeyFrame> kf(new KeyFrame());
    View V = make_view(w, P, angleOffset, 250, 0.85);
    kf->N = (int)V.keys.size(); kf->mvKeysUn = V.keys; kf->mDescriptors = V.desc; kf->mFeatVec = V.fv;
    kf->mvuRight.assign(kf->N, -1.f);
    kf->fx = FX; kf->fy = FY; kf->cx = CX; kf->cy = CY; kf->mbf = stereo ? 40.f : 0.f;
    kf->mvScaleFactors = w.scaleFactors; kf->mvLevelSigma2 = w.sigma2; kf->mvInvLevelSigma2 = w.invSigma2;
    kf->mfLogScaleFactor = std::log(SCALE);
    kf->mnMinX = 0; kf->mnMinY = 0; kf->mnMaxX = W; kf->mnMaxY = H;
    kf->mfGridElementWidthInv = 64.f / W; kf->mfGridElementHeightInv = 48.f / H;
    to_mat(P, kf->Tcw);
    double O[3]; cam_center(P, O);
    for (int i = 0; i < 3; i++) kf->Ow.at<float>(i, 0) = (float)O[i];
    kf->mvpMapPoints.assign(kf->N, nullptr);
    kf->id = (int)w.kfs.size();
    for (int i = 0; i < kf->N; i++) {
        const int j = V.owner[i];
        if (stereo && j >= 0 && rng() % 2) {
            double u, v, z; project(P, &w.X[3 * j], u, v, z);
            kf->mvuRight[i] = kf->mvKeysUn[i].pt.x - kf->mbf / (float)z + urand(-0.5f, 0.5f);
        }
        if (j >= 0 && urand(0, 1) < mpProb && !w.mps[j]->IsInKeyFrame(kf.get())) {
            kf->mvpMapPoints[i] = w.mps[j].get();
            w.mps[j]->AddObservation(kf.get(), i);
        }
    }
    w.kfs.push_back(std::move(kf));
    return w.kfs.back().get();
}

static void fill_frame(const World& w, Frame& F, const Pose& P, float angleOffset, std::vector<int>& owner, bool stereo = false)
{
    View V = make_view(w, P, angleOffset, 250, 0.85);
    F.N = (int)V.keys.size(); F.mvKeys = V.keys; F.mvKeysUn = V.keys; F.mDescriptors = V.desc; F.mFeatVec = V.fv;
    for (int i = 0; i < F.N; i++) F.mvKeys[i].angle = V.keys[i].angle;  // mvKeys / mvKeysUn differ in pt only
    F.mvuRight.assign(F.N, -1.f);
    F.mvpMapPoints.assign(F.N, nullptr); F.mvbOutlier.assign(F.N, false);
    to_mat(P, F.mTcw);
    F.mvScaleFactors = w.scaleFactors; F.mfLogScaleFactor = std::log(SCALE);
    F.mb = stereo ? 0.08f : 0.f; F.mbf = stereo ? 40.f : 0.f;
    owner = V.owner;
    if (stereo)
        for (int i = 0; i < F.N; i++) {
            const int j = owner[i];
            if (j >= 0 && rng() % 2) { double u, v, z; project(P, &w.X[3 * j], u, v, z); F.mvuRight[i] = F.mvKeysUn[i].pt.x - F.mbf / (float)z + urand(-0.6f, 0.6f); }
        }
}

static void build_world(World& w, int np)
{
    w.scaleFactors.resize(NLEVELS); w.sigma2.resize(NLEVELS); w.invSigma2.resize(NLEVELS);
    w.scaleFactors[0] = 1.f;
    for (int l = 1; l < NLEVELS; l++) w.scaleFactors[l] = w.scaleFactors[l - 1] * SCALE;
    for (int l = 0; l < NLEVELS; l++) { w.sigma2[l] = w.scaleFactors[l] * w.scaleFactors[l]; w.invSigma2[l] = 1.f / w.sigma2[l]; }
    for (int j = 0; j < np; j++) {
        const double X[3] = {urand(-7, 7), urand(-3.5f, 3.5f), urand(6, 15)};
        w.X.insert(w.X.end(), X, X + 3);
        std::unique_ptr<MapPoint> mp(new MapPoint());
        mp->id = j;
        const double d = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
        for (int i = 0; i < 3; i++) { mp->mWorldPos.at<float>(i, 0) = (float)X[i]; mp->mNormalVector.at<float>(i, 0) = (float)(X[i] / d); }
        const int lvl = (int)(rng() % 5);
        w.refLevel.push_back(lvl); w.origin.push_back(j);
        mp->mfMaxDistance = (float)d * w.scaleFactors[lvl];                      // MapPoint.cc:357-371 UpdateNormalAndDepth
        mp->mfMinDistance = mp->mfMaxDistance / w.scaleFactors[NLEVELS - 1];
        for (int b = 0; b < 32; b++) { const uint8_t v = (uint8_t)rng(); w.trueDesc.push_back(v); mp->mDescriptor.at<uint8_t>(0, b) = v; }
        flip_bits(mp->mDescriptor.ptr<uint8_t>(0), (int)(rng() % 8));
        w.baseAngle.push_back(urand(0, 359.f));
        w.mps.push_back(std::move(mp));
    }
}

// "Twin" map points: a second MapPoint for a point a keyframe already holds (what Fuse exists to merge), observed elsewhere
static void add_twins(World& w, KeyFrame* holder, KeyFrame* other1, KeyFrame* other2, int count)
{
    int made = 0;
    for (int i = 0; i < holder->N && made < count; i++) {
        MapPoint* p = holder->mvpMapPoints[i];
        if (!p || p->id >= (int)w.refLevel.size() || rng() % 3) continue;
        const int j = p->id;
        std::unique_ptr<MapPoint> t(new MapPoint(*p));
        t->id = (int)w.mps.size();
        t->mObservations.clear();
        t->mDescriptor = p->mDescriptor.clone();
        flip_bits(t->mDescriptor.ptr<uint8_t>(0), (int)(rng() % 6));
        KeyFrame* hosts[2] = {other1, made % 2 ? other2 : nullptr};
        for (int k = 0; k < 2; k++) {
            KeyFrame* h = hosts[k];
            if (!h) continue;
            for (int s = (int)(rng() % h->N), tries = 0; tries < h->N; s = (s + 1) % h->N, tries++)
                if (!h->mvpMapPoints[s]) { h->mvpMapPoints[s] = t.get(); t->AddObservation(h, s); break; }
        }
        for (int k = 0; k < 3; k++) w.X.push_back(w.X[3 * j + k]);
        for (int k = 0; k < 32; k++) w.trueDesc.push_back(w.trueDesc[(size_t)j * 32 + k]);
        w.baseAngle.push_back(w.baseAngle[j]); w.refLevel.push_back(w.refLevel[j]); w.origin.push_back(w.origin[j]);
        w.mps.push_back(std::move(t));
        made++;
    }
}

// ------------------------------------------------------------------ checker glue
struct Grid { OrcGridParams gp; std::vector<int32_t> start, idx; };
static Grid oracle_grid(const OrbmGrid& g, const OrbxKeyPoint* keys, int n)
{
    Grid G;
    G.gp.minX = g.minX; G.gp.minY = g.minY; G.gp.invW = g.invW; G.gp.invH = g.invH; G.gp.cols = g.cols; G.gp.rows = g.rows;
    G.start.assign((size_t)g.cols * g.rows + 1, 0); G.idx.assign(std::max(n, 1), 0);
    orc_grid_build(&G.gp, (const OrcKeyPoint*)keys, n, G.start.data(), G.idx.data());
    return G;
}
static OrcFeatVec ofv(const iORB_SLAM::FlatFeatVec& f) { OrcFeatVec o = {(int)f.node_id.size(), f.node_id.data(), f.start.data(), f.idx.data()}; return o; }

// (1) the projection family: flattened call -> checker -> identical assign / occupancy / count
static void check_projection(const ORBmatcher::FlatCall& c, float nnratio, bool checkOri, std::vector<int32_t>& assignOut)
{
    Grid G = oracle_grid(c.grid, c.tkeys, c.nt);
    OrcProjParams pp = {c.mode, nnratio, checkOri ? 1 : 0, c.thDist};
    std::vector<uint8_t> occ = c.tocc_in;
    assignOut.assign(c.nt, -2);
    const bool stereo = c.turight != nullptr;
    const int n = orc_search_by_projection_stereo(&pp, c.q_uvr.data(), stereo ? c.q_ur.data() : nullptr, c.q_lvl.data(), c.qdesc.data(), c.qangle.data(),
                                                  nullptr, c.qobs.data(), c.nq, &G.gp, (const OrcKeyPoint*)c.tkeys, G.start.data(), G.idx.data(),
                                                  c.tdesc, c.turight, c.nt, occ.data(), assignOut.data());
    EXPECT(n == c.nmatches);
    EXPECT(assignOut == c.assign);
    EXPECT(occ == c.tocc);
    EXPECT((int)c.q_uvr.size() == 3 * c.nq && (int)c.qdesc.size() == 32 * c.nq && (int)c.qidx.size() == c.nq);
}
static void check_window(const ORBmatcher::FlatCall& c, const std::vector<float>& invSigma2, std::vector<int32_t>& bi, std::vector<int32_t>& bd)
{
    Grid G = oracle_grid(c.grid, c.tkeys, c.nt);
    bi.assign(std::max(c.nq, 1), -1); bd.assign(std::max(c.nq, 1), 256);
    orc_window_best(c.q_uvr.data(), c.chi2 ? c.q_ur.data() : nullptr, c.q_pred.data(), c.qdesc.data(), nullptr, c.nq, &G.gp, (const OrcKeyPoint*)c.tkeys,
                    G.start.data(), G.idx.data(), c.tdesc, c.chi2 ? c.turight : nullptr, c.nt, invSigma2.data(), c.chi2, bi.data(), bd.data());
    bi.resize(c.nq); bd.resize(c.nq);
    EXPECT(bi == c.bestIdx);
    EXPECT(bd == c.bestDist);
}

// deep copy of the world's object graph (ids keep their meaning) for replaying write-backs
struct Shadow {
    std::vector<std::unique_ptr<MapPoint> > mps;
    std::vector<std::unique_ptr<KeyFrame> > kfs;
    MapPoint* mp(const MapPoint* p) const { return p ? mps[p->id].get() : nullptr; }
    KeyFrame* kf(const KeyFrame* k) const { return kfs[k->id].get(); }
};
static void clone_world(const World& w, Shadow& s)
{
    for (size_t i = 0; i < w.mps.size(); i++) { s.mps.push_back(std::unique_ptr<MapPoint>(new MapPoint(*w.mps[i]))); }
    for (size_t i = 0; i < w.kfs.size(); i++) { s.kfs.push_back(std::unique_ptr<KeyFrame>(new KeyFrame(*w.kfs[i]))); }
    for (size_t i = 0; i < s.mps.size(); i++) {
        std::map<KeyFrame*, size_t> o;
        for (std::map<KeyFrame*, size_t>::iterator it = w.mps[i]->mObservations.begin(); it != w.mps[i]->mObservations.end(); ++it) o[s.kfs[it->first->id].get()] = it->second;
        s.mps[i]->mObservations = o;
        s.mps[i]->mpReplaced = w.mps[i]->mpReplaced ? s.mps[w.mps[i]->mpReplaced->id].get() : nullptr;
    }
    for (size_t k = 0; k < s.kfs.size(); k++)
        for (size_t i = 0; i < s.kfs[k]->mvpMapPoints.size(); i++) s.kfs[k]->mvpMapPoints[i] = s.mp(w.kfs[k]->mvpMapPoints[i]);
}
static bool same_state(const World& w, const Shadow& s)
{
    for (size_t i = 0; i < w.mps.size(); i++) {
        const MapPoint &a = *w.mps[i], &b = *s.mps[i];
        if (a.mbBad != b.mbBad || a.mObservations.size() != b.mObservations.size()) return false;
        if ((a.mpReplaced ? a.mpReplaced->id : -1) != (b.mpReplaced ? b.mpReplaced->id : -1)) return false;
        for (std::map<KeyFrame*, size_t>::const_iterator it = a.mObservations.begin(); it != a.mObservations.end(); ++it) {
            std::map<KeyFrame*, size_t>::const_iterator jt = b.mObservations.find(s.kfs[it->first->id].get());
            if (jt == b.mObservations.end() || jt->second != it->second) return false;
        }
    }
    for (size_t k = 0; k < w.kfs.size(); k++)
        for (size_t i = 0; i < w.kfs[k]->mvpMapPoints.size(); i++) {
            const MapPoint *a = w.kfs[k]->mvpMapPoints[i], *b = s.kfs[k]->mvpMapPoints[i];
            if ((a ? a->id : -1) != (b ? b->id : -1)) return false;
        }
    return true;
}

int main(int argc, char** argv)
{
    // optional: seed and world size (tests/soak/fuzz_dropin.sh runs the whole program over many seeds)
    if (argc > 1) { rng.seed((unsigned)std::strtoul(argv[1], nullptr, 10)); g_fuzz = true; }
    World w;
    build_world(w, argc > 2 ? std::atoi(argv[2]) : 1400);
    Frame::fx = FX; Frame::fy = FY; Frame::cx = CX; Frame::cy = CY;
    Frame::mnMinX = 0; Frame::mnMaxX = W; Frame::mnMinY = 0; Frame::mnMaxY = H;
    Frame::mfGridElementWidthInv = 64.f / W; Frame::mfGridElementHeightInv = 48.f / H;
    const Pose P0 = make_pose(0, 0, 0, 0, 0), P1 = make_pose(2.5, -1.0, 0.25, -0.05, 0.1), P2 = make_pose(-3.0, 1.5, -0.3, 0.08, -0.15),
               P3 = make_pose(1.0, 0.5, 0.05, 0.02, -0.6);
    KeyFrame* kf0 = add_keyframe(w, P0, 0.f, 0.7);
    KeyFrame* kf1 = add_keyframe(w, P1, 12.f, 0.6);
    KeyFrame* kf2 = add_keyframe(w, P2, -20.f, 0.5, /*stereo=*/true);
    add_twins(w, kf1, kf0, kf2, 90);
    add_twins(w, kf2, kf0, kf1, 90);

    // ------------------------------------------------------------ 1. SearchByProjection(Frame&, vector<MapPoint*>&, th)
    for (int variant = 0; variant < 3; variant++) {
        const bool stereo = variant == 2;
        const float th = variant == 0 ? 1.0f : 3.0f;
        Frame F; std::vector<int> owner;
        fill_frame(w, F, P1, 12.f, owner, stereo);
        std::vector<MapPoint*> vp;
        int expectQueries = 0;
        for (size_t j = 0; j < w.mps.size(); j++) {
            MapPoint* p = w.mps[j].get();
            double u, v, z;
            p->mbTrackInView = project(P1, &w.X[3 * j], u, v, z) && rng() % 8 != 0;
            if (p->mbTrackInView) {
                p->mTrackProjX = (float)u + urand(-1, 1); p->mTrackProjY = (float)v + urand(-1, 1);
                p->mTrackProjXR = p->mTrackProjX - 40.f / (float)z;
                p->mnTrackScaleLevel = std::max(0, std::min(NLEVELS - 1, w.refLevel[j] + (int)(rng() % 3) - 1));
                p->mTrackViewCos = rng() % 2 ? 0.9995f : 0.93f;
            }
            if (rng() % 40 == 0) p->mbBad = true;
            vp.push_back(p);
            expectQueries += p->mbTrackInView && !p->mbBad;
        }
        // features that already hold a MapPoint: with observations (blocks) and without (may be taken over)
        std::unique_ptr<MapPoint> loose(new MapPoint()); loose->id = -1;
        for (int t = 0; t < F.N; t++) {
            if (rng() % 9 == 0 && owner[t] >= 0) F.mvpMapPoints[t] = w.mps[owner[t]].get();
            else if (rng() % 15 == 0) F.mvpMapPoints[t] = loose.get();
        }
        const std::vector<MapPoint*> before = F.mvpMapPoints;
        ORBmatcher m(0.8f, true);
        const int n = m.SearchByProjection(F, vp, th);
        const ORBmatcher::FlatCall& c = m.last();
        std::vector<int32_t> as;
        check_projection(c, 0.8f, true, as);
        EXPECT(c.nq == expectQueries && c.mode == 3 && c.thDist == 100 && n == c.nmatches && SC(n > 250));
        EXPECT((c.turight != nullptr) == stereo);
        for (int q = 0; q < c.nq; q++) {  // (3) the flattening
            MapPoint* p = vp[c.qidx[q]];
            const float r = (p->mTrackViewCos > 0.998 ? 2.5f : 4.0f) * (th != 1.0f ? th : 1.0f);
            EXPECT(c.q_uvr[3 * q] == p->mTrackProjX && c.q_uvr[3 * q + 1] == p->mTrackProjY && c.q_uvr[3 * q + 2] == r * w.scaleFactors[p->mnTrackScaleLevel]);
            EXPECT(c.q_lvl[2 * q] == p->mnTrackScaleLevel - 1 && c.q_lvl[2 * q + 1] == p->mnTrackScaleLevel);
            EXPECT(std::memcmp(&c.qdesc[(size_t)q * 32], p->mDescriptor.ptr<uint8_t>(0), 32) == 0 && c.qobs[q] == (p->Observations() > 0));
            if (q) EXPECT(c.qidx[q] > c.qidx[q - 1]);
        }
        for (int t = 0; t < F.N; t++) {   // (2) the write-back (:123)
            MapPoint* want = as[t] >= 0 ? vp[c.qidx[as[t]]] : before[t];
            EXPECT(F.mvpMapPoints[t] == want);
            EXPECT(c.tocc_in[t] == (before[t] && before[t]->Observations() > 0));
        }
        for (size_t j = 0; j < w.mps.size(); j++) w.mps[j]->mbBad = false;
        std::printf("1.%d SearchByProjection(Frame, MapPoints, th=%g%s): %d queries, %d matches\n", variant, th, stereo ? ", stereo" : "", c.nq, n);
    }

    // ------------------------------------------------------------ 2. SearchByProjection(Frame& Cur, const Frame& Last, th, bMono)
    for (int variant = 0; variant < 3; variant++) {
        const bool bMono = variant == 0;
        Frame Last, Cur; std::vector<int> ownerL, ownerC;
        fill_frame(w, Last, P0, 0.f, ownerL, !bMono);
        const Pose Pc = variant == 2 ? P3 : P1;   // P1: the camera centre lies behind the last one by more than the baseline (bBackward), P3: ahead (bForward)
        fill_frame(w, Cur, Pc, 7.f, ownerC, !bMono);
        int nLastPts = 0;
        for (int i = 0; i < Last.N; i++) {
            if (ownerL[i] >= 0 && rng() % 5) { Last.mvpMapPoints[i] = w.mps[ownerL[i]].get(); nLastPts++; }
            if (rng() % 17 == 0) Last.mvbOutlier[i] = true;
        }
        std::unique_ptr<MapPoint> loose(new MapPoint()); loose->id = -1;
        for (int t = 0; t < Cur.N; t++) {
            if (rng() % 11 == 0 && ownerC[t] >= 0) Cur.mvpMapPoints[t] = w.mps[ownerC[t]].get();
            else if (rng() % 13 == 0) Cur.mvpMapPoints[t] = loose.get();
        }
        const std::vector<MapPoint*> before = Cur.mvpMapPoints;
        ORBmatcher m(0.9f, true);
        const float th = variant == 1 ? 7.f : 15.f;
        const int n = m.SearchByProjection(Cur, Last, th, bMono);
        const ORBmatcher::FlatCall& c = m.last();
        std::vector<int32_t> as;
        check_projection(c, 0.9f, true, as);
        EXPECT(c.mode == 4 && c.thDist == 100 && n == c.nmatches && SC(n > 150) && SC(c.nq > 300) && c.nq <= nLastPts);
        EXPECT((c.turight != nullptr) == !bMono);
        // the camera's motion along the last frame's optical axis against the baseline mb (:1349-1350); Last sits at the origin
        double Oc[3]; cam_center(Pc, Oc);
        const bool fwd = !bMono && Oc[2] > Cur.mb, bwd = !bMono && -Oc[2] > Cur.mb;
        EXPECT(variant == 0 ? (!fwd && !bwd) : (variant == 1 ? bwd : fwd));
        int pruned = 0;
        for (int q = 0; q < c.nq; q++) {
            const int i = c.qidx[q];
            const int j = Last.mvpMapPoints[i]->id;
            double u, v, z;
            project(Pc, &w.X[3 * j], u, v, z);   // independent projection, double precision
            EXPECT(std::fabs(c.q_uvr[3 * q] - u) < 2e-2 && std::fabs(c.q_uvr[3 * q + 1] - v) < 2e-2);
            const int oct = Last.mvKeys[i].octave;
            EXPECT(c.q_uvr[3 * q + 2] == th * w.scaleFactors[oct] && !Last.mvbOutlier[i] && c.qangle[q] == Last.mvKeysUn[i].angle);
            if (fwd) EXPECT(c.q_lvl[2 * q] == oct && c.q_lvl[2 * q + 1] == -1);            // GetFeaturesInArea(u, v, radius, nLastOctave)
            else if (bwd) EXPECT(c.q_lvl[2 * q] == 0 && c.q_lvl[2 * q + 1] == oct);       // (.., 0, nLastOctave)
            else EXPECT(c.q_lvl[2 * q] == oct - 1 && c.q_lvl[2 * q + 1] == oct + 1);
            if (!bMono) EXPECT(std::fabs(c.q_ur[q] - (u - 40.0 / z)) < 2e-2);
        }
        for (int t = 0; t < Cur.N; t++) {   // :1430 and the pruning :1462
            MapPoint* want = as[t] >= 0 ? Last.mvpMapPoints[c.qidx[as[t]]] : (as[t] == -1 ? nullptr : before[t]);
            EXPECT(Cur.mvpMapPoints[t] == want);
            pruned += as[t] == -1;
        }
        EXPECT(SC(pruned > 0));
        std::printf("2.%d SearchByProjection(Cur, Last, th=%g, bMono=%d): %d queries, %d matches, %d pruned by rotation\n", variant, th, (int)bMono, c.nq, n, pruned);
    }

    // ------------------------------------------------------------ 3. SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist)
    for (int variant = 0; variant < 2; variant++) {
        Frame Cur; std::vector<int> ownerC;
        fill_frame(w, Cur, P1, 12.f, ownerC);
        std::set<MapPoint*> found;
        for (int t = 0; t < Cur.N; t++)
            if (rng() % 6 == 0 && ownerC[t] >= 0) { Cur.mvpMapPoints[t] = w.mps[ownerC[t]].get(); found.insert(Cur.mvpMapPoints[t]); }
        const std::vector<MapPoint*> before = Cur.mvpMapPoints;
        ORBmatcher m(0.9f, true);
        const int ORBdist = variant ? 64 : 100;
        const float th = variant ? 3.f : 10.f;
        const int n = m.SearchByProjection(Cur, kf0, found, th, ORBdist);
        const ORBmatcher::FlatCall& c = m.last();
        std::vector<int32_t> as;
        check_projection(c, 0.9f, true, as);
        EXPECT(c.mode == 5 && c.thDist == ORBdist && n == c.nmatches && SC(n > 100));
        for (int q = 0; q < c.nq; q++) {
            MapPoint* p = kf0->mvpMapPoints[c.qidx[q]];
            EXPECT(p && !found.count(p) && c.qangle[q] == kf0->mvKeysUn[c.qidx[q]].angle);
            double u, v, z; project(P1, &w.X[3 * p->id], u, v, z);
            EXPECT(std::fabs(c.q_uvr[3 * q] - u) < 2e-2 && std::fabs(c.q_uvr[3 * q + 1] - v) < 2e-2);
            EXPECT(c.q_lvl[2 * q + 1] - c.q_lvl[2 * q] == 2);
        }
        for (int t = 0; t < Cur.N; t++) {
            MapPoint* want = as[t] >= 0 ? kf0->mvpMapPoints[c.qidx[as[t]]] : (as[t] == -1 ? nullptr : before[t]);
            EXPECT(Cur.mvpMapPoints[t] == want);
            EXPECT(c.tocc_in[t] == (before[t] != nullptr));
        }
        std::printf("3.%d SearchByProjection(Cur, KF, found, th=%g, ORBdist=%d): %d queries, %d matches\n", variant, th, ORBdist, c.nq, n);
    }

    // ------------------------------------------------------------ 4. SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)
    {
        const float s = 1.3f;
        Mat Scw = Mat::f32(4, 4);
        to_mat(P1, Scw);
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 4; cc++) Scw.at<float>(r, cc) *= s;   // [sR | s t]
        std::vector<MapPoint*> vpPoints, vpMatched(kf1->N, nullptr);
        for (size_t j = 0; j < w.mps.size(); j++) if (rng() % 4) vpPoints.push_back(w.mps[j].get());
        for (int t = 0; t < kf1->N; t++) if (kf1->mvpMapPoints[t] && rng() % 3 == 0) vpMatched[t] = kf1->mvpMapPoints[t];
        const std::vector<MapPoint*> before = vpMatched;
        ORBmatcher m(0.75f, true);
        const int n = m.SearchByProjection(kf1, Scw, vpPoints, vpMatched, 10);
        const ORBmatcher::FlatCall& c = m.last();
        std::vector<int32_t> as;
        check_projection(c, 0.75f, true, as);
        EXPECT(c.mode == 6 && c.thDist == 50 && n == c.nmatches && SC(n > 200));
        std::set<MapPoint*> already(before.begin(), before.end());
        for (int q = 0; q < c.nq; q++) {
            MapPoint* p = vpPoints[c.qidx[q]];
            EXPECT(!already.count(p));
            double u, v, z; project(P1, &w.X[3 * p->id], u, v, z);    // Scw / scw is the plain pose again
            EXPECT(std::fabs(c.q_uvr[3 * q] - u) < 5e-2 && std::fabs(c.q_uvr[3 * q + 1] - v) < 5e-2);
            EXPECT(c.q_lvl[2 * q + 1] - c.q_lvl[2 * q] == 1);
        }
        for (int t = 0; t < kf1->N; t++) EXPECT(vpMatched[t] == (as[t] >= 0 ? vpPoints[c.qidx[as[t]]] : before[t]));
        std::printf("4   SearchByProjection(KF, Scw, points, matched, 10): %d queries, %d matches\n", c.nq, n);
    }

    // ------------------------------------------------------------ 5. SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)
    {
        Frame F; std::vector<int> owner;
        fill_frame(w, F, P1, 12.f, owner);
        w.mps[5]->mbBad = true; w.mps[77]->mbBad = true;
        std::vector<MapPoint*> out(3, w.mps[0].get());
        ORBmatcher m(0.7f, true);
        const int n = m.SearchByBoW(kf0, F, out);
        const ORBmatcher::FlatCall& c = m.last();
        std::vector<int32_t> om(F.N, -1);
        OrcFeatVec a = ofv(c.qfv), b = ofv(c.tfv);
        const int on = orc_search_by_bow(kf0->mDescriptors.ptr<uint8_t>(0), c.qangle.data(), c.qvalid.data(), c.nq, &a, F.mDescriptors.ptr<uint8_t>(0),
                                         c.tangle.data(), nullptr, c.nt, &b, 0.7f, 1, 1, om.data());
        EXPECT(on == n && SC(n > 150) && (int)out.size() == F.N && om == c.match);
        for (int i = 0; i < c.nq; i++) EXPECT(c.qvalid[i] == (kf0->mvpMapPoints[i] && !kf0->mvpMapPoints[i]->mbBad) && c.qangle[i] == kf0->mvKeysUn[i].angle);
        int right = 0;
        for (int t = 0; t < F.N; t++) {
            EXPECT(out[t] == (om[t] >= 0 ? kf0->mvpMapPoints[om[t]] : nullptr));
            right += out[t] && owner[t] >= 0 && w.origin[owner[t]] == w.origin[out[t]->id];
        }
        EXPECT(SC(right > n * 8 / 10));   // the scenario is meaningful: matches are mostly the true correspondences (twins sit on unrelated features of kf0)
        w.mps[5]->mbBad = false; w.mps[77]->mbBad = false;
        std::printf("5   SearchByBoW(KF, Frame): %d matches (%d true correspondences)\n", n, right);
    }
    // ------------------------------------------------------------ 6. SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&)
    {
        std::vector<MapPoint*> out;
        w.mps[9]->mbBad = true;
        ORBmatcher m(0.8f, true);
        const int n = m.SearchByBoW(kf0, kf1, out);
        const ORBmatcher::FlatCall& c = m.last();
        std::vector<int32_t> om(kf0->N, -1);
        OrcFeatVec a = ofv(c.qfv), b = ofv(c.tfv);
        const int on = orc_search_by_bow(kf0->mDescriptors.ptr<uint8_t>(0), c.qangle.data(), c.qvalid.data(), c.nq, &a, kf1->mDescriptors.ptr<uint8_t>(0),
                                         c.tangle.data(), c.tvalid.data(), c.nt, &b, 0.8f, 1, 0, om.data());
        EXPECT(on == n && SC(n > 100) && (int)out.size() == kf0->N && om == c.match);
        for (int i = 0; i < c.nt; i++) EXPECT(c.tvalid[i] == (kf1->mvpMapPoints[i] && !kf1->mvpMapPoints[i]->mbBad));
        for (int i = 0; i < kf0->N; i++) EXPECT(out[i] == (om[i] >= 0 ? kf1->mvpMapPoints[om[i]] : nullptr));
        w.mps[9]->mbBad = false;
        std::printf("6   SearchByBoW(KF, KF): %d matches\n", n);
    }
    // ------------------------------------------------------------ 7. SearchForInitialization
    {
        Frame F1, F2; std::vector<int> o1, o2;
        fill_frame(w, F1, P0, 0.f, o1);
        fill_frame(w, F2, make_pose(0.4, 0.1, 0.05, 0.0, 0.0), 3.f, o2);
        std::vector<Point2f> prev(F1.N);
        for (int i = 0; i < F1.N; i++) prev[i] = F1.mvKeysUn[i].pt;
        const std::vector<Point2f> prev0 = prev;
        std::vector<int> m12;
        ORBmatcher m(0.9f, true);
        const int n = m.SearchForInitialization(F1, F2, prev, m12, 100);
        const ORBmatcher::FlatCall& c = m.last();
        Grid G = oracle_grid(c.grid, c.tkeys, c.nt);
        std::vector<int32_t> om(F1.N, -1);
        const int on = orc_search_for_initialization(c.q_xy.data(), 100.f, (const OrcKeyPoint*)c.qkeys, F1.mDescriptors.ptr<uint8_t>(0), c.nq, &G.gp,
                                                     (const OrcKeyPoint*)c.tkeys, G.start.data(), G.idx.data(), F2.mDescriptors.ptr<uint8_t>(0), c.nt, 0.9f, 1, om.data());
        EXPECT(on == n && SC(n > 40) && (int)m12.size() == F1.N);
        for (int i = 0; i < F1.N; i++) {
            EXPECT(m12[i] == om[i]);
            if (m12[i] >= 0) EXPECT(prev[i].x == F2.mvKeysUn[m12[i]].pt.x && prev[i].y == F2.mvKeysUn[m12[i]].pt.y);  // :517-519
            else EXPECT(prev[i].x == prev0[i].x && prev[i].y == prev0[i].y);
        }
        std::printf("7   SearchForInitialization: %d matches\n", n);
    }
    // ------------------------------------------------------------ 8. SearchForTriangulation
    for (int variant = 0; variant < 2; variant++) {
        KeyFrame *A = variant ? kf2 : kf0, *B = kf1;
        const Pose &PA = variant ? P2 : P0, &PB = P1;
        // F12 = K^-T [t12]x R12 K^-1 (Optimizer-side algebra of the caller; any matrix would do for parity)
        double R12[9], t12[3], OA[3];
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) { R12[3 * r + cc] = 0; for (int k = 0; k < 3; k++) R12[3 * r + cc] += PA.R[3 * r + k] * PB.R[3 * cc + k]; }
        cam_center(PA, OA);
        for (int r = 0; r < 3; r++) { t12[r] = PA.t[r]; for (int k = 0; k < 3; k++) t12[r] -= R12[3 * r + k] * PB.t[k]; }
        const double tx[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
        const double Kinv[9] = {1.0 / FX, 0, -CX / FX, 0, 1.0 / FY, -CY / FY, 0, 0, 1};
        double E[9], T1[9], F[9];
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) { E[3 * r + cc] = 0; for (int k = 0; k < 3; k++) E[3 * r + cc] += tx[3 * r + k] * R12[3 * k + cc]; }
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) { T1[3 * r + cc] = 0; for (int k = 0; k < 3; k++) T1[3 * r + cc] += Kinv[3 * k + r] * E[3 * k + cc]; }
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) { F[3 * r + cc] = 0; for (int k = 0; k < 3; k++) F[3 * r + cc] += T1[3 * r + k] * Kinv[3 * k + cc]; }
        Mat F12 = Mat::f32(3, 3);
        for (int i = 0; i < 9; i++) F12.at<float>(i / 3, i % 3) = (float)F[i];
        std::vector<std::pair<size_t, size_t> > pairs;
        ORBmatcher m(0.6f, variant == 0);
        const int n = m.SearchForTriangulation(A, B, F12, pairs, false);
        const ORBmatcher::FlatCall& c = m.last();
        std::vector<int32_t> om(A->N, -1);
        OrcFeatVec a = ofv(c.qfv), b = ofv(c.tfv);
        const int on = orc_search_for_triangulation((const OrcKeyPoint*)c.qkeys, A->mDescriptors.ptr<uint8_t>(0), c.skip1.data(), A->mvuRight.data(), A->N, &a,
                                                    (const OrcKeyPoint*)c.tkeys, B->mDescriptors.ptr<uint8_t>(0), c.skip2.data(), B->mvuRight.data(), B->N, &b,
                                                    c.F12.data(), c.ex, c.ey, B->mvScaleFactors.data(), B->mvLevelSigma2.data(), 0, variant == 0, om.data());
        EXPECT(on == n && SC(n > 20) && (int)pairs.size() == n);
        size_t k = 0;
        for (int i = 0; i < A->N; i++) if (om[i] >= 0) { EXPECT(k < pairs.size() && pairs[k].first == (size_t)i && pairs[k].second == (size_t)om[i]); k++; }
        for (int i = 0; i < A->N; i++) EXPECT(c.skip1[i] == (A->mvpMapPoints[i] != nullptr));
        double e[3]; for (int i = 0; i < 3; i++) e[i] = PB.R[3 * i] * OA[0] + PB.R[3 * i + 1] * OA[1] + PB.R[3 * i + 2] * OA[2] + PB.t[i];
        EXPECT(std::fabs(c.ex - (FX * e[0] / e[2] + CX)) < 0.5 + 1e-3 * std::fabs(c.ex) && std::fabs(c.ey - (FY * e[1] / e[2] + CY)) < 0.5 + 1e-3 * std::fabs(c.ey));
        std::printf("8.%d SearchForTriangulation: %d pairs\n", variant, n);
    }
    // ------------------------------------------------------------ 9. Fuse(KeyFrame*, vector<MapPoint*>&, th)
    for (int variant = 0; variant < 2; variant++) {
        KeyFrame* kf = variant ? kf2 : kf1;    // kf2 carries right coordinates: the 3-term chi-square
        std::vector<MapPoint*> cand;
        for (size_t j = 0; j < w.mps.size(); j++) if (rng() % 3) cand.push_back(w.mps[j].get());
        for (int i = 0; i < 40; i++) cand.push_back(cand[rng() % cand.size()]);   // duplicates: IsInKeyFrame changes under the loop
        cand.push_back(nullptr);
        w.mps[21]->mbBad = true;
        Shadow sh; clone_world(w, sh);
        ORBmatcher m(0.6f, true);
        const int n = m.Fuse(kf, cand, 3.0f);
        const ORBmatcher::FlatCall& c = m.last();
        std::vector<int32_t> bi, bd;
        check_window(c, kf->mvInvLevelSigma2, bi, bd);
        EXPECT(c.chi2 == 1 && c.turight == kf->mvuRight.data());
        // (2) the reference's loop :842-972 replayed on the copy with the checker's results
        KeyFrame* skf = sh.kf(kf);
        int q = 0, want = 0, replaced = 0, added = 0;
        for (size_t i = 0; i < cand.size(); i++) {
            const bool isQuery = q < c.nq && c.qidx[q] == (int)i;
            if (!isQuery) continue;
            const int qq = q++;
            MapPoint* p = sh.mp(cand[i]);
            if (p->isBad() || p->IsInKeyFrame(skf)) continue;
            if (bd[qq] <= 50 && bi[qq] >= 0) {
                MapPoint* in = skf->GetMapPoint(bi[qq]);
                if (in) { if (!in->isBad()) { if (in->Observations() > p->Observations()) p->Replace(in); else in->Replace(p); replaced++; } }
                else { p->AddObservation(skf, bi[qq]); skf->AddMapPoint(p, bi[qq]); added++; }
                want++;
            }
        }
        EXPECT(q == c.nq && n == want && SC(n > 100) && SC(replaced > 10) && SC(added > 10));
        EXPECT(same_state(w, sh));
        w.mps[21]->mbBad = false;
        std::printf("9.%d Fuse(KF, MapPoints, 3): %d queries, %d fused (%d replaced, %d added)\n", variant, c.nq, n, replaced, added);
    }
    // ------------------------------------------------------------ 10. Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)
    {
        Mat Scw = Mat::f32(4, 4);
        to_mat(P0, Scw);
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 4; cc++) Scw.at<float>(r, cc) *= 0.8f;
        std::vector<MapPoint*> pts;
        for (size_t j = 0; j < w.mps.size(); j++) if (!w.mps[j]->mbBad && rng() % 2) pts.push_back(w.mps[j].get());
        std::vector<MapPoint*> repl(pts.size(), nullptr);
        Shadow sh; clone_world(w, sh);
        ORBmatcher m(0.8f, true);
        const int n = m.Fuse(kf0, Scw, pts, 4.f, repl);
        const ORBmatcher::FlatCall& c = m.last();
        std::vector<int32_t> bi, bd;
        check_window(c, kf0->mvInvLevelSigma2, bi, bd);
        KeyFrame* skf = sh.kf(kf0);
        std::vector<int> wantRepl(pts.size(), -1);
        int want = 0;
        const std::set<MapPoint*> inKF = kf0->GetMapPoints();
        for (int q = 0; q < c.nq; q++) {
            EXPECT(q == 0 || c.qidx[q] > c.qidx[q - 1]);
            if (bd[q] <= 50 && bi[q] >= 0) {
                MapPoint* p = sh.mp(pts[c.qidx[q]]);
                MapPoint* in = skf->GetMapPoint(bi[q]);
                if (in) { if (!in->isBad()) wantRepl[c.qidx[q]] = in->id; }
                else { p->AddObservation(skf, bi[q]); skf->AddMapPoint(p, bi[q]); }
                want++;
            }
        }
        EXPECT(n == want && SC(n > 30));
        for (size_t i = 0; i < pts.size(); i++) EXPECT((repl[i] ? repl[i]->id : -1) == wantRepl[i]);
        EXPECT(same_state(w, sh));
        std::printf("10  Fuse(KF, Scw, points, 4, replace): %d queries, %d fused\n", c.nq, n);
    }
    // ------------------------------------------------------------ 11. SearchBySim3
    {
        // the Sim3 between the two keyframes' cameras: x1 = s12 R12 x2 + t12 with s12 = 1 (their true relative pose)
        double R12[9], t12[3];
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) { R12[3 * r + cc] = 0; for (int k = 0; k < 3; k++) R12[3 * r + cc] += P0.R[3 * r + k] * P1.R[3 * cc + k]; }
        for (int r = 0; r < 3; r++) { t12[r] = P0.t[r]; for (int k = 0; k < 3; k++) t12[r] -= R12[3 * r + k] * P1.t[k]; }
        Mat R = Mat::f32(3, 3), t = Mat::f32(3, 1);
        for (int i = 0; i < 9; i++) R.at<float>(i / 3, i % 3) = (float)R12[i];
        for (int i = 0; i < 3; i++) t.at<float>(i, 0) = (float)t12[i];
        std::vector<MapPoint*> m12(kf0->N, nullptr);
        int pre = 0;
        for (int i = 0; i < kf0->N && pre < 25; i++) {   // matches found before (by BoW): excluded from both passes
            MapPoint* p = kf0->mvpMapPoints[i];
            if (p && p->IsInKeyFrame(kf1)) { m12[i] = p; pre++; }
        }
        const std::vector<MapPoint*> before = m12;
        ORBmatcher m(0.75f, true);
        const float s12 = 1.0f;
        const int n = m.SearchBySim3(kf0, kf1, m12, s12, R, t, 7.5f);
        std::vector<int32_t> bi1, bd1, bi2, bd2;
        check_window(m.last(), kf1->mvInvLevelSigma2, bi1, bd1);
        check_window(m.last2(), kf0->mvInvLevelSigma2, bi2, bd2);
        std::vector<int> v1(kf0->N, -1), v2(kf1->N, -1);
        for (int q = 0; q < m.last().nq; q++) if (bd1[q] <= 100 && bi1[q] >= 0) v1[m.last().qidx[q]] = bi1[q];
        for (int q = 0; q < m.last2().nq; q++) if (bd2[q] <= 100 && bi2[q] >= 0) v2[m.last2().qidx[q]] = bi2[q];
        int want = 0;
        for (int i = 0; i < kf0->N; i++) {
            MapPoint* exp = before[i];
            if (v1[i] >= 0 && v2[v1[i]] == i) { exp = kf1->mvpMapPoints[v1[i]]; want++; }
            EXPECT(m12[i] == exp);
        }
        EXPECT(n == want && SC(n > 30) && pre == 25);
        for (int q = 0; q < m.last().nq; q++) {
            EXPECT(!before[m.last().qidx[q]]);
            MapPoint* p = kf0->mvpMapPoints[m.last().qidx[q]];
            double u, v, z; project(P1, &w.X[3 * p->id], u, v, z);
            EXPECT(std::fabs(m.last().q_uvr[3 * q] - u) < 5e-2 && std::fabs(m.last().q_uvr[3 * q + 1] - v) < 5e-2);
        }
        std::printf("11  SearchBySim3: %d + %d queries, %d mutual matches\n", m.last().nq, m.last2().nq, n);
    }
    // ------------------------------------------------------------ static DescriptorDistance
    {
        Mat a = w.mps[1]->GetDescriptor(), b = w.mps[2]->GetDescriptor();
        EXPECT(ORBmatcher::DescriptorDistance(a, b) == orc_descriptor_distance(a.ptr<uint8_t>(0), b.ptr<uint8_t>(0)));
        EXPECT(ORBmatcher::DescriptorDistance(a, a) == 0);
    }
    if (fails) { std::printf("matcher_dropin_gpu: %d FAILURES\n", fails); return 1; }
    std::printf("matcher_dropin_gpu ok\n");
    return 0;
}
