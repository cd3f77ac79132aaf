// ORBmatcher_hip.hpp -- the reference's ORBmatcher (/root/reference/SingleRobotScenario/include/ORBmatcher.h:37-102)
// over the C ABI of liborbslamm_hip.so.  Header-only, C++11.
//
//   FlatMatcher     the C ABI with std::vector outputs, on flat arrays.  Holds (mfNNratio, mbCheckOrientation, device)
//                   and NOTHING on the device: the reference builds its matchers as stack temporaries at every call
//                   site (Tracking.cc:639, 809, 914, 1242, 1415, 1454; LocalMapping.cc:215, 483; LoopClosing.cc:245, 603;
//                   MultiMapper.cc:180, 687), so constructing / destroying one touches no HIP API.  Every member call
//                   runs on the calling thread's handle (orbm_thread_handle: queue, scratch and staging block made at
//                   the thread's first search, kept for the thread's life).
//   ORBmatcherT<Frame, KeyFrame, MapPoint>
//                   the drop-in: the reference's eleven member signatures (ORBmatcher.h:48-83).  Every member does
//                   what the reference's loop does around its distance search -- walks the object graph (MapPoint
//                   validity, Observations(), GetDescriptor(), the camera projection, radius and level window),
//                   flattens it, makes ONE device call, and writes the result back (mvpMapPoints[idx] = pMP, rotation
//                   pruning, Replace / AddObservation).  In the reference tree:
//                       typedef iORB_SLAM::ORBmatcherT<Frame, KeyFrame, MapPoint> ORBmatcher;
//                   It is a template so that it compiles (and is tested, tests/cpp/matcher_dropin_gpu.cpp) without
//                   OpenCV: the types only need the members the reference code itself touches (listed at the class).
//
// Third-party arithmetic.  The reference projects with cv::Mat algebra (`Rcw*x3Dw+tcw`, `cv::norm`, `Mat::dot`,
// `-Rcw.t()*tcw`, `sRcw/scw`); namespace cvsem below restates what OpenCV 3.0's matmul.cpp / stat.cpp do for these
// shapes (CV_32F, 3x3 and 3x1) from the published source: UNPINNED like the extractor's OpenCV primitives (DESIGN.md
// section 2).  Everything after the projection -- the candidate windows, the Hamming search, thresholds, ratio tests,
// the sequential "already matched" rules, the rotation histogram -- is pinned by the reference source and runs on the
// device, bit-exact against the CPU checker.
#pragma once

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "orbslamm_hip.h"

namespace iORB_SLAM {

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) flattened to CSR
struct FlatFeatVec {
    std::vector<uint32_t> node_id;
    std::vector<int32_t> start, idx;
    FlatFeatVec() : start(1, 0) {}
    // (re)fill from a DBoW2::FeatureVector; the vectors keep their capacity from call to call
    template <class MapT>
    void assign(const MapT& fv)
    {
        node_id.clear(); idx.clear(); start.clear();
        start.push_back(0);
        for (typename MapT::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            node_id.push_back((uint32_t)it->first);
            for (size_t k = 0; k < it->second.size(); k++) idx.push_back((int32_t)it->second[k]);
            start.push_back((int32_t)idx.size());
        }
    }
    template <class MapT>
    static FlatFeatVec from(const MapT& fv) { FlatFeatVec f; f.assign(fv); return f; }
    OrbmFeatVec view() const
    {
        OrbmFeatVec v;
        v.n_nodes = (int32_t)node_id.size();
        v.node_id = node_id.data(); v.start = start.data(); v.idx = idx.data();
        return v;
    }
};

// A Frame's matcher-side state (mvKeysUn, descriptors, 64x48 grid, FeatureVector) held in HBM: orbm_frame_*.
class DeviceFrame {
public:
    explicit DeviceFrame(orbm_frame_t* f) : f_(f) {}
    ~DeviceFrame() { orbm_frame_destroy(f_); }
    DeviceFrame(const DeviceFrame&) = delete;
    DeviceFrame& operator=(const DeviceFrame&) = delete;
    int size() const { return orbm_frame_size(f_); }
    // mvKeysUn for the host side (pose optimisation, map point creation)
    void keysUn(std::vector<OrbxKeyPoint>& out)
    {
        out.resize((size_t)size());
        if (orbm_frame_download_keys_un(f_, out.data()) != ORBX_OK) throw std::runtime_error(std::string("DeviceFrame: ") + orbx_last_error());
    }
    orbm_frame_t* get() const { return f_; }
private:
    orbm_frame_t* f_;
};

// A set of device-resident frames and the searches Tracking runs per frame over pairs of them, batched: orbm_frameset_*,
// orbm_track_*, orbm_bow_* (include/orbslamm_hip.h).  Slots are filled straight from an extractor's device results
// (the tail of Frame::Frame for the whole batch in one launch); match tables are read in place from pinned host memory.
class FrameSet {
public:
    FrameSet(orbm_t* matcher, int slots, int cap, const float K[4], const float D[5], const OrbmGrid& grid, const float bounds[4],
             const std::vector<float>& scaleFactors)
        : cap_(cap)
    {
        if (orbm_frameset_create(matcher, slots, cap, K, D, &grid, bounds, scaleFactors.data(), (int)scaleFactors.size(), &fs_) != ORBX_OK)
            throw std::runtime_error(std::string("FrameSet: ") + orbx_last_error());
    }
    ~FrameSet() { orbm_frameset_destroy(fs_); }
    FrameSet(const FrameSet&) = delete;
    FrameSet& operator=(const FrameSet&) = delete;
    int capacity() const { return cap_; }
    // the frames of the extractor's last batch into slots slot0, slot0 + 1, .. (ring); no host sync
    void build(int slot0, orbx_t* extractor) { check(orbm_frameset_build_from_extractor(fs_, slot0, extractor)); }
    // mvKeysUn / mDescriptors of a slot for the host side of Tracking
    int download(int slot, std::vector<OrbxKeyPoint>& keysUn, std::vector<uint8_t>& descriptors)
    {
        keysUn.resize((size_t)cap_); descriptors.resize((size_t)cap_ * 32);
        int n = 0;
        check(orbm_frameset_download(fs_, slot, keysUn.data(), descriptors.data(), cap_, &n));
        keysUn.resize((size_t)n); descriptors.resize((size_t)n * 32);
        return n;
    }
    // SearchByProjection(CurrentFrame, LastFrame, th, bMono = true) (ORBmatcher.cc:1330) for every (cur, last) slot pair;
    // asynchronous.  results(): assign[p * capacity() + t] = LastFrame feature held by CurrentFrame feature t, or -1.
    void track(const std::vector<int32_t>& cur, const std::vector<int32_t>& last, float th, float nnratio = 0.9f, bool checkOri = true, int thDist = 100)
    {
        OrbmProjParams pp = {4, nnratio, checkOri ? 1 : 0, thDist};
        check(orbm_track_frames(fs_, &pp, th, cur.data(), last.data(), (int)cur.size()));
    }
    int results(const int32_t*& assign, const int32_t*& nmatches, int back = 0)
    {
        int npairs = 0;
        check(orbm_track_results(fs_, back, &assign, &nmatches, &npairs, nullptr));
        return npairs;
    }
    // Frame::ComputeBoW (Frame.cc:394) for n slots, then SearchByBoW(KeyFrame, Frame) (ORBmatcher.cc:159) for slot pairs
    void computeBoW(orbv_t* vocabulary, int slot0, int n, int levelsup = 4) { check(orbm_frameset_compute_bow(fs_, vocabulary, slot0, n, levelsup)); }
    void searchByBoW(const std::vector<int32_t>& keyFrames, const std::vector<int32_t>& frames, float nnratio = 0.7f, bool checkOri = true)
    {
        check(orbm_bow_frames(fs_, keyFrames.data(), frames.data(), (int)keyFrames.size(), nnratio, checkOri ? 1 : 0));
    }
    int bowResults(const int32_t*& match, const int32_t*& nmatches, int back = 0)
    {
        int npairs = 0;
        check(orbm_bow_results(fs_, back, &match, &nmatches, &npairs, nullptr));
        return npairs;
    }
    orbm_frameset_t* get() const { return fs_; }
private:
    static void check(int rc) { if (rc != ORBX_OK) throw std::runtime_error(std::string("FrameSet: ") + orbx_last_error()); }
    orbm_frameset_t* fs_ = nullptr;
    int cap_ = 0;
};

class FlatMatcher {
public:
    // the calling thread's device handle (never owned by the matcher object; do not orbm_destroy it)
    orbm_t* handle() const
    {
        orbm_t* h = nullptr;
        if (orbm_thread_handle(device_, &h) != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher(HIP): ") + orbx_last_error());
        return h;
    }
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;  // ORBmatcher.cc:37-39

    // ORBmatcher.cc:41: stores the two parameters, like the reference's constructor; no device call
    FlatMatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device) {}
    int device() const { return device_; }
    // Microseconds the calling thread's last member call spent inside the C ABI (upload, kernels, download, wait).  What a
    // drop-in member costs beyond this is the object-graph walk around it -- the reference's own loop head and write-back.
    static double& lastDeviceUs() { static thread_local double us = 0; return us; }
    static double& pendingPrepareUs() { static thread_local double us = 0; return us; }   // a PrepareTrain not yet charged to its search
    struct DeviceClock {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~DeviceClock()
        {
            lastDeviceUs() = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() + pendingPrepareUs();
            pendingPrepareUs() = 0;
        }
    };

    // The train side of the next SearchByProjection goes up and its grid is built while the caller walks its MapPoints
    // (orbm_projection_prepare: asynchronous, a hint -- the search uploads the frame itself if it was handed other arrays).
    void PrepareTrain(const OrbmGrid& grid, const OrbxKeyPoint* t_keys_un, const uint8_t* tdesc, int nt)
    {
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        check(orbm_projection_prepare(handle(), &grid, t_keys_un, tdesc, nt));
        pendingPrepareUs() += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }

    // static int DescriptorDistance(const cv::Mat&, const cv::Mat&)  ORBmatcher.cc:1649.
    // One pair per launch: residual callers only; hot callers use the batched members below.
    int DescriptorDistance(const uint8_t a[32], const uint8_t b[32])
    {
        int32_t d = -1;
        DeviceClock clk_;
        check(orbm_distance_matrix(handle(), a, 1, b, 1, &d));
        return d;
    }

    // SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) :159 (outByTrain) / (KeyFrame*, KeyFrame*, ...) :524
    int SearchByBoW(const uint8_t* qdesc, const float* qangle, const uint8_t* qvalid, int nq, const FlatFeatVec& qfv,
                    const uint8_t* tdesc, const float* tangle, const uint8_t* tvalid, int nt, const FlatFeatVec& tfv,
                    bool outByTrain, std::vector<int32_t>& match)
    {
        match.assign(outByTrain ? nt : nq, -1);
        int n = 0;
        OrbmFeatVec a = qfv.view(), b = tfv.view();
        DeviceClock clk_;
        check(orbm_search_by_bow(handle(), qdesc, qangle, qvalid, nq, &a, tdesc, tangle, tvalid, nt, &b, mfNNratio, mbCheckOrientation,
                                 outByTrain, match.data(), &n));
        return n;
    }

    // the four SearchByProjection overloads (:45, :1330, :1474, :292) = mode 3, 4, 5, 6; q_ur / t_uright: the stereo gate
    // of modes 3 and 4 (both null for mono)
    int SearchByProjection(int mode, int thDist, const float* q_uvr, const int8_t* q_lvl, const uint8_t* qdesc,
                           const float* qangle, const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                           const OrbmGrid& grid, const OrbxKeyPoint* t_keys_un, const uint8_t* tdesc, int nt,
                           std::vector<uint8_t>& t_occ, std::vector<int32_t>& assign,
                           const float* q_ur = nullptr, const float* t_uright = nullptr)
    {
        OrbmProjParams pp = {mode, mfNNratio, mbCheckOrientation, thDist};
        int n = 0;
        DeviceClock clk_;
        check(orbm_search_by_projection_stereo(handle(), &pp, q_uvr, q_ur, q_lvl, qdesc, qangle, qvalid, q_obs_pos, nq, &grid, t_keys_un, tdesc,
                                               t_uright, nt, t_occ.data(), assign.data(), &n));
        return n;
    }

    // SearchForInitialization :407
    int SearchForInitialization(const float* vbPrevMatched_xy, int windowSize, const OrbxKeyPoint* keys1, const uint8_t* desc1, int n1,
                                const OrbmGrid& grid, const OrbxKeyPoint* keys2, const uint8_t* desc2, int n2,
                                std::vector<int>& vnMatches12)
    {
        vnMatches12.assign(n1, -1);
        int n = 0;
        DeviceClock clk_;
        check(orbm_search_for_initialization(handle(), vbPrevMatched_xy, (float)windowSize, keys1, desc1, n1, &grid, keys2, desc2, n2,
                                             mfNNratio, mbCheckOrientation, vnMatches12.data(), &n));
        return n;
    }

    // SearchForTriangulation :659
    int SearchForTriangulation(const OrbxKeyPoint* k1, const uint8_t* d1, const uint8_t* skip1, int n1, const FlatFeatVec& fv1,
                               const OrbxKeyPoint* k2, const uint8_t* d2, const uint8_t* skip2, int n2, const FlatFeatVec& fv2,
                               const float F12[9], float ex, float ey, const std::vector<float>& mvScaleFactors2,
                               const std::vector<float>& mvLevelSigma2_2, bool bOnlyStereo,
                               std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                               const float* uright1 = nullptr, const float* uright2 = nullptr)
    {
        std::vector<int32_t> m12(n1, -1);
        int n = 0;
        OrbmFeatVec a = fv1.view(), b = fv2.view();
        DeviceClock clk_;
        check(orbm_search_for_triangulation(handle(), k1, d1, skip1, uright1, n1, &a, k2, d2, skip2, uright2, n2, &b, F12, ex, ey,
                                            mvScaleFactors2.data(), mvLevelSigma2_2.data(), (int)mvScaleFactors2.size(),
                                            bOnlyStereo, mbCheckOrientation, m12.data(), &n));
        vMatchedPairs.clear();
        vMatchedPairs.reserve(n);
        for (int i = 0; i < n1; i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));  // :816-821
        return n;
    }

    // device part of Fuse (:827, chi2) / Fuse(KF,Scw) (:977) / SearchBySim3 (:1104)
    void WindowBest(const float* q_uvr, const float* q_ur, const int8_t* q_pred, const uint8_t* qdesc, const uint8_t* qvalid, int nq,
                    const OrbmGrid& grid, const OrbxKeyPoint* t_keys_un, const uint8_t* tdesc, const float* t_uright, int nt,
                    const std::vector<float>& mvInvLevelSigma2, bool chi2, std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist)
    {
        bestIdx.assign(nq, -1); bestDist.assign(nq, 256);
        DeviceClock clk_;
        check(orbm_window_best(handle(), q_uvr, q_ur, q_pred, qdesc, qvalid, nq, &grid, t_keys_un, tdesc, t_uright, nt,
                               mvInvLevelSigma2.data(), (int)mvInvLevelSigma2.size(), chi2, bestIdx.data(), bestDist.data()));
    }

    // ---- device-resident frames: tail of Frame::Frame (UndistortKeyPoints + AssignFeaturesToGrid)
    // on the extractor's device output (ORBextractor::lastOnDevice); K = fx, fy, cx, cy; D = mDistCoef
    std::unique_ptr<DeviceFrame> makeFrame(const OrbxKeyPoint* d_keypoints, const uint8_t* d_descriptors, int n,
                                           const float K[4], const float D[5], const OrbmGrid& grid)
    {
        orbm_frame_t* f = nullptr;
        DeviceClock clk_;
        check(orbm_frame_create(handle(), d_keypoints, d_descriptors, n, K, D, &grid, &f));
        return std::unique_ptr<DeviceFrame>(new DeviceFrame(f));
    }
    // SearchForInitialization(F1, F2, ...) :407 between two device frames
    int SearchForInitialization(const float* vbPrevMatched_xy, int windowSize, DeviceFrame& F1, DeviceFrame& F2, std::vector<int>& vnMatches12)
    {
        vnMatches12.assign((size_t)F1.size(), -1);
        int n = 0;
        DeviceClock clk_;
        check(orbm_search_for_initialization_frames(handle(), vbPrevMatched_xy, (float)windowSize, F1.get(), F2.get(), mfNNratio, mbCheckOrientation,
                                                    vnMatches12.data(), &n));
        return n;
    }
    // the four SearchByProjection overloads with a device frame as train side
    int SearchByProjection(int mode, int thDist, const float* q_uvr, const int8_t* q_lvl, const uint8_t* qdesc, const float* qangle,
                           const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq, DeviceFrame& train,
                           std::vector<uint8_t>& tOcc, std::vector<int32_t>& assign)
    {
        OrbmProjParams pp = {mode, mfNNratio, mbCheckOrientation ? 1 : 0, thDist};
        int n = 0;
        DeviceClock clk_;
        check(orbm_search_by_projection_frame(handle(), &pp, q_uvr, q_lvl, qdesc, qangle, qvalid, q_obs_pos, nq, train.get(), tOcc.data(), assign.data(), &n));
        return n;
    }

    float mfNNratio;
    bool mbCheckOrientation;

protected:
    static void check(int rc) { if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher(HIP): ") + orbx_last_error()); }
    int device_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------------
// What OpenCV 3.0 computes for the cv::Mat expressions of ORBmatcher.cc on CV_32F 3x3 / 3x1 operands (matmul.cpp,
// stat.cpp; restated, unpinned).  M is anything with `.template at<float>(r, c)`.
namespace cvsem {
template <class M> inline float at(const M& m, int r, int c) { return m.template at<float>(r, c); }
struct Vec3 { float v[3]; float operator[](int i) const { return v[i]; } };
struct Mat33 { float m[9]; float operator()(int r, int c) const { return m[3 * r + c]; } };
template <class M> inline Mat33 mat33(const M& m, int r0 = 0, int c0 = 0)
{
    Mat33 o;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o.m[3 * r + c] = at(m, r0 + r, c0 + c);
    return o;
}
template <class M> inline Vec3 col3(const M& m, int r0 = 0, int c = 0)
{
    Vec3 o;
    for (int r = 0; r < 3; r++) o.v[r] = at(m, r0 + r, c);
    return o;
}
// `A*b + c` (MatExpr folds it into one gemm(A, b, 1, c, 1)).  gemm's small-matrix branch (flags == 0, 2 <= len <= 4):
// the products are summed in float, left to right; then d = (float)(t * alpha + c * beta) with double alpha = beta = 1.
inline Vec3 mulAdd(const Mat33& A, const Vec3& b, const Vec3& c)
{
    Vec3 o;
    for (int i = 0; i < 3; i++) {
        const float t = A(i, 0) * b[0] + A(i, 1) * b[1] + A(i, 2) * b[2];
        o.v[i] = (float)((double)t * 1.0 + (double)c[i] * 1.0);
    }
    return o;
}
// `A*b` alone (gemm(A, b, 1, noArray(), 0)): same branch, d = (float)(t * alpha)
inline Vec3 mul(const Mat33& A, const Vec3& b)
{
    Vec3 o;
    for (int i = 0; i < 3; i++) o.v[i] = A(i, 0) * b[0] + A(i, 1) * b[1] + A(i, 2) * b[2];
    return o;
}
// `-A.t()*b` (gemm(A, b, -1, noArray(), 0, GEMM_1_T)): flags != 0 takes the generic kernel, GEMMSingleMul<float,double>:
// double accumulation in k order, d = (float)(s * alpha)
inline Vec3 negTransposeMul(const Mat33& A, const Vec3& b)
{
    Vec3 o;
    for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)A(k, i) * (double)b[k];
        o.v[i] = (float)(s * -1.0);
    }
    return o;
}
// `alpha * A` / `A / s` (MatOp_AddEx -> convertTo with a double scale): (float)((double)a * alpha)
inline Mat33 scale(const Mat33& A, double alpha)
{
    Mat33 o;
    for (int i = 0; i < 9; i++) o.m[i] = (float)((double)A.m[i] * alpha);
    return o;
}
inline Vec3 scale(const Vec3& a, double alpha)
{
    Vec3 o;
    for (int i = 0; i < 3; i++) o.v[i] = (float)((double)a[i] * alpha);
    return o;
}
inline Mat33 transpose(const Mat33& A)
{
    Mat33 o;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o.m[3 * r + c] = A(c, r);
    return o;
}
inline Vec3 sub(const Vec3& a, const Vec3& b) { Vec3 o; for (int i = 0; i < 3; i++) o.v[i] = a[i] - b[i]; return o; }
// cv::norm(v) (NORM_L2, CV_32F: normL2_<float, double>) and Mat::dot (dotProd_<float>): double accumulation
inline double norm(const Vec3& a) { double s = 0; for (int i = 0; i < 3; i++) s += (double)a[i] * (double)a[i]; return std::sqrt(s); }
inline double dot(const Vec3& a, const Vec3& b) { double s = 0; for (int i = 0; i < 3; i++) s += (double)a[i] * (double)b[i]; return s; }
// Decompose Scw (ORBmatcher.cc:301-306, :986-991): sRcw, scw = sqrt(row0 . row0), Rcw = sRcw / scw, tcw = Scw.col(3) / scw,
// Ow = -Rcw.t() * tcw
template <class M> inline void decomposeSim3(const M& Scw, Mat33& Rcw, Vec3& tcw, Vec3& Ow)
{
    const Mat33 sR = mat33(Scw);
    Vec3 r0; for (int c = 0; c < 3; c++) r0.v[c] = sR(0, c);
    const float scw = (float)std::sqrt(dot(r0, r0));
    Rcw = scale(sR, 1.0 / (double)scw);
    tcw = scale(col3(Scw, 0, 3), 1.0 / (double)scw);
    Ow = negTransposeMul(Rcw, tcw);
}
}  // namespace cvsem

// ---------------------------------------------------------------------------------------------------------------------
// The drop-in.  Members the three types must have (= what ORBmatcher.cc touches):
//   MapPoint  isBad() Observations() GetDescriptor() GetWorldPos() GetNormal() GetMinDistanceInvariance()
//             GetMaxDistanceInvariance() PredictScale(dist, logScaleFactor) IsInKeyFrame(pKF) GetIndexInKeyFrame(pKF)
//             Replace(pMP) AddObservation(pKF, idx)  mbTrackInView mnTrackScaleLevel mTrackViewCos mTrackProjX/Y/XR
//   Frame     N mvKeys mvKeysUn mvuRight mDescriptors mvpMapPoints mvbOutlier mFeatVec mTcw mb mbf mvScaleFactors
//             mfLogScaleFactor fx fy cx cy mnMinX mnMaxX mnMinY mnMaxY mfGridElementWidthInv mfGridElementHeightInv
//   KeyFrame  N mvKeysUn mvuRight mDescriptors mFeatVec fx fy cx cy mbf mvScaleFactors mvLevelSigma2 mvInvLevelSigma2
//             mfLogScaleFactor mnMinX mnMinY mnGridCols mnGridRows mfGridElementWidthInv mfGridElementHeightInv
//             GetMapPointMatches() GetMapPoints() GetMapPoint(idx) AddMapPoint(pMP, idx) IsInImage(u, v)
//             GetRotation() GetTranslation() GetCameraCenter()
// Keypoints are cv::KeyPoint-layout records (28 bytes); matrices anything with at<float>(r, c) (3x1: at<float>(r, 0)),
// descriptor matrices additionally ptr<unsigned char>(row).
template <class Frame, class KeyFrame, class MapPoint>
class ORBmatcherT {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;  // ORBmatcher.cc:37-39
    static const int FRAME_GRID_COLS = 64, FRAME_GRID_ROWS = 48;      // Frame.h:36-37

    // ORBmatcher.cc:41-43: two parameters, nothing else -- the object is a stack temporary in the reference
    ORBmatcherT(float nnratio = 0.6f, bool checkOri = true, int device = 0)
        : flat_(nnratio, checkOri, device) {}

    // What the last member call OF THIS THREAD handed to the device and got back (flattened arrays + the query -> object
    // index map).  The storage is per thread, not per matcher (the matcher is a temporary; its flattening scratch must
    // outlive it to be reused): the vectors keep their capacity from call to call, whichever matcher object makes the
    // call.  Tests feed them to the CPU checker right after a call.
    struct FlatCall {
        std::string fn;
        int mode = 0, thDist = 0, nq = 0, nt = 0, chi2 = 0;
        OrbmGrid grid{};
        std::vector<float> q_uvr, q_ur, qangle, q_xy;
        std::vector<int8_t> q_lvl, q_pred;
        std::vector<uint8_t> qdesc, qvalid, qobs, tvalid, tocc_in, tocc, skip1, skip2;
        std::vector<int32_t> qidx;       // query q -> index into the caller's container (vpMapPoints, LastFrame, ...)
        std::vector<int32_t> assign, match, bestIdx, bestDist;
        std::vector<int> m12;
        FlatFeatVec qfv, tfv;
        const OrbxKeyPoint* tkeys = nullptr; const OrbxKeyPoint* qkeys = nullptr;
        std::vector<uint8_t> tdesc_store, qdesc_store;  // only when the caller's descriptor matrix is not continuous
        const uint8_t* tdesc = nullptr; const float* turight = nullptr;
        const uint8_t* qdescBlock = nullptr;             // query side given as a whole descriptor matrix (BoW, initialisation, triangulation)
        std::vector<float> tangle, F12; float ex = 0, ey = 0;
        int nmatches = 0;
    };
    // resolved at call time: a matcher built on one thread and used on another reads and writes the CALLING thread's scratch,
    // and the object stays copyable / assignable like the reference's (it holds no reference)
    static FlatCall& last() { return threadScratch(0); }
    // second pass of SearchBySim3 (KF2's points into KF1)
    static FlatCall& last2() { return threadScratch(1); }

    // static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b)   ORBmatcher.h:45, ORBmatcher.cc:1649-1665
    // ONE pair: eight xor + popcount on the host.  The reference calls this per pair from loops the device entries
    // replace wholesale (MapPoint::ComputeDistinctiveDescriptors -> orbm_distinctive_descriptors, Frame::
    // ComputeStereoMatches -> orbx_compute_stereo_matches); a caller that keeps such a loop must not pay two copies, a
    // launch and a sync per pair, nor depend on which device some other thread's handle lives on.  (It is the
    // reference's own arithmetic on 8 native-endian 32-bit words, no third-party code involved.)
    template <class Mat> static int DescriptorDistance(const Mat& a, const Mat& b)
    {
        const unsigned char* pa = a.template ptr<unsigned char>(0);
        const unsigned char* pb = b.template ptr<unsigned char>(0);
        int dist = 0;
        for (int i = 0; i < 8; i++) {
            uint32_t x, y;
            std::memcpy(&x, pa + 4 * i, 4);
            std::memcpy(&y, pb + 4 * i, 4);
            dist += __builtin_popcount(x ^ y);
        }
        return dist;
    }

    // ---------------------------------------------------------------- SearchByProjection(Frame&, vector<MapPoint*>&, th)  :45-129
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3)
    {
        FlatCall& c = begin("SearchByProjection(Frame,MapPoints)", 3, TH_HIGH);
        trainFromFrame(c, F, /*blockOnObservations=*/true);   // first: the frame goes up and its grid is built under the walk below
        flat_.PrepareTrain(c.grid, c.tkeys, c.tdesc, c.nt);
        const bool bFactor = th != 1.0;
        bool stereo = false;
        for (size_t i = 0; i < F.mvuRight.size() && !stereo; i++) stereo = F.mvuRight[i] > 0;
        for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
            MapPoint* pMP = vpMapPoints[iMP];
            if (!pMP->mbTrackInView) continue;
            if (pMP->isBad()) continue;
            const int nPredictedLevel = pMP->mnTrackScaleLevel;
            if (nPredictedLevel < 0 || nPredictedLevel >= (int)F.mvScaleFactors.size()) continue;  // the reference would index out of range
            float r = RadiusByViewingCos(pMP->mTrackViewCos);
            if (bFactor) r *= th;
            pushQuery(c, pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel,
                      pMP, 0.f, (int)iMP);
            if (stereo) c.q_ur.push_back(pMP->mTrackProjXR);
        }
        runProjection(c, stereo ? F.mvuRight.data() : nullptr);
        for (int t = 0; t < c.nt; t++)
            if (c.assign[t] >= 0) F.mvpMapPoints[t] = vpMapPoints[c.qidx[c.assign[t]]];
        return c.nmatches;
    }

    // ---------------------------------------------------------------- SearchByProjection(Frame& Cur, const Frame& Last, th, bMono)  :1330-1472
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono)
    {
        using namespace cvsem;
        FlatCall& c = begin("SearchByProjection(Frame,Frame)", 4, TH_HIGH);
        trainFromFrame(c, CurrentFrame, /*blockOnObservations=*/true);   // (under the walk of LastFrame's MapPoints, see mode 3)
        flat_.PrepareTrain(c.grid, c.tkeys, c.tdesc, c.nt);
        const Mat33 Rcw = mat33(CurrentFrame.mTcw);
        const Vec3 tcw = col3(CurrentFrame.mTcw, 0, 3);
        const Vec3 twc = negTransposeMul(Rcw, tcw);
        const Mat33 Rlw = mat33(LastFrame.mTcw);
        const Vec3 tlw = col3(LastFrame.mTcw, 0, 3);
        const Vec3 tlc = mulAdd(Rlw, twc, tlw);
        const bool bForward = tlc[2] > CurrentFrame.mb && !bMono;
        const bool bBackward = -tlc[2] > CurrentFrame.mb && !bMono;
        bool stereo = false;
        for (size_t i = 0; i < CurrentFrame.mvuRight.size() && !stereo; i++) stereo = CurrentFrame.mvuRight[i] > 0;
        const int nLevels = (int)CurrentFrame.mvScaleFactors.size();
        for (int i = 0; i < LastFrame.N; i++) {
            MapPoint* pMP = LastFrame.mvpMapPoints[i];
            if (!pMP || LastFrame.mvbOutlier[i]) continue;
            const Vec3 x3Dc = mulAdd(Rcw, col3(pMP->GetWorldPos()), tcw);
            const float xc = x3Dc[0], yc = x3Dc[1];
            const float invzc = (float)(1.0 / (double)x3Dc[2]);
            if (invzc < 0) continue;
            const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
            const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
            if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
            if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
            const int nLastOctave = LastFrame.mvKeys[i].octave;
            if (nLastOctave < 0 || nLastOctave >= nLevels) continue;
            const float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
            int minL, maxL;
            if (bForward) { minL = nLastOctave; maxL = -1; }
            else if (bBackward) { minL = 0; maxL = nLastOctave; }
            else { minL = nLastOctave - 1; maxL = nLastOctave + 1; }
            pushQuery(c, u, v, radius, minL, maxL, pMP, LastFrame.mvKeysUn[i].angle, i);
            if (stereo) c.q_ur.push_back(u - CurrentFrame.mbf * invzc);
        }
        runProjection(c, stereo ? CurrentFrame.mvuRight.data() : nullptr);
        for (int t = 0; t < c.nt; t++) {
            if (c.assign[t] >= 0) CurrentFrame.mvpMapPoints[t] = LastFrame.mvpMapPoints[c.qidx[c.assign[t]]];
            else if (c.assign[t] == -1) CurrentFrame.mvpMapPoints[t] = static_cast<MapPoint*>(NULL);  // claimed, then pruned (:1462)
        }
        return c.nmatches;
    }

    // ---------------------------------------------------------------- SearchByProjection(Frame&, KeyFrame*, set<MapPoint*>&, th, ORBdist)  :1474-1601
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist)
    {
        using namespace cvsem;
        FlatCall& c = begin("SearchByProjection(Frame,KeyFrame)", 5, ORBdist);
        trainFromFrame(c, CurrentFrame, /*blockOnObservations=*/false);
        flat_.PrepareTrain(c.grid, c.tkeys, c.tdesc, c.nt);
        const Mat33 Rcw = mat33(CurrentFrame.mTcw);
        const Vec3 tcw = col3(CurrentFrame.mTcw, 0, 3);
        const Vec3 Ow = negTransposeMul(Rcw, tcw);
        const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
        const int nLevels = (int)CurrentFrame.mvScaleFactors.size();
        for (size_t i = 0; i < vpMPs.size(); i++) {
            MapPoint* pMP = vpMPs[i];
            if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
            const Vec3 x3Dw = col3(pMP->GetWorldPos());
            const Vec3 x3Dc = mulAdd(Rcw, x3Dw, tcw);
            const float xc = x3Dc[0], yc = x3Dc[1];
            const float invzc = (float)(1.0 / (double)x3Dc[2]);
            const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
            const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
            if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
            if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
            const float dist3D = (float)norm(sub(x3Dw, Ow));
            const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, CurrentFrame.mfLogScaleFactor);
            if (nPredictedLevel < 0 || nPredictedLevel >= nLevels) continue;
            const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
            pushQuery(c, u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, pMP, pKF->mvKeysUn[i].angle, (int)i);
        }
        runProjection(c, nullptr);
        for (int t = 0; t < c.nt; t++) {
            if (c.assign[t] >= 0) CurrentFrame.mvpMapPoints[t] = vpMPs[c.qidx[c.assign[t]]];
            else if (c.assign[t] == -1) CurrentFrame.mvpMapPoints[t] = static_cast<MapPoint*>(NULL);
        }
        return c.nmatches;
    }

    // ---------------------------------------------------------------- SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)  :292-405
    template <class Mat>
    int SearchByProjection(KeyFrame* pKF, Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th)
    {
        FlatCall& c = begin("SearchByProjection(KeyFrame,Scw)", 6, TH_LOW);
        std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
        spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
        trainFromKeyFrame(c, pKF);
        flat_.PrepareTrain(c.grid, c.tkeys, c.tdesc, c.nt);
        projectIntoKeyFrame(c, pKF, Scw, vpPoints, spAlreadyFound, (float)th, /*windowMode=*/false);
        c.tocc_in.assign(c.nt, 0);
        for (int t = 0; t < c.nt; t++) c.tocc_in[t] = vpMatched[t] != NULL;
        runProjection(c, nullptr);
        for (int t = 0; t < c.nt; t++)
            if (c.assign[t] >= 0) vpMatched[t] = vpPoints[c.qidx[c.assign[t]]];
        return c.nmatches;
    }

    // ---------------------------------------------------------------- SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)  :159-290
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches)
    {
        FlatCall& c = begin("SearchByBoW(KeyFrame,Frame)", 1, TH_LOW);
        const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
        c.nq = (int)vpMapPointsKF.size(); c.nt = F.N;
        c.qvalid.assign(c.nq, 0); c.qangle.resize(c.nq);
        for (int i = 0; i < c.nq; i++) {
            MapPoint* pMP = vpMapPointsKF[i];
            c.qvalid[i] = pMP && !pMP->isBad();
            c.qangle[i] = pKF->mvKeysUn[i].angle;
        }
        c.tangle.resize(c.nt);
        for (int t = 0; t < c.nt; t++) c.tangle[t] = F.mvKeys[t].angle;  // :238: the Frame side reads mvKeys, not mvKeysUn
        c.qfv.assign(pKF->mFeatVec); c.tfv.assign(F.mFeatVec);
        const uint8_t* qd = c.qdescBlock = descPtr(pKF->mDescriptors, c.nq, c.qdesc_store);
        c.tdesc = descPtr(F.mDescriptors, c.nt, c.tdesc_store);
        c.nmatches = flat_.SearchByBoW(qd, c.qangle.data(), c.qvalid.data(), c.nq, c.qfv, c.tdesc, c.tangle.data(), nullptr, c.nt, c.tfv,
                                       /*outByTrain=*/true, c.match);
        for (int t = 0; t < c.nt; t++)
            if (c.match[t] >= 0) vpMapPointMatches[t] = vpMapPointsKF[c.match[t]];
        return c.nmatches;
    }

    // ---------------------------------------------------------------- SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&)  :524-657
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12)
    {
        FlatCall& c = begin("SearchByBoW(KeyFrame,KeyFrame)", 2, TH_LOW);
        const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));
        c.nq = (int)vpMapPoints1.size(); c.nt = (int)vpMapPoints2.size();
        c.qvalid.assign(c.nq, 0); c.tvalid.assign(c.nt, 0); c.qangle.resize(c.nq); c.tangle.resize(c.nt);
        for (int i = 0; i < c.nq; i++) { MapPoint* p = vpMapPoints1[i]; c.qvalid[i] = p && !p->isBad(); c.qangle[i] = pKF1->mvKeysUn[i].angle; }
        for (int i = 0; i < c.nt; i++) { MapPoint* p = vpMapPoints2[i]; c.tvalid[i] = p && !p->isBad(); c.tangle[i] = pKF2->mvKeysUn[i].angle; }
        c.qfv.assign(pKF1->mFeatVec); c.tfv.assign(pKF2->mFeatVec);
        const uint8_t* qd = c.qdescBlock = descPtr(pKF1->mDescriptors, c.nq, c.qdesc_store);
        c.tdesc = descPtr(pKF2->mDescriptors, c.nt, c.tdesc_store);
        c.nmatches = flat_.SearchByBoW(qd, c.qangle.data(), c.qvalid.data(), c.nq, c.qfv, c.tdesc, c.tangle.data(), c.tvalid.data(), c.nt, c.tfv,
                                       /*outByTrain=*/false, c.match);
        for (int i = 0; i < c.nq; i++)
            if (c.match[i] >= 0) vpMatches12[i] = vpMapPoints2[c.match[i]];
        return c.nmatches;
    }

    // ---------------------------------------------------------------- SearchForInitialization  :407-522
    template <class Point2f>
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10)
    {
        FlatCall& c = begin("SearchForInitialization", 7, TH_LOW);
        c.nq = (int)F1.mvKeysUn.size(); c.nt = (int)F2.mvKeysUn.size();
        c.q_xy.resize(2 * (size_t)c.nq);
        for (int i = 0; i < c.nq; i++) { c.q_xy[2 * i] = vbPrevMatched[i].x; c.q_xy[2 * i + 1] = vbPrevMatched[i].y; }
        c.grid = frameGrid(F2);
        c.qkeys = keys(F1.mvKeysUn); c.tkeys = keys(F2.mvKeysUn);
        const uint8_t* qd = descPtr(F1.mDescriptors, c.nq, c.qdesc_store);
        c.tdesc = descPtr(F2.mDescriptors, c.nt, c.tdesc_store);
        c.nmatches = flat_.SearchForInitialization(c.q_xy.data(), windowSize, c.qkeys, qd, c.nq, c.grid, c.tkeys, c.tdesc, c.nt, c.m12);
        vnMatches12 = c.m12;
        for (size_t i1 = 0; i1 < vnMatches12.size(); i1++)  // :517-519
            if (vnMatches12[i1] >= 0) vbPrevMatched[i1] = F2.mvKeysUn[vnMatches12[i1]].pt;
        return c.nmatches;
    }

    // ---------------------------------------------------------------- SearchForTriangulation  :659-825
    template <class Mat>
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs, const bool bOnlyStereo)
    {
        using namespace cvsem;
        FlatCall& c = begin("SearchForTriangulation", 8, TH_LOW);
        // epipole in the second image (:666-672)
        const Vec3 C2 = mulAdd(mat33(pKF2->GetRotation()), col3(pKF1->GetCameraCenter()), col3(pKF2->GetTranslation()));
        const float invz = 1.0f / C2[2];
        c.ex = pKF2->fx * C2[0] * invz + pKF2->cx;
        c.ey = pKF2->fy * C2[1] * invz + pKF2->cy;
        c.nq = pKF1->N; c.nt = pKF2->N;
        c.skip1.assign(c.nq, 0); c.skip2.assign(c.nt, 0);
        for (int i = 0; i < c.nq; i++) c.skip1[i] = pKF1->GetMapPoint(i) != NULL;
        for (int i = 0; i < c.nt; i++) c.skip2[i] = pKF2->GetMapPoint(i) != NULL;
        c.F12.resize(9);
        for (int r = 0; r < 3; r++) for (int k = 0; k < 3; k++) c.F12[3 * r + k] = at(F12, r, k);
        c.qfv.assign(pKF1->mFeatVec); c.tfv.assign(pKF2->mFeatVec);
        c.qkeys = keys(pKF1->mvKeysUn); c.tkeys = keys(pKF2->mvKeysUn);
        const uint8_t* qd = descPtr(pKF1->mDescriptors, c.nq, c.qdesc_store);
        c.tdesc = descPtr(pKF2->mDescriptors, c.nt, c.tdesc_store);
        c.nmatches = flat_.SearchForTriangulation(c.qkeys, qd, c.skip1.data(), c.nq, c.qfv, c.tkeys, c.tdesc, c.skip2.data(), c.nt, c.tfv,
                                                  c.F12.data(), c.ex, c.ey, pKF2->mvScaleFactors, pKF2->mvLevelSigma2, bOnlyStereo, vMatchedPairs,
                                                  pKF1->mvuRight.data(), pKF2->mvuRight.data());
        return c.nmatches;
    }

    // ---------------------------------------------------------------- Fuse(KeyFrame*, vector<MapPoint*>&, th)  :827-975
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0)
    {
        using namespace cvsem;
        FlatCall& c = begin("Fuse(KeyFrame,MapPoints)", 9, TH_LOW);
        c.chi2 = 1;
        const Mat33 Rcw = mat33(pKF->GetRotation());
        const Vec3 tcw = col3(pKF->GetTranslation()), Ow = col3(pKF->GetCameraCenter());
        const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy, bf = pKF->mbf;
        const int nLevels = (int)pKF->mvScaleFactors.size();
        const int nMPs = (int)vpMapPoints.size();
        for (int i = 0; i < nMPs; i++) {
            MapPoint* pMP = vpMapPoints[i];
            if (!pMP) continue;
            // isBad() / IsInKeyFrame(pKF) (:849) can change while the reference's loop runs (Replace, AddObservation): they
            // are evaluated in the sequential pass below; the search of a point does not depend on them
            const Vec3 p3Dw = col3(pMP->GetWorldPos());
            const Vec3 p3Dc = mulAdd(Rcw, p3Dw, tcw);
            if (p3Dc[2] < 0.0f) continue;
            const float invz = 1 / p3Dc[2];
            const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
            const float u = fx * x + cx, v = fy * y + cy;
            if (!pKF->IsInImage(u, v)) continue;
            const float ur = u - bf * invz;
            const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
            const Vec3 PO = sub(p3Dw, Ow);
            const float dist3D = (float)norm(PO);
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            if (dot(PO, col3(pMP->GetNormal())) < 0.5 * dist3D) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, pKF->mfLogScaleFactor);
            if (nPredictedLevel < 0 || nPredictedLevel >= nLevels) continue;
            const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
            pushWindowQuery(c, u, v, radius, nPredictedLevel, pMP, i);
            c.q_ur.push_back(ur);
        }
        trainFromKeyFrame(c, pKF);
        c.turight = pKF->mvuRight.data();
        flat_.WindowBest(c.q_uvr.data(), c.q_ur.data(), c.q_pred.data(), c.qdesc.data(), nullptr, c.nq, c.grid, c.tkeys, c.tdesc, c.turight, c.nt,
                         pKF->mvInvLevelSigma2, true, c.bestIdx, c.bestDist);
        int nFused = 0;
        for (int q = 0; q < c.nq; q++) {  // the reference's order: ascending i
            MapPoint* pMP = vpMapPoints[c.qidx[q]];
            if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
            if (c.bestDist[q] <= TH_LOW && c.bestIdx[q] >= 0) {  // :949-968
                const int bestIdx = c.bestIdx[q];
                MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
                if (pMPinKF) {
                    if (!pMPinKF->isBad()) {
                        if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                        else pMPinKF->Replace(pMP);
                    }
                } else {
                    pMP->AddObservation(pKF, bestIdx);
                    pKF->AddMapPoint(pMP, bestIdx);
                }
                nFused++;
            }
        }
        c.nmatches = nFused;
        return nFused;
    }

    // ---------------------------------------------------------------- Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)  :977-1102
    template <class Mat>
    int Fuse(KeyFrame* pKF, Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint)
    {
        FlatCall& c = begin("Fuse(KeyFrame,Scw)", 10, TH_LOW);
        const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
        projectIntoKeyFrame(c, pKF, Scw, vpPoints, spAlreadyFound, th, /*windowMode=*/true);
        trainFromKeyFrame(c, pKF);
        flat_.WindowBest(c.q_uvr.data(), nullptr, c.q_pred.data(), c.qdesc.data(), nullptr, c.nq, c.grid, c.tkeys, c.tdesc, nullptr, c.nt,
                         pKF->mvInvLevelSigma2, false, c.bestIdx, c.bestDist);
        int nFused = 0;
        for (int q = 0; q < c.nq; q++) {
            if (c.bestDist[q] <= TH_LOW && c.bestIdx[q] >= 0) {  // :1083-1096
                const int iMP = c.qidx[q], bestIdx = c.bestIdx[q];
                MapPoint* pMP = vpPoints[iMP];
                MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
                if (pMPinKF) {
                    if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
                } else {
                    pMP->AddObservation(pKF, bestIdx);
                    pKF->AddMapPoint(pMP, bestIdx);
                }
                nFused++;
            }
        }
        c.nmatches = nFused;
        return nFused;
    }

    // ---------------------------------------------------------------- SearchBySim3  :1104-1328
    template <class Mat>
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const Mat& R12, const Mat& t12, const float th)
    {
        using namespace cvsem;
        const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;  // (sic: KF1's intrinsics for both directions, :1107-1110)
        const Mat33 R1w = mat33(pKF1->GetRotation()), R2w = mat33(pKF2->GetRotation());
        const Vec3 t1w = col3(pKF1->GetTranslation()), t2w = col3(pKF2->GetTranslation());
        const Mat33 R12m = mat33(R12);
        const Vec3 t12v = col3(t12);
        const Mat33 sR12 = scale(R12m, (double)s12);                 // s12*R12
        const Mat33 sR21 = scale(transpose(R12m), 1.0 / s12);        // (1.0/s12)*R12.t()
        const Vec3 t21 = scale(mul(sR21, t12v), -1.0);               // -sR21*t12: gemm(sR21, t12, -1) in the small-matrix branch
        const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
        const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
        std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
        for (int i = 0; i < N1; i++) {
            MapPoint* pMP = vpMatches12[i];
            if (pMP) {
                vbAlreadyMatched1[i] = true;
                const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
                if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
            }
        }
        // one direction: points of `from` (camera Rfw, tfw) moved by (sR, t) into `to` and searched there
        auto pass = [&](FlatCall& c, const char* name, const std::vector<MapPoint*>& pts, const std::vector<bool>& already,
                        const Mat33& Rfw, const Vec3& tfw, const Mat33& sR, const Vec3& t, KeyFrame* to, std::vector<int>& vnMatch) {
            reset(c, name, 11, TH_HIGH);
            const int nLevels = (int)to->mvScaleFactors.size();
            for (int i = 0; i < (int)pts.size(); i++) {
                MapPoint* pMP = pts[i];
                if (!pMP || already[i]) continue;
                if (pMP->isBad()) continue;
                const Vec3 pf = mulAdd(Rfw, col3(pMP->GetWorldPos()), tfw);
                const Vec3 pt = mulAdd(sR, pf, t);
                if (pt[2] < 0.0) continue;
                const float invz = (float)(1.0 / (double)pt[2]);
                const float x = pt[0] * invz, y = pt[1] * invz;
                const float u = fx * x + cx, v = fy * y + cy;
                if (!to->IsInImage(u, v)) continue;
                const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
                const float dist3D = (float)norm(pt);
                if (dist3D < minDistance || dist3D > maxDistance) continue;
                const int nPredictedLevel = pMP->PredictScale(dist3D, to->mfLogScaleFactor);
                if (nPredictedLevel < 0 || nPredictedLevel >= nLevels) continue;
                const float radius = th * to->mvScaleFactors[nPredictedLevel];
                pushWindowQuery(c, u, v, radius, nPredictedLevel, pMP, i);
            }
            trainFromKeyFrame(c, to);
            flat_.WindowBest(c.q_uvr.data(), nullptr, c.q_pred.data(), c.qdesc.data(), nullptr, c.nq, c.grid, c.tkeys, c.tdesc, nullptr, c.nt,
                             to->mvInvLevelSigma2, false, c.bestIdx, c.bestDist);
            vnMatch.assign(pts.size(), -1);
            for (int q = 0; q < c.nq; q++)
                if (c.bestDist[q] <= TH_HIGH && c.bestIdx[q] >= 0) vnMatch[c.qidx[q]] = c.bestIdx[q];
        };
        std::vector<int> vnMatch1, vnMatch2;
        pass(last(), "SearchBySim3(1->2)", vpMapPoints1, vbAlreadyMatched1, R1w, t1w, sR21, t21, pKF2, vnMatch1);
        pass(last2(), "SearchBySim3(2->1)", vpMapPoints2, vbAlreadyMatched2, R2w, t2w, sR12, t12v, pKF1, vnMatch2);
        int nFound = 0;  // :1306-1322
        for (int i1 = 0; i1 < N1; i1++) {
            const int idx2 = vnMatch1[i1];
            if (idx2 >= 0 && vnMatch2[idx2] == i1) { vpMatches12[i1] = vpMapPoints2[idx2]; nFound++; }
        }
        last().nmatches = nFound;
        return nFound;
    }

    FlatMatcher& flat() { return flat_; }

protected:
    static float RadiusByViewingCos(const float& viewCos) { return viewCos > 0.998 ? 2.5f : 4.0f; }  // :131-137
    static FlatCall& threadScratch(int which) { static thread_local FlatCall c[2]; return c[which]; }

    FlatMatcher flat_;

    FlatCall& begin(const char* fn, int mode, int thDist) { FlatCall& c = last(); reset(c, fn, mode, thDist); return c; }
    static void reset(FlatCall& c, const char* fn, int mode, int thDist)
    {
        c.fn = fn; c.mode = mode; c.thDist = thDist; c.nq = c.nt = 0; c.chi2 = 0; c.nmatches = 0;
        c.q_uvr.clear(); c.q_ur.clear(); c.qangle.clear(); c.q_xy.clear(); c.q_lvl.clear(); c.q_pred.clear(); c.qdesc.clear();
        c.qvalid.clear(); c.qobs.clear(); c.tvalid.clear(); c.tocc_in.clear(); c.qidx.clear(); c.tangle.clear();
        c.tkeys = c.qkeys = nullptr; c.tdesc = nullptr; c.turight = nullptr;
    }
    template <class KP> static const OrbxKeyPoint* keys(const std::vector<KP>& v)
    {
        static_assert(sizeof(KP) == sizeof(OrbxKeyPoint), "keypoints must have cv::KeyPoint's 28-byte layout");
        return reinterpret_cast<const OrbxKeyPoint*>(v.data());
    }
    // N x 32 descriptor matrix as one block (cv::Mat rows are contiguous unless the Mat is a ROI)
    template <class Mat> static const uint8_t* descPtr(const Mat& m, int n, std::vector<uint8_t>& store)
    {
        if (n == 0) return nullptr;
        if (m.isContinuous()) return m.template ptr<unsigned char>(0);
        store.resize((size_t)n * 32);
        for (int i = 0; i < n; i++) std::memcpy(&store[(size_t)i * 32], m.template ptr<unsigned char>(i), 32);
        return store.data();
    }
    void pushDescriptor(FlatCall& c, MapPoint* pMP)
    {
        const auto d = pMP->GetDescriptor();  // a clone taken under the MapPoint's mutex, as in the reference
        const unsigned char* p = d.template ptr<unsigned char>(0);
        c.qdesc.insert(c.qdesc.end(), p, p + 32);
    }
    void pushQuery(FlatCall& c, float u, float v, float r, int minL, int maxL, MapPoint* pMP, float angle, int idx)
    {
        c.q_uvr.push_back(u); c.q_uvr.push_back(v); c.q_uvr.push_back(r);
        c.q_lvl.push_back((int8_t)minL); c.q_lvl.push_back((int8_t)maxL);
        pushDescriptor(c, pMP);
        c.qangle.push_back(angle);
        c.qobs.push_back(pMP->Observations() > 0);
        c.qidx.push_back(idx);
        c.nq++;
    }
    void pushWindowQuery(FlatCall& c, float u, float v, float r, int pred, MapPoint* pMP, int idx)
    {
        c.q_uvr.push_back(u); c.q_uvr.push_back(v); c.q_uvr.push_back(r);
        c.q_pred.push_back((int8_t)pred);
        pushDescriptor(c, pMP);
        c.qidx.push_back(idx);
        c.nq++;
    }
    static OrbmGrid frameGrid(const Frame& F)
    {
        OrbmGrid g = {F.mnMinX, F.mnMinY, F.mfGridElementWidthInv, F.mfGridElementHeightInv, FRAME_GRID_COLS, FRAME_GRID_ROWS};
        return g;
    }
    // train side = a Frame; a feature is skipped when it holds a MapPoint (modes 5) / one with observations (modes 3, 4)
    void trainFromFrame(FlatCall& c, Frame& F, bool blockOnObservations)
    {
        c.nt = F.N;
        c.grid = frameGrid(F);
        c.tkeys = keys(F.mvKeysUn);
        c.tdesc = descPtr(F.mDescriptors, c.nt, c.tdesc_store);
        c.tocc_in.assign(c.nt, 0);
        for (int t = 0; t < c.nt; t++) {
            MapPoint* p = F.mvpMapPoints[t];
            c.tocc_in[t] = p && (!blockOnObservations || p->Observations() > 0);
        }
    }
    void trainFromKeyFrame(FlatCall& c, KeyFrame* pKF)
    {
        c.nt = pKF->N;
        OrbmGrid g = {(float)pKF->mnMinX, (float)pKF->mnMinY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv, pKF->mnGridCols, pKF->mnGridRows};
        c.grid = g;
        c.tkeys = keys(pKF->mvKeysUn);
        c.tdesc = descPtr(pKF->mDescriptors, c.nt, c.tdesc_store);
    }
    void runProjection(FlatCall& c, const float* t_uright)
    {
        c.turight = t_uright;
        c.tocc = c.tocc_in;
        c.assign.assign(c.nt, -2);  // -2 = never touched; the device leaves -1 where a claim was pruned by the rotation check
        c.nmatches = flat_.SearchByProjection(c.mode, c.thDist, c.q_uvr.data(), c.q_lvl.data(), c.qdesc.data(), c.qangle.data(), nullptr,
                                              c.qobs.data(), c.nq, c.grid, c.tkeys, c.tdesc, c.nt, c.tocc, c.assign,
                                              t_uright ? c.q_ur.data() : nullptr, t_uright);
    }
    // the projection block shared by SearchByProjection(KF, Scw, ...) :311-363 and Fuse(KF, Scw, ...) :1003-1051
    template <class Mat>
    void projectIntoKeyFrame(FlatCall& c, KeyFrame* pKF, const Mat& Scw, const std::vector<MapPoint*>& vpPoints,
                             const std::set<MapPoint*>& spAlreadyFound, float th, bool windowMode)
    {
        using namespace cvsem;
        const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;
        Mat33 Rcw; Vec3 tcw, Ow;
        decomposeSim3(Scw, Rcw, tcw, Ow);
        const int nLevels = (int)pKF->mvScaleFactors.size();
        for (int iMP = 0, iend = (int)vpPoints.size(); iMP < iend; iMP++) {
            MapPoint* pMP = vpPoints[iMP];
            if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
            const Vec3 p3Dw = col3(pMP->GetWorldPos());
            const Vec3 p3Dc = mulAdd(Rcw, p3Dw, tcw);
            if (p3Dc[2] < 0.0) continue;
            // :326 `1/z` (float) in SearchByProjection, :1020 `1.0/z` (double, rounded) in Fuse
            const float invz = windowMode ? (float)(1.0 / (double)p3Dc[2]) : 1 / p3Dc[2];
            const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
            const float u = fx * x + cx, v = fy * y + cy;
            if (!pKF->IsInImage(u, v)) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
            const Vec3 PO = sub(p3Dw, Ow);
            const float dist = (float)norm(PO);
            if (dist < minDistance || dist > maxDistance) continue;
            if (dot(PO, col3(pMP->GetNormal())) < 0.5 * dist) continue;
            const int nPredictedLevel = pMP->PredictScale(dist, pKF->mfLogScaleFactor);
            if (nPredictedLevel < 0 || nPredictedLevel >= nLevels) continue;
            const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
            if (windowMode) pushWindowQuery(c, u, v, radius, nPredictedLevel, pMP, iMP);
            else pushQuery(c, u, v, radius, nPredictedLevel - 1, nPredictedLevel, pMP, 0.f, iMP);
        }
    }
};

}  // namespace iORB_SLAM
