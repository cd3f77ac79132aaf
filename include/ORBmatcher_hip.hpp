// ORBmatcher_hip.hpp -- C++ adapter with the reference's ORBmatcher vocabulary
// (/root/reference/SingleRobotScenario/include/ORBmatcher.h:37-102) over the C ABI, on flat
// arrays.  The reference's member functions walk Frame/KeyFrame/MapPoint objects; a drop-in
// keeps that walking in src/ORBmatcher.cc and calls these members
// with the flattened inputs.  Header-only, C++11, no OpenCV needed.
#pragma once

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <memory>
#include <vector>

#include "orbslamm_hip.h"

namespace iORB_SLAM {

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>) flattened to CSR
struct FlatFeatVec {
    std::vector<uint32_t> node_id;
    std::vector<int32_t> start, idx;
    FlatFeatVec() : start(1, 0) {}
    template <class MapT>
    static FlatFeatVec from(const MapT& fv)
    {
        FlatFeatVec f;
        for (typename MapT::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            f.node_id.push_back((uint32_t)it->first);
            for (size_t k = 0; k < it->second.size(); k++) f.idx.push_back((int32_t)it->second[k]);
            f.start.push_back((int32_t)f.idx.size());
        }
        return f;
    }
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

class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;  // ORBmatcher.cc:37-39

    ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int device = 0)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri)
    {
        if (orbm_create(device, &h_) != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher(HIP): ") + orbx_last_error());
    }
    ~ORBmatcher() { orbm_destroy(h_); }
    ORBmatcher(const ORBmatcher&) = delete;
    ORBmatcher& operator=(const ORBmatcher&) = delete;

    // static int DescriptorDistance(const cv::Mat&, const cv::Mat&)  ORBmatcher.cc:1649.
    // One pair per launch: residual callers only; hot callers use the batched members below.
    int DescriptorDistance(const uint8_t a[32], const uint8_t b[32])
    {
        int32_t d = -1;
        check(orbm_distance_matrix(h_, a, 1, b, 1, &d));
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
        check(orbm_search_by_bow(h_, qdesc, qangle, qvalid, nq, &a, tdesc, tangle, tvalid, nt, &b, mfNNratio, mbCheckOrientation,
                                 outByTrain, match.data(), &n));
        return n;
    }

    // the four SearchByProjection overloads (:45, :1330, :1474, :292) = mode 3, 4, 5, 6
    int SearchByProjection(int mode, int thDist, const float* q_uvr, const int8_t* q_lvl, const uint8_t* qdesc,
                           const float* qangle, const uint8_t* qvalid, const uint8_t* q_obs_pos, int nq,
                           const OrbmGrid& grid, const OrbxKeyPoint* t_keys_un, const uint8_t* tdesc, int nt,
                           std::vector<uint8_t>& t_occ, std::vector<int32_t>& assign)
    {
        OrbmProjParams pp = {mode, mfNNratio, mbCheckOrientation, thDist};
        int n = 0;
        check(orbm_search_by_projection(h_, &pp, q_uvr, q_lvl, qdesc, qangle, qvalid, q_obs_pos, nq, &grid, t_keys_un, tdesc, nt,
                                        t_occ.data(), assign.data(), &n));
        return n;
    }

    // SearchForInitialization :407
    int SearchForInitialization(const float* vbPrevMatched_xy, int windowSize, const OrbxKeyPoint* keys1, const uint8_t* desc1, int n1,
                                const OrbmGrid& grid, const OrbxKeyPoint* keys2, const uint8_t* desc2, int n2,
                                std::vector<int>& vnMatches12)
    {
        vnMatches12.assign(n1, -1);
        int n = 0;
        check(orbm_search_for_initialization(h_, vbPrevMatched_xy, (float)windowSize, keys1, desc1, n1, &grid, keys2, desc2, n2,
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
        check(orbm_search_for_triangulation(h_, k1, d1, skip1, uright1, n1, &a, k2, d2, skip2, uright2, n2, &b, F12, ex, ey,
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
        check(orbm_window_best(h_, q_uvr, q_ur, q_pred, qdesc, qvalid, nq, &grid, t_keys_un, tdesc, t_uright, nt,
                               mvInvLevelSigma2.data(), (int)mvInvLevelSigma2.size(), chi2, bestIdx.data(), bestDist.data()));
    }

    // ---- device-resident frames: tail of Frame::Frame (UndistortKeyPoints + AssignFeaturesToGrid)
    // on the extractor's device output (ORBextractor::lastOnDevice); K = fx, fy, cx, cy; D = mDistCoef
    std::unique_ptr<DeviceFrame> makeFrame(const OrbxKeyPoint* d_keypoints, const uint8_t* d_descriptors, int n,
                                           const float K[4], const float D[5], const OrbmGrid& grid)
    {
        orbm_frame_t* f = nullptr;
        check(orbm_frame_create(h_, d_keypoints, d_descriptors, n, K, D, &grid, &f));
        return std::unique_ptr<DeviceFrame>(new DeviceFrame(f));
    }
    // SearchForInitialization(F1, F2, ...) :407 between two device frames
    int SearchForInitialization(const float* vbPrevMatched_xy, int windowSize, DeviceFrame& F1, DeviceFrame& F2, std::vector<int>& vnMatches12)
    {
        vnMatches12.assign((size_t)F1.size(), -1);
        int n = 0;
        check(orbm_search_for_initialization_frames(h_, vbPrevMatched_xy, (float)windowSize, F1.get(), F2.get(), mfNNratio, mbCheckOrientation,
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
        check(orbm_search_by_projection_frame(h_, &pp, q_uvr, q_lvl, qdesc, qangle, qvalid, q_obs_pos, nq, train.get(), tOcc.data(), assign.data(), &n));
        return n;
    }

    float mfNNratio;
    bool mbCheckOrientation;

protected:
    static void check(int rc) { if (rc != ORBX_OK) throw std::runtime_error(std::string("ORBmatcher(HIP): ") + orbx_last_error()); }
    orbm_t* h_ = nullptr;
};

}  // namespace iORB_SLAM
