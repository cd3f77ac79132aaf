// ORBextractor_hip.hpp -- C++ adapter with the reference's ORBextractor interface
// (/root/reference/SingleRobotScenario/include/ORBextractor.h:45-111) over the C ABI.
//
// With -DORBSLAMM_WITH_OPENCV it uses the reference's exact signatures
// (cv::InputArray / std::vector<cv::KeyPoint> / cv::OutputArray) and is a drop-in for
// src/ORBextractor.cc; without it the same class works on flat arrays so that the
// header can be compiled and tested where OpenCV is absent (this image).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "orbslamm_hip.h"
#include "orbslamm_hub.hpp"

#ifdef ORBSLAMM_WITH_OPENCV
#include <cassert>
#include <opencv/cv.h>
#endif

namespace iORB_SLAM {  // the reference's namespace (ORBextractor.h:29)

class ORBextractor;

// One level of mvImagePyramid as the reference lays it out (ORBextractor.cc:1107-1132): the level's pixels sit inside a
// buffer with a 19 px (EDGE_THRESHOLD) BORDER_REFLECT_101 frame; data points at pixel (0, 0) of the level, so
// data[y * step + x] is valid for -19 <= x < cols + 19, -19 <= y < rows + 19 -- what code that indexes the member
// directly (Frame::ComputeStereoMatches, Frame.cc:561,578) relies on.
struct PyramidLevel {
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    std::vector<uint8_t> buf;   // (rows + 38) x (cols + 38)
#ifdef ORBSLAMM_WITH_OPENCV
    cv::Mat mat() { return cv::Mat(rows, cols, CV_8UC1, data, step); }
#endif
};

// std::vector<cv::Mat> mvImagePyramid (ORBextractor.h:85) as a member that fills itself on first use after each
// operator() call: the pyramid lives in HBM and only the stereo path ever reads it on the host, so the eight images
// are not copied back per frame.  mvImagePyramid[l] and .size() work as on the vector; with OpenCV an element IS a
// cv::Mat (a header over the framed buffer, no copy), so the reference's only reader compiles unchanged:
//   mpORBextractorLeft->mvImagePyramid[kpL.octave].rowRange(...).colRange(...)   and   ....cols   (Frame.cc:561,573,578)
class LazyPyramid {
public:
#ifdef ORBSLAMM_WITH_OPENCV
    typedef cv::Mat Level;
#else
    typedef PyramidLevel Level;
#endif
    explicit LazyPyramid(ORBextractor* owner) : owner_(owner) {}
    size_t size() const;
    Level& operator[](size_t level);
    PyramidLevel& framed(size_t level);   // the level inside its 19 px frame, whatever Level is
    void invalidate() { valid_ = false; }
private:
    void fill();
    ORBextractor* owner_;
    std::vector<PyramidLevel> levels_;
#ifdef ORBSLAMM_WITH_OPENCV
    std::vector<cv::Mat> mats_;
#endif
    bool valid_ = false;
};

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // same five arguments as the reference; the extra ones size the device buffers
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int maxWidth = 1920, int maxHeight = 1080, int device = 0) : mvImagePyramid(this)
    {
        OrbxParams p = {nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST};
        if (orbx_create(&p, maxWidth, maxHeight, 1, device, &h_) != ORBX_OK)
            throw std::runtime_error(std::string("ORBextractor(HIP): ") + orbx_last_error());
        const int L = orbx_levels(h_);
        mvScaleFactor.resize(L); mvInvScaleFactor.resize(L); mvLevelSigma2.resize(L); mvInvLevelSigma2.resize(L);
        orbx_scale_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
        cap_ = orbx_max_keypoints(h_);
        kps_.resize(cap_);
    }
    // Several robots on one GPU (MultipleRobotsScenario: a System -- and with it an ORBextractor -- per robot, each driven by
    // its own tracking thread, Examples/Monocular/mono_kitti.cc:83-101): the robots' extractors share ONE hub, opened with
    // Config::track = false and the reference's five parameters; operator() of robot `camera` blocks like the reference's
    // and returns that robot's keypoints and descriptors, while the frames of the robots that call at the same time go
    // through the GPU together (orbslamm_hub.hpp).  Monocular: mvImagePyramid (read by the stereo path only) is not kept.
    ORBextractor(std::shared_ptr<orbslamm::CameraHub> hub, int camera) : mvImagePyramid(this), hub_(hub), cam_(camera)
    {
        if (!hub_ || !hub_->extractor()) throw std::runtime_error("ORBextractor(HIP): the hub is not open");
        h_ = hub_->extractor();
        const int L = orbx_levels(h_);
        mvScaleFactor.resize(L); mvInvScaleFactor.resize(L); mvLevelSigma2.resize(L); mvInvLevelSigma2.resize(L);
        orbx_scale_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data());
        cap_ = hub_->cap();
        kps_.resize(cap_);
    }
    bool onHub() const { return (bool)hub_; }
    orbx_t* handle() { return h_; }
    ~ORBextractor() { if (hub_) hub_->leave(cam_); else orbx_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // ORBextractor.h:60-83, the reference's spelling
    int inline GetLevels() { return orbx_levels(h_); }
    float inline GetScaleFactor() { return orbx_scale_factor(h_); }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    LazyPyramid mvImagePyramid;   // ORBextractor.h:85

#ifdef ORBSLAMM_WITH_OPENCV
    // void operator()(InputArray image, InputArray mask, vector<KeyPoint>&, OutputArray)  ORBextractor.cc:1043
    void operator()(cv::InputArray _image, cv::InputArray /*mask: ignored like the reference*/,
                    std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
    {
        if (_image.empty()) return;
        cv::Mat image = _image.getMat();
        assert(image.type() == CV_8UC1);
        static_assert(sizeof(cv::KeyPoint) == sizeof(OrbxKeyPoint), "cv::KeyPoint layout");
        desc_.resize((size_t)cap_ * 32);
        int n = 0;
        const int rc = run(image.data, image.cols, image.rows, (int)image.step, n);
        if (rc != ORBX_OK) {  // never exit(): Tracking already copes with 0 keypoints (Frame.cc:195-196)
            std::fprintf(stderr, "ORBextractor(HIP): %s\n", orbx_last_error());
            n = 0;
        }
        mvImagePyramid.invalidate();
        _keypoints.clear();
        if (n == 0) { _descriptors.release(); return; }
        _keypoints.resize(n);
        std::memcpy((void*)_keypoints.data(), kps_.data(), (size_t)n * sizeof(OrbxKeyPoint));
        _descriptors.create(n, 32, CV_8U);
        std::memcpy(_descriptors.getMat().data, desc_.data(), (size_t)n * 32);
    }
#else
    // flat-array form of operator(): image rows `stride` bytes apart
    void operator()(const uint8_t* image, int width, int height, int stride,
                    std::vector<OrbxKeyPoint>& keypoints, std::vector<uint8_t>& descriptors)
    {
        if (!image || width <= 0 || height <= 0) return;
        desc_.resize((size_t)cap_ * 32);
        int n = 0;
        const int rc = run(image, width, height, stride, n);
        if (rc != ORBX_OK) { std::fprintf(stderr, "ORBextractor(HIP): %s\n", orbx_last_error()); n = 0; }
        mvImagePyramid.invalidate();
        keypoints.assign(kps_.begin(), kps_.begin() + n);
        descriptors.assign(desc_.begin(), desc_.begin() + (size_t)n * 32);
    }
#endif

    // Device pointers of the frame the last operator() call extracted (keypoints, descriptor rows); they stay
    // untouched until the next-but-one call.  Feed them to ORBmatcher::makeFrame so that the frame's undistorted
    // keys, grid and BoW are built without another trip through the host.
    void lastOnDevice(const OrbxKeyPoint*& d_keypoints, const uint8_t*& d_descriptors)
    {
        OrbxKeyPoint* k = nullptr; uint8_t* d = nullptr;
        orbx_device_results(h_, &k, &d, nullptr, nullptr);
        d_keypoints = k; d_descriptors = d;
    }

protected:
    // one frame through the handle of this extractor, or through the hub it shares with the other robots' extractors
    int run(const uint8_t* image, int width, int height, int stride, int& n)
    {
        if (!hub_) return orbx_extract(h_, image, width, height, stride, kps_.data(), desc_.data(), cap_, &n);
        (void)width; (void)height;   // the hub's frames have one shape (Config::w, h)
        orbslamm::CameraHub::Result res;
        const int rc = hub_->track(cam_, image, stride, kps_.data(), desc_.data(), nullptr, &res);
        n = rc == ORBX_OK ? res.n : 0;
        return rc;
    }
    std::shared_ptr<orbslamm::CameraHub> hub_;
    int cam_ = 0;
    orbx_t* h_ = nullptr;
    int cap_ = 0;
    std::vector<OrbxKeyPoint> kps_;
    std::vector<uint8_t> desc_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

inline size_t LazyPyramid::size() const { return (size_t)orbx_levels(owner_->handle()); }
inline PyramidLevel& LazyPyramid::framed(size_t level)
{
    if (!valid_) fill();
    return levels_.at(level);
}
inline LazyPyramid::Level& LazyPyramid::operator[](size_t level)
{
    if (!valid_) fill();
#ifdef ORBSLAMM_WITH_OPENCV
    return mats_.at(level);
#else
    return levels_.at(level);
#endif
}
inline void LazyPyramid::fill()
{
    if (owner_->onHub()) throw std::runtime_error("mvImagePyramid: an extractor on a hub keeps no pyramid (monocular robots; give a stereo rig a handle of its own)");
    const int E = 19;  // EDGE_THRESHOLD, ORBextractor.cc:74
    const int L = orbx_levels(owner_->handle());
    levels_.resize((size_t)L);
    for (int l = 0; l < L; l++) {
        PyramidLevel& P = levels_[(size_t)l];
        int w = 0, h = 0;
        if (orbx_pyramid_level(owner_->handle(), 0, l, 0, nullptr, &w, &h) != ORBX_OK)
            throw std::runtime_error(std::string("mvImagePyramid: ") + orbx_last_error());
        P.rows = h; P.cols = w; P.step = (size_t)w + 2 * E;
        P.buf.assign(P.step * ((size_t)h + 2 * E), 0);
        P.data = P.buf.data() + (size_t)E * P.step + E;
        std::vector<uint8_t> tight((size_t)w * h);
        orbx_pyramid_level(owner_->handle(), 0, l, 0, tight.data(), &w, &h);
        auto refl = [](int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p; return p; };  // BORDER_REFLECT_101
        for (int y = -E; y < h + E; y++) {
            const uint8_t* src = &tight[(size_t)refl(y, h) * w];
            uint8_t* dst = P.data + (ptrdiff_t)y * (ptrdiff_t)P.step;
            std::memcpy(dst, src, (size_t)w);
            for (int x = 1; x <= E; x++) { dst[-x] = src[refl(-x, w)]; dst[w - 1 + x] = src[refl(w - 1 + x, w)]; }
        }
    }
#ifdef ORBSLAMM_WITH_OPENCV
    mats_.clear();
    for (PyramidLevel& P : levels_) mats_.push_back(P.mat());
#endif
    valid_ = true;
}

}  // namespace iORB_SLAM
