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
#include <stdexcept>
#include <string>
#include <vector>

#include "orbslamm_hip.h"

#ifdef ORBSLAMM_WITH_OPENCV
#include <opencv/cv.h>
#endif

namespace iORB_SLAM {  // the reference's namespace (ORBextractor.h:29)

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // same five arguments as the reference; the extra ones size the device buffers
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                 int maxWidth = 1920, int maxHeight = 1080, int device = 0)
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
    ~ORBextractor() { orbx_destroy(h_); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    int GetLevels() { return orbx_levels(h_); }
    float GetScaleFactor() { return orbx_scale_factor(h_); }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

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
        const int rc = orbx_extract(h_, image.data, image.cols, image.rows, (int)image.step, kps_.data(), desc_.data(), cap_, &n);
        if (rc != ORBX_OK) {  // never exit(): Tracking already copes with 0 keypoints (Frame.cc:195-196)
            std::fprintf(stderr, "ORBextractor(HIP): %s\n", orbx_last_error());
            n = 0;
        }
        _keypoints.clear();
        if (n == 0) { _descriptors.release(); return; }
        _keypoints.resize(n);
        std::memcpy((void*)_keypoints.data(), kps_.data(), (size_t)n * sizeof(OrbxKeyPoint));
        _descriptors.create(n, 32, CV_8U);
        std::memcpy(_descriptors.getMat().data, desc_.data(), (size_t)n * 32);
        // mvImagePyramid is filled lazily: see pyramidLevel()
    }
    // std::vector<cv::Mat> mvImagePyramid (ORBextractor.h:85) is only read by the stereo path;
    // fetch a level on demand instead of copying 8 images back per frame.
    cv::Mat pyramidLevel(int level)
    {
        int w = 0, hh = 0;
        orbx_pyramid_level(h_, 0, level, 0, nullptr, &w, &hh);
        cv::Mat m(hh, w, CV_8UC1);
        orbx_pyramid_level(h_, 0, level, 0, m.data, &w, &hh);
        return m;
    }
#else
    // flat-array form of operator(): image rows `stride` bytes apart
    void operator()(const uint8_t* image, int width, int height, int stride,
                    std::vector<OrbxKeyPoint>& keypoints, std::vector<uint8_t>& descriptors)
    {
        if (!image || width <= 0 || height <= 0) return;
        desc_.resize((size_t)cap_ * 32);
        int n = 0;
        const int rc = orbx_extract(h_, image, width, height, stride, kps_.data(), desc_.data(), cap_, &n);
        if (rc != ORBX_OK) { std::fprintf(stderr, "ORBextractor(HIP): %s\n", orbx_last_error()); n = 0; }
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
    orbx_t* h_ = nullptr;
    int cap_ = 0;
    std::vector<OrbxKeyPoint> kps_;
    std::vector<uint8_t> desc_;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace iORB_SLAM
