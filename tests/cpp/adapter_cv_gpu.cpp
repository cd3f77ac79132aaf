// GPU run of the drop-in extractor class through the REFERENCE's operator() signature
// (include/ORBextractor.h:57-58: InputArray image, InputArray mask, vector<KeyPoint>&, OutputArray descriptors),
// compiled with -DORBSLAMM_WITH_OPENCV against tests/cpp/mock_opencv (OpenCV is not in this image), checked against
// the C oracle: the call Frame::ExtractORB makes, `(*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors)`
// (src/Frame.cc:247-253).
#include <cassert>
#include <cstdio>
#include <cstring>
#include <random>

#include "ORBextractor_hip.hpp"
#include "../../oracle/orb_oracle.h"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main()
{
    const int W = 752, H = 480, STRIDE = 768;   // a cv::Mat with padded rows (step != cols)
    std::mt19937 rng(11);
    std::vector<uint8_t> buf((size_t)STRIDE * H, 0xEE);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) buf[(size_t)y * STRIDE + x] = (uint8_t)((((x / 17) ^ (y / 9)) * 41 + (rng() & 7)) & 0xFF);
    cv::Mat im(H, W, CV_8UC1, buf.data(), STRIDE);

    iORB_SLAM::ORBextractor* mpORBextractorLeft = new iORB_SLAM::ORBextractor(1200, 1.2f, 8, 20, 7, W, H, 0);
    std::vector<cv::KeyPoint> mvKeys(3);   // stale content: operator() clears it (:1072)
    cv::Mat mDescriptors;
    (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);

    OrcExtractor oex;
    orc_extractor_init(&oex, 1200, 1.2f, 8, 20, 7);
    std::vector<OrcKeyPoint> okps(4000); std::vector<uint8_t> odesc(4000 * 32);
    const int on = orc_extract(&oex, buf.data(), W, H, STRIDE, okps.data(), odesc.data(), 4000, nullptr, nullptr, nullptr);
    EXPECT(on > 300 && (size_t)on == mvKeys.size());
    EXPECT(mDescriptors.rows == on && mDescriptors.cols == 32 && mDescriptors.step == 32);
    EXPECT(std::memcmp(okps.data(), mvKeys.data(), (size_t)on * sizeof(cv::KeyPoint)) == 0);
    EXPECT(std::memcmp(odesc.data(), mDescriptors.data, (size_t)on * 32) == 0);
    EXPECT(mvKeys[0].class_id == -1 && mvKeys[0].octave == 0);

    // getters read by the Frame constructors (Frame.cc:69-75)
    EXPECT(mpORBextractorLeft->GetLevels() == 8 && mpORBextractorLeft->GetScaleFactor() == 1.2f);
    EXPECT(mpORBextractorLeft->GetScaleFactors().size() == 8 && mpORBextractorLeft->GetInverseScaleSigmaSquares().size() == 8);
    // mvImagePyramid[l] as a cv::Mat header over the 19 px framed buffer (Frame.cc:561,578 index it with negative offsets)
    iORB_SLAM::PyramidLevel& P = mpORBextractorLeft->mvImagePyramid.framed(1);
    cv::Mat& lvl = mpORBextractorLeft->mvImagePyramid[1];   // an element is a cv::Mat, as in std::vector<cv::Mat>
    EXPECT(lvl.cols == P.cols && lvl.rows == P.rows && lvl.data == P.data && lvl.data[-19 * (long)lvl.step - 19] == lvl.data[19 * (long)lvl.step + 19]);
    {   // the reference's reader, verbatim in shape (Frame.cc:561, :573): a window around a keypoint near the image corner
        const int w = 5;
        const float scaledvL = 3.f, scaleduL = 2.f;   // the window reaches into the 19 px frame, like the reference's can
        cv::Mat IL = mpORBextractorLeft->mvImagePyramid[1].rowRange(scaledvL - w, scaledvL + w + 1).colRange(scaleduL - w, scaleduL + w + 1);
        EXPECT(IL.rows == 2 * w + 1 && IL.cols == 2 * w + 1 && IL.step == lvl.step);
        EXPECT(IL.data == lvl.data + (ptrdiff_t)(3 - w) * (ptrdiff_t)lvl.step + (2 - w));
        EXPECT(IL.data[0] == lvl.data[(ptrdiff_t)(w - 3) * (ptrdiff_t)lvl.step + (w - 2)]);   // BORDER_REFLECT_101: pixel (-3, -2) mirrors (3, 2)
        EXPECT(mpORBextractorLeft->mvImagePyramid[1].cols == P.cols);
    }

    // an empty image leaves the outputs untouched (:1046-1047)
    std::vector<cv::KeyPoint> keep = mvKeys;
    cv::Mat none;
    (*mpORBextractorLeft)(none, cv::Mat(), mvKeys, mDescriptors);
    EXPECT(mvKeys.size() == keep.size() && mDescriptors.rows == on);

    // a featureless image: keypoints cleared, descriptors released (:1064-1065)
    std::vector<uint8_t> flat((size_t)W * H, 90);
    cv::Mat gray(H, W, CV_8UC1, flat.data(), W);
    (*mpORBextractorLeft)(gray, cv::Mat(), mvKeys, mDescriptors);
    EXPECT(mvKeys.empty() && mDescriptors.empty());

    delete mpORBextractorLeft;
    if (fails) { std::printf("adapter_cv_gpu: %d failures\n", fails); return 1; }
    std::printf("adapter_cv_gpu ok (%d keypoints)\n", on);
    return 0;
}
