// Test infrastructure: stand-ins for the reference's Frame / KeyFrame / MapPoint (and the few cv:: types they expose)
// carrying exactly the member names src/ORBmatcher.cc touches, so that the drop-in template
// iORB_SLAM::ORBmatcherT<Frame, KeyFrame, MapPoint> (include/ORBmatcher_hip.hpp) can be instantiated and run without
// OpenCV.  Plain data holders: nothing here computes what the product computes.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <vector>

namespace mock {

struct Point2f { float x, y; };
struct KeyPoint {  // cv::KeyPoint's layout (28 bytes)
    Point2f pt; float size, angle, response; int octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// row-major matrix of float (CV_32F) or unsigned char (CV_8U) elements, shared storage like cv::Mat
class Mat {
public:
    int rows = 0, cols = 0;
    Mat() {}
    static Mat f32(int r, int c) { Mat m; m.rows = r; m.cols = c; m.esz = 4; m.buf = std::make_shared<std::vector<uint8_t> >((size_t)r * c * 4, (uint8_t)0); return m; }
    static Mat u8(int r, int c) { Mat m; m.rows = r; m.cols = c; m.esz = 1; m.buf = std::make_shared<std::vector<uint8_t> >((size_t)r * c, (uint8_t)0); return m; }
    Mat clone() const { Mat m = *this; if (buf) m.buf = std::make_shared<std::vector<uint8_t> >(*buf); return m; }  // two allocations, like cv::Mat::clone in OpenCV 3 (UMatData + data)
    bool isContinuous() const { return true; }
    template <class T> T& at(int r, int c = 0) { return *reinterpret_cast<T*>(&(*buf)[((size_t)r * cols + c) * esz]); }
    template <class T> const T& at(int r, int c = 0) const { return *reinterpret_cast<const T*>(&(*buf)[((size_t)r * cols + c) * esz]); }
    template <class T> T* ptr(int r) { return reinterpret_cast<T*>(&(*buf)[(size_t)r * cols * esz]); }
    template <class T> const T* ptr(int r) const { return reinterpret_cast<const T*>(&(*buf)[(size_t)r * cols * esz]); }
private:
    int esz = 4;
    std::shared_ptr<std::vector<uint8_t> > buf;
};

typedef std::map<unsigned, std::vector<unsigned> > FeatureVector;  // DBoW2::FeatureVector

struct KeyFrame;

struct MapPoint {
    Mat mWorldPos = Mat::f32(3, 1), mNormalVector = Mat::f32(3, 1), mDescriptor = Mat::u8(1, 32);
    float mfMinDistance = 0.f, mfMaxDistance = 0.f;
    std::map<KeyFrame*, size_t> mObservations;
    bool mbBad = false;
    MapPoint* mpReplaced = nullptr;
    int id = 0;
    // Tracking's per-frame fields (MapPoint.h:97-102)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = -1;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 1.f;

    Mat GetWorldPos() { return mWorldPos.clone(); }
    Mat GetNormal() { return mNormalVector.clone(); }
    Mat GetDescriptor() { return mDescriptor.clone(); }
    int Observations() { return (int)mObservations.size(); }
    bool isBad() { return mbBad; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }  // MapPoint.cc:373-383
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
    int PredictScale(const float& currentDist, const float& logScaleFactor)  // MapPoint.cc:385-394
    {
        const float ratio = mfMaxDistance / currentDist;
        return (int)std::ceil(std::log(ratio) / logScaleFactor);
    }
    bool IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
    int GetIndexInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
    void AddObservation(KeyFrame* pKF, size_t idx) { if (!mObservations.count(pKF)) mObservations[pKF] = idx; }
    void Replace(MapPoint* pMP);  // below
};

struct KeyFrame {
    int N = 0;
    std::vector<KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    Mat mDescriptors;
    FeatureVector mFeatVec;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    float mfLogScaleFactor = 0;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0, mnGridCols = 64, mnGridRows = 48;
    float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    std::vector<MapPoint*> mvpMapPoints;
    Mat Tcw = Mat::f32(4, 4), Ow = Mat::f32(3, 1);
    int id = 0;

    std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; }
    std::set<MapPoint*> GetMapPoints()  // KeyFrame.cc: the good ones
    {
        std::set<MapPoint*> s;
        for (size_t i = 0; i < mvpMapPoints.size(); i++) if (mvpMapPoints[i] && !mvpMapPoints[i]->isBad()) s.insert(mvpMapPoints[i]);
        return s;
    }
    MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
    bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }
    Mat GetRotation() { Mat R = Mat::f32(3, 3); for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R.at<float>(r, c) = Tcw.at<float>(r, c); return R; }
    Mat GetTranslation() { Mat t = Mat::f32(3, 1); for (int r = 0; r < 3; r++) t.at<float>(r, 0) = Tcw.at<float>(r, 3); return t; }
    Mat GetCameraCenter() { return Ow.clone(); }
};

inline void MapPoint::Replace(MapPoint* pMP)  // MapPoint.cc:172-217, the parts the matcher's callers can observe
{
    if (pMP == this) return;
    std::map<KeyFrame*, size_t> obs = mObservations;
    mObservations.clear();
    mbBad = true;
    mpReplaced = pMP;
    for (std::map<KeyFrame*, size_t>::iterator it = obs.begin(); it != obs.end(); ++it) {
        KeyFrame* pKF = it->first;
        if (!pMP->IsInKeyFrame(pKF)) { pKF->mvpMapPoints[it->second] = pMP; pMP->AddObservation(pKF, it->second); }
        else pKF->mvpMapPoints[it->second] = nullptr;
    }
}

struct Frame {
    int N = 0;
    std::vector<KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight;
    Mat mDescriptors;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    FeatureVector mFeatVec;
    Mat mTcw = Mat::f32(4, 4);
    float mb = 0, mbf = 0;
    std::vector<float> mvScaleFactors;
    float mfLogScaleFactor = 0;
    // process-wide statics in the reference (Frame.cc:29-33)
    static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
};
float Frame::fx = 0, Frame::fy = 0, Frame::cx = 0, Frame::cy = 0, Frame::mnMinX = 0, Frame::mnMaxX = 0, Frame::mnMinY = 0,
      Frame::mnMaxY = 0, Frame::mfGridElementWidthInv = 0, Frame::mfGridElementHeightInv = 0;

}  // namespace mock
