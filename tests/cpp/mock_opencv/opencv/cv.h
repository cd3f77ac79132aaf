// Test infrastructure: the handful of cv:: names include/ORBextractor_hip.hpp touches under -DORBSLAMM_WITH_OPENCV
// (cv::Mat, cv::InputArray, cv::OutputArray, cv::KeyPoint, CV_8U / CV_8UC1), so that the reference-signature
// operator() of the drop-in class is compiled and run where OpenCV is absent.  Plain data holders with OpenCV's member
// names and cv::KeyPoint's 28-byte layout; nothing here computes anything.  The image-processing functions
// tools/check_vs_opencv names (FAST, resize, GaussianBlur, fastAtan2) are DECLARED only, for a -fsyntax-only pass of
// that program: they have no definition anywhere in this repository.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_VERSION "mock (declarations only)"
typedef unsigned char uchar;

namespace cv {

struct Point2f { float x, y; };
template <class T> struct Point_ { T x, y; };
typedef Point_<int> Point;
typedef Point_<int> Point2i;
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
enum { INTER_LINEAR = 1, BORDER_REFLECT_101 = 4 };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int /*type*/, void* d, size_t s) : rows(r), cols(c), step(s), data((unsigned char*)d) {}
    void create(int r, int c, int /*type*/)
    {
        rows = r; cols = c; step = (size_t)c;
        own_.reset(new std::vector<unsigned char>((size_t)r * c));
        data = own_->data();
    }
    void release() { rows = cols = 0; step = 0; data = nullptr; own_.reset(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    template <class T> T& at(int y, int x) { return *(T*)(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    template <class T> const T& at(int y, int x) const { return *(const T*)(data + (size_t)y * step + (size_t)x * sizeof(T)); }
    unsigned char* ptr(int y) { return data + (size_t)y * step; }
    const unsigned char* ptr(int y) const { return data + (size_t)y * step; }
    Mat clone() const;   // declared only
    // sub-matrix headers over the same memory (what Frame::ComputeStereoMatches asks of a pyramid level, Frame.cc:561,578)
    Mat rowRange(int a, int b) const { Mat m(*this); m.data = data + (ptrdiff_t)a * (ptrdiff_t)step; m.rows = b - a; return m; }
    Mat colRange(int a, int b) const { Mat m(*this); m.data = data + a; m.cols = b - a; return m; }
    int type() const { return CV_8UC1; }
private:
    std::shared_ptr<std::vector<unsigned char> > own_;
};

class _InputArray {
public:
    _InputArray() {}
    _InputArray(const Mat& m) : m_(&m) {}
    bool empty() const { return !m_ || m_->empty(); }
    Mat getMat() const { return m_ ? *m_ : Mat(); }
private:
    const Mat* m_ = nullptr;
};
typedef const _InputArray& InputArray;

class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
private:
    Mat* m_;
};
typedef const _OutputArray& OutputArray;

// declared only (see the head of this file)
void FAST(InputArray image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmaxSuppression);
void resize(InputArray src, OutputArray dst, Size dsize, double fx, double fy, int interpolation);
void GaussianBlur(InputArray src, OutputArray dst, Size ksize, double sigmaX, double sigmaY, int borderType);
float fastAtan2(float y, float x);

}  // namespace cv
