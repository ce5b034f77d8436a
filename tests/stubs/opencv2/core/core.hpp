// tests/stubs/opencv2/core/core.hpp -- DECLARATIONS ONLY, for the syntax pass over bindings/pwn_hip/*.cpp (see tests/stubs/Eigen/Core)
#ifndef PWN_STUB_OPENCV_CORE
#define PWN_STUB_OPENCV_CORE
#include <string>
#include <vector>
#define CV_8UC1 0
#define CV_16UC1 2
#define CV_32SC1 4
#define CV_32FC1 5
#define CV_64FC1 6
namespace cv {
typedef unsigned char uchar;
struct Size { int width, height; Size(); Size(int, int); };
struct Scalar { Scalar(); Scalar(double); Scalar(double, double, double, double = 0); };
class Mat {
 public:
  int rows, cols, flags, dims; uchar* data;
  Mat(); Mat(int, int, int); Mat(int, int, int, const Scalar&); Mat(int, int, int, void*, size_t = 0); Mat(Size, int);
  void create(int, int, int); void create(Size, int); void release(); Mat clone() const; void copyTo(Mat&) const; void convertTo(Mat&, int, double = 1, double = 0) const;
  Mat& setTo(const Scalar&); Mat& operator=(const Scalar&); int type() const; int depth() const; int channels() const; size_t total() const; bool empty() const; bool isContinuous() const; Size size() const;
  size_t elemSize() const; template <typename T> T* ptr(int = 0); template <typename T> const T* ptr(int = 0) const; uchar* ptr(int = 0); const uchar* ptr(int = 0) const;
  template <typename T> T& at(int, int); template <typename T> const T& at(int, int) const; template <typename T> T& at(int); template <typename T> const T& at(int) const;
  static Mat zeros(int, int, int); static Mat ones(int, int, int);
};
template <typename T> class Mat_ : public Mat {
 public:
  Mat_(); Mat_(int, int); Mat_(int, int, const T&); Mat_(const Mat&); Mat_(int, int, T*, size_t = 0);
  Mat_& operator=(const Mat&); Mat_& operator=(const T&); void create(int, int); Mat_ clone() const;
  T& operator()(int, int); const T& operator()(int, int) const; T& operator()(int); const T& operator()(int) const; T* operator[](int); const T* operator[](int) const;
  T* ptr(int = 0); const T* ptr(int = 0) const;
};
Mat abs(const Mat&); Mat operator-(const Mat&, const Mat&); Mat operator&(const Mat&, const Mat&); Mat operator>(const Mat&, double); Mat operator<(const Mat&, double);
int countNonZero(const Mat&); Scalar sum(const Mat&);
}  // namespace cv
#endif
