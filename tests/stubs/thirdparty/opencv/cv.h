// stand-in: the OpenCV 2 types the reference's driver files name (Mat, Mat_<T>, Vec, Size_, resize)
#pragma once
#include <cstring>
#define CV_8UC3 16
#define CV_16SC1 3
#define CV_16UC1 2
#define CV_32FC1 5
#define CV_INTER_CUBIC 2
typedef unsigned char uchar;
namespace cv {
template <typename T, int N>
struct Vec {
  T val[N];
  Vec() {}
  Vec(T a, T b) { val[0] = a; val[1] = b; }
  Vec(T a, T b, T c) { val[0] = a; val[1] = b; val[2] = c; }
  Vec(T a, T b, T c, T d) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  T &operator[](int i) { return val[i]; }
  const T &operator[](int i) const { return val[i]; }
};
typedef Vec<uchar, 3> Vec3b;
template <typename T>
struct Size_ {
  T width, height;
  Size_() : width(), height() {}
  Size_(T w, T h) : width(w), height(h) {}
};
typedef Size_<int> Size2i;
typedef Size2i Size;
class Mat {
 public:
  int rows, cols;
  uchar *data;
  Mat() : rows(0), cols(0), data(nullptr) {}
  Mat(int r, int c, int /*type*/) : rows(r), cols(c), data(nullptr) {}
  template <typename T> T &at(int, int) { static T t; return t; }
  template <typename T> const T &at(int, int) const { static T t; return t; }
  int type() const { return 0; }
  Size size() const { return Size(cols, rows); }
};
template <typename T>
class Mat_ : public Mat {
 public:
  Mat_() {}
  Mat_(int r, int c) : Mat(r, c, 0) {}
  T &operator()(int, int) { static T t; return t; }
  const T &operator()(int, int) const { static T t; return t; }
};
typedef Mat_<uchar> Mat1b;
typedef Mat_<short> Mat1s;
typedef Mat_<Vec3b> Mat3b;
inline void resize(const Mat &, Mat &, Size, double = 0, double = 0, int = 1) {}
}  // namespace cv
