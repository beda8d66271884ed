// ORUtils.h -- the ORUtils types the reference callers touch:
//   Vector2/3/4<T> (x,y,z,w / r,g,b,a / width,height, getValues()), Matrix4f (column-major m[16], at(), operator(),
//   inv(), setIdentity(), 16-argument column-major ctor, operator*), MemoryBlock<T>/Image<T> (GetData, ChangeDims,
//   noDims, ctor (size, allocate_CPU, allocate_GPU)), MEMORYDEVICE_CPU.
// Call sites: InfiniTamDriver.h:24-41,103-104,153-156; InfiniTamDriver.cpp:42-45,74,85-98,107-118,208-225;
// DenseSlam.h:103-104,152,163.  Host-only: the engine moves data to and from the GPU through the C ABI, so
// images need CPU storage only (GetData(MEMORYDEVICE_CUDA) is rejected).
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <stdexcept>

#include "dslam_fusion.h"

enum MemoryDeviceType { MEMORYDEVICE_CPU, MEMORYDEVICE_CUDA };

namespace ORUtils {

template <class T> struct Vector2 {
  union { struct { T x, y; }; struct { T width, height; }; struct { T s, t; }; T v[2]; };
  Vector2() : x(0), y(0) {}
  Vector2(T a) : x(a), y(a) {}
  Vector2(T a, T b) : x(a), y(b) {}
  T &operator[](int i) { return v[i]; }
  const T &operator[](int i) const { return v[i]; }
  const T *getValues() const { return v; }
  T *getValues() { return v; }
};
template <class T> struct Vector3 {
  union { struct { T x, y, z; }; struct { T r, g, b; }; T v[3]; };
  Vector3() : x(0), y(0), z(0) {}
  Vector3(T a) : x(a), y(a), z(a) {}
  Vector3(T a, T b, T c) : x(a), y(b), z(c) {}
  T &operator[](int i) { return v[i]; }
  const T &operator[](int i) const { return v[i]; }
  const T *getValues() const { return v; }
  T *getValues() { return v; }
};
template <class T> struct Vector4 {
  union { struct { T x, y, z, w; }; struct { T r, g, b, a; }; T v[4]; };
  Vector4() : x(0), y(0), z(0), w(0) {}
  Vector4(T s) : x(s), y(s), z(s), w(s) {}
  Vector4(T a_, T b_, T c_, T d_) : x(a_), y(b_), z(c_), w(d_) {}
  T &operator[](int i) { return v[i]; }
  const T &operator[](int i) const { return v[i]; }
  const T *getValues() const { return v; }
  T *getValues() { return v; }
};

template <class T> struct Matrix4 {
  T m[16];  // column-major: at(col, row) = m[col * 4 + row]
  Matrix4() { for (int i = 0; i < 16; i++) m[i] = 0; }
  Matrix4(T a00, T a01, T a02, T a03, T a10, T a11, T a12, T a13, T a20, T a21, T a22, T a23, T a30, T a31, T a32,
          T a33) {
    const T t[16] = {a00, a01, a02, a03, a10, a11, a12, a13, a20, a21, a22, a23, a30, a31, a32, a33};
    for (int i = 0; i < 16; i++) m[i] = t[i];
  }
  void setIdentity() { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? (T)1 : (T)0; }
  T &at(int x, int y) { return m[y | (x << 2)]; }
  const T &at(int x, int y) const { return m[y | (x << 2)]; }
  T &operator()(int x, int y) { return at(x, y); }
  const T &operator()(int x, int y) const { return at(x, y); }
  Vector4<T> getColumn(int c) const { return Vector4<T>(m[c * 4], m[c * 4 + 1], m[c * 4 + 2], m[c * 4 + 3]); }

  friend Matrix4 operator*(const Matrix4 &l, const Matrix4 &r) {
    Matrix4 o;
    for (int c = 0; c < 4; c++)
      for (int rr = 0; rr < 4; rr++) {
        T s = 0;
        for (int k = 0; k < 4; k++) s += l.m[k * 4 + rr] * r.m[c * 4 + k];
        o.m[c * 4 + rr] = s;
      }
    return o;
  }
  friend Vector4<T> operator*(const Matrix4 &l, const Vector4<T> &v) {
    return Vector4<T>(l.m[0] * v.x + l.m[4] * v.y + l.m[8] * v.z + l.m[12] * v.w,
                      l.m[1] * v.x + l.m[5] * v.y + l.m[9] * v.z + l.m[13] * v.w,
                      l.m[2] * v.x + l.m[6] * v.y + l.m[10] * v.z + l.m[14] * v.w,
                      l.m[3] * v.x + l.m[7] * v.y + l.m[11] * v.z + l.m[15] * v.w);
  }
  // cofactor inverse (the engine applies the same expansion to derive invM from M)
  bool inv(Matrix4 &out) const {
    T tmp[12], src[16], det;
    T *dst = out.m;
    for (int i = 0; i < 4; i++) { src[i] = m[i * 4]; src[i + 4] = m[i * 4 + 1]; src[i + 8] = m[i * 4 + 2]; src[i + 12] = m[i * 4 + 3]; }
    tmp[0] = src[10] * src[15]; tmp[1] = src[11] * src[14]; tmp[2] = src[9] * src[15]; tmp[3] = src[11] * src[13];
    tmp[4] = src[9] * src[14]; tmp[5] = src[10] * src[13]; tmp[6] = src[8] * src[15]; tmp[7] = src[11] * src[12];
    tmp[8] = src[8] * src[14]; tmp[9] = src[10] * src[12]; tmp[10] = src[8] * src[13]; tmp[11] = src[9] * src[12];
    dst[0] = (tmp[0] * src[5] + tmp[3] * src[6] + tmp[4] * src[7]) - (tmp[1] * src[5] + tmp[2] * src[6] + tmp[5] * src[7]);
    dst[1] = (tmp[1] * src[4] + tmp[6] * src[6] + tmp[9] * src[7]) - (tmp[0] * src[4] + tmp[7] * src[6] + tmp[8] * src[7]);
    dst[2] = (tmp[2] * src[4] + tmp[7] * src[5] + tmp[10] * src[7]) - (tmp[3] * src[4] + tmp[6] * src[5] + tmp[11] * src[7]);
    dst[3] = (tmp[5] * src[4] + tmp[8] * src[5] + tmp[11] * src[6]) - (tmp[4] * src[4] + tmp[9] * src[5] + tmp[10] * src[6]);
    dst[4] = (tmp[1] * src[1] + tmp[2] * src[2] + tmp[5] * src[3]) - (tmp[0] * src[1] + tmp[3] * src[2] + tmp[4] * src[3]);
    dst[5] = (tmp[0] * src[0] + tmp[7] * src[2] + tmp[8] * src[3]) - (tmp[1] * src[0] + tmp[6] * src[2] + tmp[9] * src[3]);
    dst[6] = (tmp[3] * src[0] + tmp[6] * src[1] + tmp[11] * src[3]) - (tmp[2] * src[0] + tmp[7] * src[1] + tmp[10] * src[3]);
    dst[7] = (tmp[4] * src[0] + tmp[9] * src[1] + tmp[10] * src[2]) - (tmp[5] * src[0] + tmp[8] * src[1] + tmp[11] * src[2]);
    tmp[0] = src[2] * src[7]; tmp[1] = src[3] * src[6]; tmp[2] = src[1] * src[7]; tmp[3] = src[3] * src[5];
    tmp[4] = src[1] * src[6]; tmp[5] = src[2] * src[5]; tmp[6] = src[0] * src[7]; tmp[7] = src[3] * src[4];
    tmp[8] = src[0] * src[6]; tmp[9] = src[2] * src[4]; tmp[10] = src[0] * src[5]; tmp[11] = src[1] * src[4];
    dst[8] = (tmp[0] * src[13] + tmp[3] * src[14] + tmp[4] * src[15]) - (tmp[1] * src[13] + tmp[2] * src[14] + tmp[5] * src[15]);
    dst[9] = (tmp[1] * src[12] + tmp[6] * src[14] + tmp[9] * src[15]) - (tmp[0] * src[12] + tmp[7] * src[14] + tmp[8] * src[15]);
    dst[10] = (tmp[2] * src[12] + tmp[7] * src[13] + tmp[10] * src[15]) - (tmp[3] * src[12] + tmp[6] * src[13] + tmp[11] * src[15]);
    dst[11] = (tmp[5] * src[12] + tmp[8] * src[13] + tmp[11] * src[14]) - (tmp[4] * src[12] + tmp[9] * src[13] + tmp[10] * src[14]);
    dst[12] = (tmp[2] * src[10] + tmp[5] * src[11] + tmp[1] * src[9]) - (tmp[4] * src[11] + tmp[0] * src[9] + tmp[3] * src[10]);
    dst[13] = (tmp[8] * src[11] + tmp[0] * src[8] + tmp[7] * src[10]) - (tmp[6] * src[10] + tmp[9] * src[11] + tmp[1] * src[8]);
    dst[14] = (tmp[6] * src[9] + tmp[11] * src[11] + tmp[3] * src[8]) - (tmp[10] * src[11] + tmp[2] * src[8] + tmp[7] * src[9]);
    dst[15] = (tmp[10] * src[10] + tmp[4] * src[8] + tmp[9] * src[9]) - (tmp[8] * src[9] + tmp[11] * src[10] + tmp[5] * src[8]);
    det = src[0] * dst[0] + src[1] * dst[1] + src[2] * dst[2] + src[3] * dst[3];
    if (det == 0) { for (int i = 0; i < 16; i++) dst[i] = 0; return false; }
    for (int i = 0; i < 16; i++) dst[i] = dst[i] * ((T)1 / det);
    return true;
  }
};

template <typename T> class MemoryBlock {
 public:
  size_t dataSize;
  /// allocate_CUDA: upstream gives such a block a device side and a page-locked host side (ORUtils/MemoryBlock.h
  /// Allocate).  Here the device side lives inside libdslam_fusion; the host side is page-locked all the same, so
  /// dslam_view_update can DMA straight out of the caller's image.
  MemoryBlock(size_t n, bool /*allocate_CPU*/, bool allocate_CUDA) : dataSize(n), data_(nullptr), pinned_(allocate_CUDA), stale_(false) {
    data_ = allocate(n);
  }
  virtual ~MemoryBlock() { release(data_); }
  T *GetData(MemoryDeviceType t) { check(t); sync(); return data_; }
  const T *GetData(MemoryDeviceType t) const { check(t); sync(); return data_; }
  void Clear(unsigned char v = 0) { stale_ = false; memset(data_, v, dataSize * sizeof(T)); }
  void UpdateDeviceFromHost() const {}
  void UpdateHostFromDevice() const { sync(); }
  /// Lazy host mirror: the device copy inside libdslam_fusion is the master; `pull` fills the host buffer from it
  /// the next time somebody asks for the host pointer (most frames nobody does, which saves a D2H copy per frame).
  void SetHostPull(std::function<void(T *)> pull) { pull_ = std::move(pull); }
  void MarkHostStale() { stale_ = (bool)pull_; }
 protected:
  static void check(MemoryDeviceType t) {
    if (t != MEMORYDEVICE_CPU) throw std::runtime_error("device mirrors live inside libdslam_fusion; use MEMORYDEVICE_CPU");
  }
  void sync() const {
    if (stale_) { stale_ = false; pull_(data_); }
  }
  void resize(size_t n) {
    if (n > dataSize) { release(data_); data_ = allocate(n); }
    dataSize = n;
  }
  T *allocate(size_t n) const {
    void *p = nullptr;
    if (!pinned_) p = calloc(n ? n : 1, sizeof(T));
    else if (dslam_host_alloc((n ? n : 1) * sizeof(T), &p) != DSLAM_OK) throw std::runtime_error(dslam_last_error());
    if (p == nullptr) throw std::bad_alloc();
    return (T *)p;
  }
  void release(T *p) const {
    if (pinned_) dslam_host_free(p); else free(p);
  }
  T *data_;
  bool pinned_;
  mutable bool stale_;
  std::function<void(T *)> pull_;
  MemoryBlock(const MemoryBlock &) = delete;
  MemoryBlock &operator=(const MemoryBlock &) = delete;
};

template <typename T> class Image : public MemoryBlock<T> {
 public:
  Vector2<int> noDims;
  Image(Vector2<int> dims, bool allocate_CPU, bool allocate_CUDA)
      : MemoryBlock<T>((size_t)dims.x * dims.y, allocate_CPU, allocate_CUDA), noDims(dims) {}
  Image(Vector2<int> dims, MemoryDeviceType) : MemoryBlock<T>((size_t)dims.x * dims.y, true, false), noDims(dims) {}
  void ChangeDims(Vector2<int> newDims) {
    if (newDims.x != noDims.x || newDims.y != noDims.y) {
      this->resize((size_t)newDims.x * newDims.y);
      noDims = newDims;
    }
  }
};

}  // namespace ORUtils

typedef ORUtils::Vector2<short> Vector2s;
typedef ORUtils::Vector2<int> Vector2i;
typedef ORUtils::Vector2<float> Vector2f;
typedef ORUtils::Vector3<short> Vector3s;
typedef ORUtils::Vector3<int> Vector3i;
typedef ORUtils::Vector3<float> Vector3f;
typedef ORUtils::Vector3<unsigned char> Vector3u;
typedef ORUtils::Vector4<float> Vector4f;
typedef ORUtils::Vector4<int> Vector4i;
typedef ORUtils::Vector4<short> Vector4s;
typedef ORUtils::Vector4<unsigned char> Vector4u;
typedef ORUtils::Matrix4<float> Matrix4f;
typedef unsigned char uchar;
typedef ORUtils::Image<Vector4u> ITMUChar4Image;
typedef ORUtils::Image<short> ITMShortImage;
typedef ORUtils::Image<float> ITMFloatImage;
typedef ORUtils::Image<Vector4f> ITMFloat4Image;
