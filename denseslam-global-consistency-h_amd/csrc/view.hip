// view.hip -- the pixel loops that run just before the fusion path, on the device:
//   * CvToItm(cv::Mat3b -> ITMUChar4Image): BGR -> RGBA, a = 255           [REF InfiniTamDriver.cpp:84-103]
//   * ITMViewBuilder::UpdateView(..., useBilateralFilter = true): five passes of the 5x5 bilateral depth filter
//     (upstream InfiniTAM v2 ITMViewBuilder_Shared.h filterDepth, call site [REF InfiniTamDriver.cpp:280-288])
//   * DenseSlam::depthPostProcessing: reprojection consistency filter against the previous keyframe's depth
//                                                                           [REF DenseSlam.cpp:434-552]
// Pixel-parallel byte/float work (the five filter passes share one LDS-tiled launch); nothing here is GEMM shaped.
#include "dslam_internal.h"

namespace dslam {

// ---------------------------------------------------------------------------------------------------------
// BGR -> RGBA
// ---------------------------------------------------------------------------------------------------------
// Each thread converts four pixels: three aligned dword loads (12 bytes) in, one 16-byte store out.
__global__ __launch_bounds__(256) void k_bgr_to_rgba(const unsigned *__restrict__ bgr, uint4 *__restrict__ rgba4,
                                                     const unsigned char *__restrict__ bgr_bytes,
                                                     uchar4 *__restrict__ rgba, int npix) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int n4 = npix >> 2;
  if (q < n4) {
    const unsigned a = bgr[3 * q], b = bgr[3 * q + 1], c = bgr[3 * q + 2];
    // bytes: a = B0 G0 R0 B1 | b = G1 R1 B2 G2 | c = R2 B3 G3 R3 (little endian)
    auto pack = [](unsigned bl, unsigned g, unsigned r) { return (r & 255u) | ((g & 255u) << 8) | ((bl & 255u) << 16) | 0xff000000u; };
    uint4 o;
    o.x = pack(a, a >> 8, a >> 16);
    o.y = pack(a >> 24, b, b >> 8);
    o.z = pack(b >> 16, b >> 24, c);
    o.w = pack(c >> 8, c >> 16, c >> 24);
    rgba4[q] = o;
  }
  // tail pixels (npix not a multiple of four)
  if (q == 0)
    for (int i = n4 * 4; i < npix; i++) rgba[i] = make_uchar4(bgr_bytes[3 * i + 2], bgr_bytes[3 * i + 1], bgr_bytes[3 * i], 255);
}

int launch_bgr_to_rgba(dslam_engine *e, const void *bgr_dev, uchar4 *rgba_dev, int npix) {
  const int n4 = npix >> 2;
  hipLaunchKernelGGL(k_bgr_to_rgba, dim3((max(n4, 1) + 255) / 256), dim3(256), 0, e->stream,
                     reinterpret_cast<const unsigned *>(bgr_dev), reinterpret_cast<uint4 *>(rgba_dev),
                     reinterpret_cast<const unsigned char *>(bgr_dev), rgba_dev, npix);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// bilateral depth filter
// ---------------------------------------------------------------------------------------------------------
// exp(x) for x <= 0, the same operation sequence as oracle/dslam_oracle.cpp det_exp(): Cody-Waite reduction, a
// degree-6 polynomial in explicit FMAs, scaling by an exact power of two.  Within 1 ulp of libm's expf
// (tests/test_oracle_kat.py); arguments below -86 return 0 -- such a weight is < 2^-124 next to the centre tap's
// weight of exactly 1, far below rounding in both sums.  Sharing the sequence makes the filter bit-identical on
// host and device, so the fused map stays byte-comparable when the filter is on.
__device__ __forceinline__ float det_exp(float x) {
  if (x < -86.0f) return 0.0f;
  const float n = rintf(x * 1.44269504f);
  float r = __fmaf_rn(-n, 0.693359375f, x);
  r = __fmaf_rn(-n, -2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = __fmaf_rn(p, r, 1.3981999507e-3f);
  p = __fmaf_rn(p, r, 8.3334519073e-3f);
  p = __fmaf_rn(p, r, 4.1665795894e-2f);
  p = __fmaf_rn(p, r, 1.6666665459e-1f);
  p = __fmaf_rn(p, r, 5.0000001201e-1f);
  const float y = __fmaf_rn(p, r * r, r) + 1.0f;
  return y * __int_as_float(((int)n + 127) << 23);
}

constexpr float kMeanSigmaL = 1.2232f;

// filterDepth for one pixel, reading its 5x5 neighbourhood from an LDS tile of row stride S
template <int S>
__device__ __forceinline__ float filter_depth_pixel(const float *in, int c) {
  const float z = in[c];
  if (z < 0.0f) return -1.0f;
  const float sigma_z = 1.0f / (0.0012f + 0.0019f * (z - 0.4f) * (z - 0.4f) + 0.0001f / sqrtf(z) * 0.25f);
  float final_depth = 0.0f, w_sum = 0.0f;
#pragma unroll
  for (int i = -2; i <= 2; i++)
#pragma unroll
    for (int j = -2; j <= 2; j++) {
      const float tmpz = in[c + j + i * S];
      if (tmpz < 0.0f) continue;
      float dz = tmpz - z;
      dz *= dz;
      const float w = det_exp(-0.5f * ((float)(abs(i) + abs(j)) * kMeanSigmaL * kMeanSigmaL + dz * sigma_z * sigma_z));
      w_sum += w;
      final_depth += w * tmpz;
    }
  return final_depth / w_sum;
}

// UpdateView with useBilateralFilter, all of it in one launch: depth = convert(raw); then
//   filter(float_image <- depth); filter(depth <- float_image); ... five passes; depth = float_image.
// Upstream's filter writes only the interior [2, W-2) x [2, H-2) and its float_image is zero-initialised, so the
// result has a 2-pixel border of 0.0 (= invalid for the fusion path) and passes 2 and 4 see zeros as neighbours
// there.  Five launches take 180 us for 640x480; instead a workgroup keeps a 16x16 output tile plus the 10-pixel
// halo the five passes need in two LDS images and ping-pongs between them, the valid region shrinking by 2 per
// pass (one read of the raw image, one write of the result; 86 us).  The filter is arithmetic-bound (25 taps x 5
// passes x ~40 instructions per pixel), so the tile is small: 1200 workgroups fill the SIMDs where 32x32 tiles
// (300 workgroups, 1.6x instead of 2.3x redundant arithmetic) left them one wave each and took 118 us.
constexpr int kFilterTile = 16, kFilterHalo = 10, kFilterSpan = kFilterTile + 2 * kFilterHalo;

__global__ __launch_bounds__(256) void k_bilateral5(const short *__restrict__ raw, float *__restrict__ out, int W, int H,
                                                    float a, float b) {
  constexpr int S = kFilterSpan;
  __shared__ float A[S * S], B[S * S];
  const int gx0 = blockIdx.x * kFilterTile - kFilterHalo, gy0 = blockIdx.y * kFilterTile - kFilterHalo;
  for (int c = threadIdx.x; c < S * S; c += blockDim.x) {
    const int gx = gx0 + c % S, gy = gy0 + c / S;
    float d = -1.0f;
    if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
      const int r = raw[gx + (size_t)gy * W];
      d = (r <= 0 || r > 32000) ? -1.0f : (float)r * a + b;
    }
    A[c] = d;     // view->depth: converted depth everywhere
    B[c] = 0.0f;  // float_image: zero wherever no pass writes
  }
  __syncthreads();
#pragma unroll 1
  for (int pass = 1; pass <= 5; pass++) {
    const float *src = (pass & 1) ? A : B;
    float *dst = (pass & 1) ? B : A;
    const int m = 2 * pass, n = S - 2 * m;  // cells [m, S - m)^2 are computable from the previous pass
    for (int k = threadIdx.x; k < n * n; k += blockDim.x) {
      const int tx = m + k % n, ty = m + k / n;
      const int gx = gx0 + tx, gy = gy0 + ty;
      if (gx >= 2 && gx < W - 2 && gy >= 2 && gy < H - 2) dst[tx + ty * S] = filter_depth_pixel<S>(src, tx + ty * S);
    }
    __syncthreads();
  }
  for (int k = threadIdx.x; k < kFilterTile * kFilterTile; k += blockDim.x) {
    const int tx = kFilterHalo + k % kFilterTile, ty = kFilterHalo + k / kFilterTile;
    const int gx = gx0 + tx, gy = gy0 + ty;
    if (gx < W && gy < H) out[gx + (size_t)gy * W] = B[tx + ty * S];
  }
}

int launch_bilateral(dslam_engine *e, dslam_view *v) {
  const int W = v->w_d, H = v->h_d;
  hipLaunchKernelGGL(k_bilateral5, dim3((W + kFilterTile - 1) / kFilterTile, (H + kFilterTile - 1) / kFilterTile), dim3(256), 0,
                     e->stream, v->raw_src, v->depth, W, H, v->affine_a, v->affine_b);
  DSLAM_HIP(hipGetLastError());
  v->depth_dirty = false;
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// dataset wire formats (PrecomputedDepthProvider::ReadPrecomputed's loop; FloatDepthmapToShort / ...ToInt16)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ short wrap_i16(float f) { return (short)(int)f; }  // float -> int32 (truncate) -> low 16 bits

__global__ __launch_bounds__(256) void k_dataset_depth(short *__restrict__ depth, int n, int format, float max_x256,
                                                       short max_mm_s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  short d = depth[i];
  if (format == DSLAM_DEPTH_KITTI_X256) {
    if ((float)d > max_x256) d = 0;
    d = wrap_i16((float)d * 3.90625f);  // kitti_factor = 1000.0 / 256.0
  } else {
    d = (short)(int)((double)(float)d / 5.0);
    if (d > max_mm_s) d = 0;
  }
  depth[i] = d;
}

int launch_dataset_depth(dslam_engine *e, short *depth_dev, int n, int format, float max_depth_m) {
  const float max_mm_f = max_depth_m * 1000.0f;  // GetMaxDepthMeters() * kMetersToMillimeters
  const short max_mm_s = (short)(int)roundf(max_mm_f);
  hipLaunchKernelGGL(k_dataset_depth, dim3((n + 255) / 256), dim3(256), 0, e->stream, depth_dev, n, format,
                     max_depth_m * 256.0f, max_mm_s);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

__global__ __launch_bounds__(256) void k_depth_to_int16(const float *__restrict__ depth, short *__restrict__ out, int n, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = wrap_i16(depth[i] * scale);
}

int launch_depth_to_int16(dslam_engine *e, const float *depth_dev, short *out_dev, int n, int scale) {
  hipLaunchKernelGGL(k_depth_to_int16, dim3((n + 255) / 256), dim3(256), 0, e->stream, depth_dev, out_dev, n, (float)scale);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// ---------------------------------------------------------------------------------------------------------
// depthPostProcessing
// ---------------------------------------------------------------------------------------------------------
struct PostParams {
  int rows, cols;
  float fx, fy, cx, cy, inv_fx, inv_fy;
  float r[9], t[3];  // Tpc = prev_pose^-1 * curr_pose: rotation row-major, translation
  float threshold, area_rows;
};

__device__ __forceinline__ int d2i_sat(double v) {
  // C's double -> int conversion is undefined outside int's range (and for NaN); the restatement saturates and
  // maps NaN to INT_MIN, both of which fail the bounds test that follows, like any out-of-image projection
  if (!(v == v)) return INT_MIN;
  if (v >= 2147483647.0) return INT_MAX;
  if (v <= -2147483648.0) return INT_MIN;
  return (int)v;
}

// The reference indexes the image with (row, col) but feeds `row` into the x (cx, fx) terms and `col` into the y
// terms, and projects to (row_u, col_v) the same way [REF DenseSlam.cpp:500-513]; kept as is -- a drop-in must
// blank the same pixels.  cv::Mat products of CV_32F matrices accumulate in double and round once (OpenCV
// GEMMSingleMul<float,double>), the translation is added in float.
__global__ __launch_bounds__(256) void k_depth_post(short *__restrict__ curr, const unsigned short *__restrict__ prev,
                                                    PostParams p, int *__restrict__ count) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), row = blockIdx.y * 4 + (threadIdx.x >> 6);
  bool counted = false;
  if (col < p.cols && row < p.rows) {
    const int idx = row * p.cols + col;
    const float z = (float)((double)(float)curr[idx] / 1000.0);
    if (!((double)z < 0.005)) {
      const float X = z * ((float)row - p.cx) * p.inv_fx;
      const float Y = z * ((float)col - p.cy) * p.inv_fy;
      float P[3];
      for (int k = 0; k < 3; k++) {
        const double acc = (double)p.r[3 * k] * (double)X + (double)p.r[3 * k + 1] * (double)Y + (double)p.r[3 * k + 2] * (double)z;
        P[k] = (float)acc + p.t[k];
      }
      const int row_u = d2i_sat((double)(p.fx * P[0]) * (1.0 / (double)P[2]) + (double)p.cx + 0.5);
      const int col_v = d2i_sat((double)(p.fy * P[1]) * (1.0 / (double)P[2]) + (double)p.cy + 0.5);
      if (!(row_u < 1 || col_v < 1 || row_u >= p.rows || col_v >= p.cols)) {
        const float prev_z = (float)((double)(float)prev[row_u * p.cols + col_v] / 1000.0);
        if (!((double)prev_z < 0.005)) {
          const float diff = fabsf(prev_z - P[2]);
          if (diff / P[2] > p.threshold && (float)row > p.area_rows) curr[idx] = 0;
          counted = true;
        }
      }
    }
  }
  const unsigned long long m = __ballot(counted);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}

int launch_depth_post(dslam_engine *e, short *curr_dev, const unsigned short *prev_dev, int cols, int rows,
                      const float *Tpc, const float *intr, float threshold, float area, int *count_dev) {
  PostParams p;
  p.rows = rows; p.cols = cols;
  p.fx = intr[0]; p.fy = intr[1]; p.cx = intr[2]; p.cy = intr[3];
  p.inv_fx = (float)(1.0 / (double)p.fx); p.inv_fy = (float)(1.0 / (double)p.fy);  // float inv_fx = 1.0/fx  [REF DenseSlam.cpp:440-441]
  // Tpc arrives column-major (the ABI's matrix convention): element (r, c) = Tpc[c*4 + r]
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) p.r[3 * r + c] = Tpc[c * 4 + r];
    p.t[r] = Tpc[12 + r];
  }
  p.threshold = threshold;
  p.area_rows = area * (float)rows;
  DSLAM_HIP(hipMemsetAsync(count_dev, 0, sizeof(int), e->stream));
  hipLaunchKernelGGL(k_depth_post, dim3((cols + 63) / 64, (rows + 3) / 4), dim3(256), 0, e->stream, curr_dev, prev_dev, p, count_dev);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

}  // namespace dslam
