// track.hip -- the depth tracker (ICP) that precedes the fusion path when the reference runs without ORB-SLAM2
// odometry: trackingController->Track(trackingState, view) [REF InfiniTamDriver.h:151-163; DenseSlam.cpp:200-206],
// i.e. upstream InfiniTAM v2's ITMDepthTracker::TrackCamera.
//
// Device work: the depth pyramid (FilterSubsampleWithHoles) and, per iteration, the sums of ComputeGandH -- one
// lane per pixel evaluates the point-to-plane residual and its Jacobian row exactly as the per-pixel function does
// (float), the 28 sums are accumulated in double (per lane, wave shuffle tree, LDS, one partial per workgroup) and
// the workgroup partials are added on the host in index order: deterministic, and equal to the oracle's double
// accumulation up to the last bit of a double.  The 6x6 Levenberg-Marquardt step, the pose update and the
// convergence test are a few dozen flops per iteration and stay on the host, like upstream's own CUDA tracker.
#include "dslam_internal.h"

namespace dslam {

__global__ __launch_bounds__(256) void k_subsample_with_holes(const float *__restrict__ in, int w, float *__restrict__ out,
                                                              int nw, int nh) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= nw || y >= nh) return;
  float acc = 0.0f, good = 0.0f;
#pragma unroll
  for (int dy = 0; dy < 2; dy++)
#pragma unroll
    for (int dx = 0; dx < 2; dx++) {
      const float p = in[(2 * x + dx) + (size_t)(2 * y + dy) * w];
      if (p > 0.0f) { acc += p; good += 1.0f; }
    }
  if (good > 0.0f) acc /= good;
  out[x + (size_t)y * nw] = acc;
}

struct IcpParams {
  const float *depth;
  int lw, lh;
  float vfx, vfy, vcx, vcy;   // view intrinsics at this level
  int sw, sh;
  float sfx, sfy, scx, scy;   // scene intrinsics (level 0)
  Mat4 approxInvPose, scenePose;
  const float4 *points, *normals;
  float dist_thresh;
  double *partials;           // [gridDim.x][kIcpSums]
};

constexpr int kIcpSums = 29;  // 21 Hessian (lower triangle) + 6 gradient + f + valid count

__device__ __forceinline__ float4 bilinear_with_holes(const float4 *src, float px, float py, int w) {
  const int ix = (short)floorf(px), iy = (short)floorf(py);
  const float dx = px - (float)ix, dy = py - (float)iy;
  const float4 a = src[ix + (size_t)iy * w], b = src[(ix + 1) + (size_t)iy * w];
  const float4 c = src[ix + (size_t)(iy + 1) * w], d = src[(ix + 1) + (size_t)(iy + 1) * w];
  if (a.w < 0 || b.w < 0 || c.w < 0 || d.w < 0) return make_float4(0, 0, 0, -1.0f);
  float4 r;
  r.x = a.x * (1.0f - dx) * (1.0f - dy) + b.x * dx * (1.0f - dy) + c.x * (1.0f - dx) * dy + d.x * dx * dy;
  r.y = a.y * (1.0f - dx) * (1.0f - dy) + b.y * dx * (1.0f - dy) + c.y * (1.0f - dx) * dy + d.y * dx * dy;
  r.z = a.z * (1.0f - dx) * (1.0f - dy) + b.z * dx * (1.0f - dy) + c.z * (1.0f - dx) * dy + d.z * dx * dy;
  r.w = a.w * (1.0f - dx) * (1.0f - dy) + b.w * dx * (1.0f - dy) + c.w * (1.0f - dx) * dy + d.w * dx * dy;
  return r;
}

// computePerPointGH_Depth for TYPE (1 rotation, 2 translation, 3 both)
template <int TYPE>
__global__ __launch_bounds__(256) void k_icp_gh(IcpParams p) {
  constexpr int NP = (TYPE == DSLAM_TRACKER_ITERATION_BOTH) ? 6 : 3;
  constexpr int NH = NP * (NP + 1) / 2;
  double sH[NH], sN[NP], sF = 0.0;
  int valid = 0;
#pragma unroll
  for (int i = 0; i < NH; i++) sH[i] = 0.0;
#pragma unroll
  for (int i = 0; i < NP; i++) sN[i] = 0.0;
  const int npix = p.lw * p.lh;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
    const int y = i / p.lw, x = i - y * p.lw;
    const float depth = p.depth[i];
    if (depth <= 1e-8f) continue;
    Vec4 pt;
    pt.x = depth * (((float)x - p.vcx) / p.vfx);
    pt.y = depth * (((float)y - p.vcy) / p.vfy);
    pt.z = depth;
    pt.w = 1.0f;
    pt = mul(p.approxInvPose, pt);
    pt.w = 1.0f;
    const Vec4 q = mul(p.scenePose, pt);
    if (q.z <= 0.0f) continue;
    const float u = p.sfx * q.x / q.z + p.scx, v = p.sfy * q.y / q.z + p.scy;
    if (!((u >= 0.0f) && (u <= (float)(p.sw - 2)) && (v >= 0.0f) && (v <= (float)(p.sh - 2)))) continue;
    const float4 cp = bilinear_with_holes(p.points, u, v, p.sw);
    if (cp.w < 0.0f) continue;
    const float ddx = cp.x - pt.x, ddy = cp.y - pt.y, ddz = cp.z - pt.z;
    const float dist = ddx * ddx + ddy * ddy + ddz * ddz;
    if (dist > p.dist_thresh) continue;
    const float4 n = bilinear_with_holes(p.normals, u, v, p.sw);
    const float b = n.x * ddx + n.y * ddy + n.z * ddz;
    float A[NP];
    const float r0 = +pt.z * n.y - pt.y * n.z, r1 = -pt.z * n.x + pt.x * n.z, r2 = +pt.y * n.x - pt.x * n.y;
    if (TYPE == DSLAM_TRACKER_ITERATION_ROTATION) { A[0] = r0; A[1] = r1; A[2] = r2; }
    else if (TYPE == DSLAM_TRACKER_ITERATION_TRANSLATION) { A[0] = n.x; A[1] = n.y; A[2] = n.z; }
    else { A[0] = r0; A[1] = r1; A[2] = r2; A[3] = n.x; A[4] = n.y; A[5] = n.z; }
    valid++;
    sF += (double)(b * b);
#pragma unroll
    for (int k = 0, c = 0; k < NP; k++) {
      sN[k] += (double)(b * A[k]);
#pragma unroll
      for (int j = 0; j <= k; j++, c++) sH[c] += (double)(A[k] * A[j]);
    }
  }
  // workgroup reduction in a fixed order: wave shuffle tree, then the four wave partials through LDS
  __shared__ double red[4][kIcpSums];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double vals[kIcpSums];
#pragma unroll
  for (int i = 0; i < kIcpSums; i++) vals[i] = 0.0;
#pragma unroll
  for (int i = 0; i < NH; i++) vals[i] = sH[i];
#pragma unroll
  for (int i = 0; i < NP; i++) vals[21 + i] = sN[i];
  vals[27] = sF;
  vals[28] = (double)valid;
#pragma unroll
  for (int i = 0; i < kIcpSums; i++) {
    if (i >= NH && i < 21) continue;              // unused Hessian slots of the 3-parameter variants
    if (i >= 21 + NP && i < 27) continue;
    double v = vals[i];
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < kIcpSums) {
    const int i = threadIdx.x;
    const bool used = !((i >= NH && i < 21) || (i >= 21 + NP && i < 27));
    p.partials[(size_t)blockIdx.x * kIcpSums + i] = used ? ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) : 0.0;
  }
}

// ---- host side of TrackCamera ----------------------------------------------------------------------------------
namespace {
// Cholesky solve of the (damped) normal equations, float like ORUtils::Cholesky
void cholesky_solve(const float *Ain, int n, const float *bvec, float *x) {
  float L[36];
  for (int i = 0; i < n * n; i++) L[i] = Ain[i];
  for (int c = 0; c < n; c++) {
    float inv_diag = 1.0f;
    for (int r = c; r < n; r++) {
      float val = L[c + r * n];
      for (int c2 = 0; c2 < c; c2++) val -= L[c + c2 * n] * L[c2 + r * n];
      if (r == c) { L[c + r * n] = val; inv_diag = (val == 0.0f) ? 0.0f : 1.0f / val; }
      else { L[r + c * n] = val; L[c + r * n] = val * inv_diag; }
    }
  }
  float yv[6];
  for (int i = 0; i < n; i++) {
    float val = bvec[i];
    for (int j = 0; j < i; j++) val -= L[j + i * n] * yv[j];
    yv[i] = val;
  }
  for (int i = 0; i < n; i++) yv[i] = (L[i + i * n] == 0.0f) ? 0.0f : yv[i] / L[i + i * n];
  for (int i = n - 1; i >= 0; i--) {
    float val = yv[i];
    for (int j = i + 1; j < n; j++) val -= L[i + j * n] * x[j];
    x[i] = val;
  }
}

// ITMPose::Coerce stand-in: Gram-Schmidt on the rotation columns, bottom row (0,0,0,1)
void coerce_pose(float *M) {
  double c0[3] = {M[0], M[1], M[2]}, c1[3] = {M[4], M[5], M[6]}, c2[3];
  double n = sqrt(c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
  if (n > 0) for (double &v : c0) v /= n;
  const double d = c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2];
  for (int i = 0; i < 3; i++) c1[i] -= d * c0[i];
  n = sqrt(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
  if (n > 0) for (double &v : c1) v /= n;
  c2[0] = c0[1] * c1[2] - c0[2] * c1[1]; c2[1] = c0[2] * c1[0] - c0[0] * c1[2]; c2[2] = c0[0] * c1[1] - c0[1] * c1[0];
  for (int i = 0; i < 3; i++) { M[i] = (float)c0[i]; M[4 + i] = (float)c1[i]; M[8 + i] = (float)c2[i]; }
  M[3] = M[7] = M[11] = 0.0f; M[15] = 1.0f;
}

constexpr int kIcpGrid = 240;
}  // namespace

int launch_track_camera(dslam_engine *e, const dslam_view *v, dslam_render_state *r, const float *scenePose, float *pose_M,
                        const float *intr, const dslam_tracker_params *tp, dslam_tracker_result *res) {
  const int levels = tp->no_hierarchy_levels;
  DSLAM_REQUIRE(levels >= 1 && levels <= DSLAM_TRACKER_MAX_LEVELS && tp->no_icp_run_till_level >= 0, "bad tracker parameters");
  DSLAM_REQUIRE(r->icp_points && r->icp_normals, "no ICP maps: call dslam_create_icp_maps first");
  DSLAM_REQUIRE(v->w_d == r->w && v->h_d == r->h, "view and render state sizes differ");
  int rc = ensure_view_depth(e, v);
  if (rc) return rc;
  // depth pyramid (levels 1.. live in one buffer of the view) and the partial-sum buffers
  if (!v->pyramid) DSLAM_HIP(hipMalloc(&v->pyramid, (size_t)v->w_d * v->h_d * sizeof(float)));  // sum of levels 1.. < 1/3
  if (!e->icp_partials_host) {
    // the workgroup partials (240 x 29 doubles) are written straight into mapped pinned host memory: one stream
    // synchronise per iteration instead of a copy launch plus a synchronise
    DSLAM_HIP(hipHostMalloc((void **)&e->icp_partials_host, (size_t)kIcpGrid * kIcpSums * sizeof(double), hipHostMallocMapped));
    DSLAM_HIP(hipHostGetDevicePointer((void **)&e->icp_partials, e->icp_partials_host, 0));
  }
  const float *ldepth[DSLAM_TRACKER_MAX_LEVELS];
  int lw[DSLAM_TRACKER_MAX_LEVELS], lh[DSLAM_TRACKER_MAX_LEVELS];
  float lintr[DSLAM_TRACKER_MAX_LEVELS][4];
  ldepth[0] = v->depth; lw[0] = v->w_d; lh[0] = v->h_d;
  for (int k = 0; k < 4; k++) lintr[0][k] = intr[k];
  float *next = v->pyramid;
  for (int i = 1; i < levels; i++) {
    lw[i] = lw[i - 1] / 2; lh[i] = lh[i - 1] / 2;
    DSLAM_REQUIRE(lw[i] > 0 && lh[i] > 0, "too many hierarchy levels for this image size");
    hipLaunchKernelGGL(k_subsample_with_holes, dim3((lw[i] + 31) / 32, (lh[i] + 7) / 8), dim3(256), 0, e->stream, ldepth[i - 1],
                       lw[i - 1], next, lw[i], lh[i]);
    ldepth[i] = next;
    next += (size_t)lw[i] * lh[i];
    for (int k = 0; k < 4; k++) lintr[i][k] = lintr[i - 1][k] * 0.5f;
  }
  int iters_per_level[DSLAM_TRACKER_MAX_LEVELS];
  float dist_per_level[DSLAM_TRACKER_MAX_LEVELS];
  iters_per_level[0] = 2;
  for (int i = 1; i < levels; i++) iters_per_level[i] = iters_per_level[i - 1] + 2;
  const float dstep = tp->dist_thresh / levels;
  dist_per_level[levels - 1] = tp->dist_thresh;
  for (int i = levels - 2; i >= 0; i--) dist_per_level[i] = dist_per_level[i + 1] - dstep;

  float M[16], approxInvPose[16];
  memcpy(M, pose_M, 64);
  float hessian_good[36] = {0}, nabla_good[6] = {0};
  int total_iters = 0, last_valid = 0;
  float last_f = 0.0f;
  for (int level = levels - 1; level >= tp->no_icp_run_till_level; level--) {
    const int type = tp->regime[level];
    if (type == DSLAM_TRACKER_ITERATION_NONE) continue;
    DSLAM_REQUIRE(type >= DSLAM_TRACKER_ITERATION_ROTATION && type <= DSLAM_TRACKER_ITERATION_BOTH, "bad iteration type");
    const int npara = (type == DSLAM_TRACKER_ITERATION_BOTH) ? 6 : 3;
    if (!invert_matrix(M, approxInvPose)) { set_last_error("pose matrix is singular"); return DSLAM_ERR_INVALID; }
    float good_M[16];
    memcpy(good_M, M, 64);
    float f_old = 1e20f, lambda = 1.0f;
    for (int it = 0; it < iters_per_level[level]; it++) {
      IcpParams ip;
      ip.depth = ldepth[level]; ip.lw = lw[level]; ip.lh = lh[level];
      ip.vfx = lintr[level][0]; ip.vfy = lintr[level][1]; ip.vcx = lintr[level][2]; ip.vcy = lintr[level][3];
      ip.sw = r->w; ip.sh = r->h;
      ip.sfx = lintr[0][0]; ip.sfy = lintr[0][1]; ip.scx = lintr[0][2]; ip.scy = lintr[0][3];
      memcpy(ip.approxInvPose.m, approxInvPose, 64);
      memcpy(ip.scenePose.m, scenePose, 64);
      ip.points = r->icp_points; ip.normals = r->icp_normals;
      ip.dist_thresh = dist_per_level[level];
      ip.partials = e->icp_partials;
      const int grid = std::min(kIcpGrid, (lw[level] * lh[level] + 255) / 256);
      if (type == DSLAM_TRACKER_ITERATION_ROTATION) hipLaunchKernelGGL((k_icp_gh<1>), dim3(grid), dim3(256), 0, e->stream, ip);
      else if (type == DSLAM_TRACKER_ITERATION_TRANSLATION) hipLaunchKernelGGL((k_icp_gh<2>), dim3(grid), dim3(256), 0, e->stream, ip);
      else hipLaunchKernelGGL((k_icp_gh<3>), dim3(grid), dim3(256), 0, e->stream, ip);
      DSLAM_HIP(hipGetLastError());
      DSLAM_HIP(hipStreamSynchronize(e->stream));
      double sums[kIcpSums];
      for (int i = 0; i < kIcpSums; i++) sums[i] = 0.0;
      for (int g = 0; g < grid; g++)
        for (int i = 0; i < kIcpSums; i++) sums[i] += e->icp_partials_host[(size_t)g * kIcpSums + i];
      const int valid = (int)sums[28];
      float hessian_new[36] = {0}, nabla_new[6] = {0};
      for (int k = 0, c = 0; k < npara; k++)
        for (int j = 0; j <= k; j++, c++) hessian_new[k + j * 6] = hessian_new[j + k * 6] = (float)sums[c];
      for (int k = 0; k < npara; k++) nabla_new[k] = (float)sums[21 + k];
      const float f_new = (valid > 100) ? sqrtf((float)sums[27]) / (float)valid : 1e5f;
      total_iters++; last_valid = valid; last_f = f_new;

      if (valid <= 0 || f_new > f_old) {
        memcpy(M, good_M, 64);
        invert_matrix(M, approxInvPose);
        lambda *= 10.0f;
      } else {
        memcpy(good_M, M, 64);
        f_old = f_new;
        for (int i = 0; i < 36; i++) hessian_good[i] = hessian_new[i] / (float)valid;
        for (int i = 0; i < 6; i++) nabla_good[i] = nabla_new[i] / (float)valid;
        lambda /= 10.0f;
      }
      float A6[36];
      for (int i = 0; i < 36; i++) A6[i] = hessian_good[i];
      for (int i = 0; i < 6; i++) A6[i + i * 6] *= 1.0f + lambda;
      float step[6] = {0, 0, 0, 0, 0, 0};
      if (npara == 3) {
        float small[9];
        for (int rr = 0; rr < 3; rr++) for (int cc = 0; cc < 3; cc++) small[rr + cc * 3] = A6[rr + cc * 6];
        cholesky_solve(small, 3, nabla_good, step);
      } else {
        cholesky_solve(A6, 6, nabla_good, step);
      }
      float s6[6] = {0, 0, 0, 0, 0, 0};  // ApplyDelta
      if (type == DSLAM_TRACKER_ITERATION_ROTATION) { s6[0] = step[0]; s6[1] = step[1]; s6[2] = step[2]; }
      else if (type == DSLAM_TRACKER_ITERATION_TRANSLATION) { s6[3] = step[0]; s6[4] = step[1]; s6[5] = step[2]; }
      else for (int i = 0; i < 6; i++) s6[i] = step[i];
      const float Tinc[16] = {1.0f, -s6[2], s6[1], 0.0f, s6[2], 1.0f, -s6[0], 0.0f, -s6[1], s6[0], 1.0f, 0.0f, s6[3], s6[4], s6[5], 1.0f};
      float nextInv[16];
      for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++) {
          float acc = 0;
          for (int k = 0; k < 4; k++) acc += Tinc[k * 4 + rr] * approxInvPose[c * 4 + k];
          nextInv[c * 4 + rr] = acc;
        }
      invert_matrix(nextInv, M);   // pose_d->SetInvM(approxInvPose)
      coerce_pose(M);              // pose_d->Coerce()
      invert_matrix(M, approxInvPose);
      float len = 0.0f;
      for (int i = 0; i < 6; i++) len += step[i] * step[i];
      if (sqrtf(len) / 6 < tp->termination_threshold) break;  // HasConverged
    }
  }
  memcpy(pose_M, M, 64);
  if (res) { res->iterations = total_iters; res->valid_points_last = last_valid; res->f_last = last_f; res->pad = 0; }
  return DSLAM_OK;
}

}  // namespace dslam
