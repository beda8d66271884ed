// integrate.hip -- IntegrateIntoScene and its inverse (de-integration) for gfx950: the bandwidth kernel.
//
// Reference call sites: denseMapper->ProcessFrame -> IntegrateIntoScene (InfiniTamDriver.h:187-192,
// DenseSlam.cpp:213,236,403) and denseMapper->DeProcessFrame (InfiniTamDriver.h:194-199, DenseSlam.cpp:393,425).
// Algorithm: SURVEY.md Appendix A.5 / A.11.
//
// Mapping (wave64, no MFMA -- this is streaming read-modify-write):
//   one wavefront owns one 8x8x8 voxel block = 4 KiB.  It moves the block as 4 x global_load_dwordx4 per lane
//   (1 KiB per wave instruction, the widest coalesced shape), so a lane holds 8 voxels:
//       load j (0..3), lane l  ->  voxels x in {2(l&3), 2(l&3)+1}, y = (l>>2)&7, z = 2j + (l>>5)
//   A wave takes G consecutive entries of the visible list at a time: lanes 0..G-1 gather the G hash entries
//   with one vector load each (16 B, the table stays in the Infinity Cache) and the wave then walks them with
//   v_readlane -- no per-block dependent scalar-load chain.  Depth / RGB gathers hit L2 (1.2 MB images).
//   Only 16-byte chunks that changed are written back.
//   Grid = fixed 512 workgroups of 16 waves (one full residency wave of the 256 CUs), grid-stride over the
//   visible list whose length is read from device memory (no host round trip after allocation).
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>

#include "dslam_internal.h"

#pragma clang fp contract(off)

// -DDSLAM_PACKED=0 builds the fusion path voxel by voxel (the scalar form the packed one must match bit for bit;
// used for A/B timing)
#ifndef DSLAM_PACKED
#define DSLAM_PACKED 1
#endif
// -DDSLAM_COLOUR_QUEUE=0: the colour update inline in every lane's voxel loop (the form the queued one must match; A/B timing)
#ifndef DSLAM_COLOUR_QUEUE
#define DSLAM_COLOUR_QUEUE 1
#endif
namespace dslam {

// Voxel chunks of a launch that is larger than the Infinity Cache (>= push_job_min = 65536 visible blocks = 256 MiB) are read and
// written with the non-temporal policy (template parameter STREAM of k_integrate): nothing of such a launch is still cached when
// somebody comes back for it, and the S-stress launch is 8 % shorter (0.533 -> 0.576 of the peak, three alternations on one box;
// nt loads alone 0.542, nt stores alone 0.533, write-through `sc1` stores 0.529).  Launches of a real sequence's size keep the
// default policy -- the ray march that follows reads these lines: nt loads on the bench scene leave the launch at 21.0 us and
// make the frame 148 us instead of 140.  The policy is part of the instruction, so the HOST chooses the instantiation, by the
// visible count it last heard of (dslam_render_state::vis_hint: a page-locked word the allocation sweep writes; no wait).  A
// flag tested inside ONE kernel was tried first: the plain kernel has no register for it (12 bytes of scratch per lane, the
// bench launch +0.9 us), and the block loop written twice behind one test doubled the scalar spills.
typedef unsigned vox_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void vox_load2(uint4 &v0, uint4 &v1, const uint4 *a0, const uint4 *a1) {
  const vox_v4 r0 = __builtin_nontemporal_load(reinterpret_cast<const vox_v4 *>(a0));
  const vox_v4 r1 = __builtin_nontemporal_load(reinterpret_cast<const vox_v4 *>(a1));
  v0 = make_uint4(r0.x, r0.y, r0.z, r0.w);
  v1 = make_uint4(r1.x, r1.y, r1.z, r1.w);
}
__device__ __forceinline__ void vox_store(uint4 *a, const uint4 &v) {
  const vox_v4 r = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(r, reinterpret_cast<vox_v4 *>(a));
}


struct IntegrateParams {
  const int *visible_ids;
  const RenderCounters *rc;
  const HashEntry *hash;
  uint4 *voxels16;  // voxel blocks as 16-byte chunks (2 voxels)
  const float *depth;
  const uchar4 *rgba;
  int Wd, Hd, Wr, Hr;
  Mat4 M_d, M_rgb;
  float fx_d, fy_d, cx_d, cy_d;
  float fx_r, fy_r, cx_r, cy_r;
  float voxel_size, mu;
  float inv_32767, inv_255;  // correctly rounded reciprocals (host division) for div_exact()
  int same_cam;                      // RGB camera == depth camera (identity calib of the reference): reuse projection
  int max_w, stop_max;
  int depth_weighting, max_new_w;
  float max_distance;
  int shard, num_shards, chunk_blocks;
  int shard_first, shard_count;
  // visible-list ring push fused into this kernel (push_words == 0: off)
  unsigned long long *masks;
  int *last_seen;
  int push_words, push_ring, push_bit, push_frame;
  int push_job_min;  // visible blocks from which on the trailing workgroups queue the list instead of the block waves (kPushJobMin)
  int *timer_slot;  // bench instrumentation: where to record this launch's visible-block count (or null)
  unsigned char *dirty;  // sharded re-integration: per slot "visited since tracking began" (null: not tracked)
  const short4 *expect_pos;  // stored keyframe list: the block each listed entry held at fusion time (null: a live list)
  int spec_ids;  // the id list has a slot for every wave of the grid: a wave may read "its" id before the count is known
  // diagnostics only (env DSLAM_DBG_INTEGRATE=<file>, DIAG instantiation): per wave 16 words -- s_memrealtime (10 ns ticks)
  // at [0] entry, [1] table ready, [2] list length known, [3] entries gathered, then per half block [4 + 4h] chunk 0 and
  // [5 + 4h] chunk 1 updated, [6 + 4h] colour pass done, [7 + 4h] stores issued; [12], [13] shader clock at entry and end;
  // [14] voxel-block slot; [15] XCC id << 32 | HW_ID
  unsigned long long *dbg_waves;
};

// a / b for a divisor whose correctly rounded reciprocal y = RN(1/b) is known: q = RN(a*y), r = a - b*q (exact, FMA),
// q' = RN(q + r*y) -- 3 instructions instead of the ~10 of v_div_scale/v_rcp/v_div_fmas/v_div_fixup.  This equals
// the IEEE quotient RN(a/b) only for suitable divisors: tests/tools/verify_exact_div.cpp shows 0 mismatches over
// EVERY finite float a for b = 32767 and b = 255 and for the integer weights b = 1..256 over the operand range, but
// 0.06-0.2 % one-ulp mismatches for b = 0.2, 0.02, ... -- so it is used for those constants only, never for mu.
__device__ __forceinline__ float div_exact(float a, float b, float y) {
  const float q = a * y;
  const float r = __fmaf_rn(-b, q, a);
  return __fmaf_rn(r, y, q);
}

// IEEE division without the scaling / fix-up instructions, two quotients at a time.  hipcc expands a float division
// into  d' = div_scale(b), n' = div_scale(a), r = rcp(d'), two FMAs refining r, q = n' r, two residual corrections,
// div_fmas, div_fixup  (10 VALU instructions).  div_scale / div_fmas / div_fixup only act when an exponent is near
// the ends of the float range or an operand is 0, inf or NaN; for the operands of this kernel (depths of 0.1-100 m,
// image coordinates, eta / mu) they pass their inputs through, so the remaining arithmetic -- the same instructions,
// in the same order, with the same roundings -- gives the same bits.  Written on 2-vectors it compiles to v_rcp_f32
// x 2 + 7 packed instructions for TWO quotients: 9 instead of 20.  dslam_selftest_division() compares it with the
// native division over random operands of these ranges on the device (tests/test_gpu_parity.py).
typedef float __attribute__((ext_vector_type(2))) f2;

__device__ __forceinline__ f2 div_ieee2(f2 a, f2 b) {
  f2 r;
  r.x = __builtin_amdgcn_rcpf(b.x);
  r.y = __builtin_amdgcn_rcpf(b.y);
  const f2 nb = -b;
  const f2 one = {1.0f, 1.0f};
  const f2 e = __builtin_elementwise_fma(nb, r, one);
  r = __builtin_elementwise_fma(e, r, r);
  f2 q = a * r;
  f2 t = __builtin_elementwise_fma(nb, q, a);
  q = __builtin_elementwise_fma(t, r, q);
  t = __builtin_elementwise_fma(nb, q, a);
  return __builtin_elementwise_fma(t, r, q);
}

// RN(1 / i) for the integer weights of the table: the hardware reciprocal (1 ulp) and one Newton step.  Equal to the IEEE
// quotient 1.0f / i for every i of the table -- for every i up to 65535 in fact (k_selftest_division counts the
// differences on the device, tests/test_gpu_parity.py) -- in 3 instructions instead of the 12 of a division, at the top
// of a kernel whose 8192 waves all start with this.
__device__ __forceinline__ float recip_table_entry(int i) {
  const float b = (float)i;
  const float y = __builtin_amdgcn_rcpf(b);
  return __fmaf_rn(__fmaf_rn(-b, y, 1.0f), y, y);
}

// random operands in the kernel's ranges: projection (|a| in 2^[-20,24], b in 2^[-10,10]) and eta / mu
// (|a| in 2^[-30,8], b in [2^-10, 1])
__global__ __launch_bounds__(256) void k_selftest_division(unsigned long long seed, int per_thread, unsigned long long *mismatches) {
  unsigned long long x = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * 256ull + threadIdx.x + 1);
  auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (unsigned)(x >> 16); };
  auto make = [&](int e_lo, int e_hi, bool may_be_negative) {
    const unsigned r = next();
    const unsigned mant = r & 0x7fffffu;
    const unsigned ex = (unsigned)(127 + e_lo + (int)((r >> 23) % (unsigned)(e_hi - e_lo + 1)));
    const unsigned sign = may_be_negative ? (next() & 1u) << 31 : 0u;
    return __uint_as_float(sign | (ex << 23) | mant);
  };
  unsigned bad = 0;
  for (int i = 0; i < per_thread; i++) {
    f2 a, b;
    a.x = make(-20, 24, true); b.x = make(-10, 10, false);
    a.y = make(-30, 8, true);  b.y = make(-10, 0, false);
    const f2 q = div_ieee2(a, b);
    const float n0 = a.x / b.x, n1 = a.y / b.y;
    bad += (__float_as_uint(q.x) != __float_as_uint(n0)) + (__float_as_uint(q.y) != __float_as_uint(n1));
  }
  // the reciprocal table of k_integrate: every entry against the IEEE division (entry 0 is never read)
  if (blockIdx.x == 0)
    for (int i = 1 + (int)threadIdx.x; i < 65536; i += 256)
      bad += __float_as_uint(recip_table_entry(i)) != __float_as_uint(1.0f / (float)i);
  if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

int launch_selftest_division(dslam_engine *e, long long samples, unsigned long long *mismatches_dev) {
  const int per_thread = 4096;
  const long long threads = (samples / 2 + per_thread - 1) / per_thread;
  const int blocks = (int)std::max(1LL, (threads + 255) / 256);
  DSLAM_HIP(hipMemsetAsync(mismatches_dev, 0, sizeof(unsigned long long), e->stream));
  hipLaunchKernelGGL(k_selftest_division, dim3(blocks), dim3(256), 0, e->stream, 0x243F6A8885A308D3ull, per_thread, mismatches_dev);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

constexpr int kInvTab = 512;  // reciprocals of the integer weights 1..511 (w_depth <= 255, newW <= 255: the ABI rejects
                              // max_new_w > 255, so w_depth + newW <= 510 always indexes inside the table)

// A voxel is skipped unless its camera-frame depth is a NORMAL positive float (upstream: `pt_camera.z <= 0` skips).
// A denormal z (> 0, < 1.18e-38 m: the voxel centre lies in the camera plane to within nothing) would be divided by:
// v_rcp_f32 has no denormal support (-> inf -> NaN in the scaling-free division below), and NaN passes every
// `u < 1 || u > W - 2` test, so the depth image would be indexed with (int)NaN.  Both the kernel and the oracle
// therefore treat a denormal z like z <= 0, and every bounds test is written so that NaN fails it.
constexpr float kMinCamZ = 1.17549435e-38f;  // FLT_MIN

__device__ __forceinline__ bool in_image(float u, float w, float umax, float wmax) {
  return u >= 1.0f && u <= umax && w >= 1.0f && w <= wmax;  // false for NaN
}

template <bool PLAIN = false>  // PLAIN: the weight is 1 by construction
__device__ __forceinline__ int new_weight(const IntegrateParams &p, float depth_measure) {
  if (PLAIN || !p.depth_weighting) return 1;
  const float dd = depth_measure < p.max_distance ? depth_measure : p.max_distance;
  const int w = (int)roundf((float)p.max_new_w * (1.0f - dd / p.max_distance));
  // (the upper clamp never acts on a measured depth > 0; it keeps the table index of an inactive lane, which
  // evaluates this on depth[0] whatever that holds, inside the table)
  return w < 1 ? 1 : (w > p.max_new_w ? p.max_new_w : w);
}

// The depth and colour images are read through buffer resources on the packed fusion path: a lane addresses its pixel
// with a 32-bit byte offset (pixel indices are < 2^24: one 24-bit multiply and one add-shift instead of 64-bit address
// arithmetic), and an offset outside the image -- only a lane whose voxel already failed a test can produce one --
// reads 0 instead of having to be replaced by a safe index first.  Worth 0.5 us of a 21 us launch (and it is what
// makes requesting the depth pixels of two chunks ahead of either update cheap, see pair_project).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const void *base, int W, int H) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, W * H * 4, 0x00020000);
}
__device__ __forceinline__ unsigned pixel_offset(int x, int y, int W) { return ((unsigned)__mul24(y, W) + (unsigned)x) << 2; }

// bilinear_rgb below with its four texels fetched through `rs` (same arithmetic, same order)
__device__ __forceinline__ void bilinear_rgb_buf(__amdgpu_buffer_rsrc_t rs, float px, float py, int W, float out[3]) {
  const int ix = (int)floorf(px), iy = (int)floorf(py);
  const float dx = px - (float)ix, dy = py - (float)iy;
  const unsigned o = pixel_offset(ix, iy, W), o2 = o + ((unsigned)W << 2);
  const unsigned a = __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b32(rs, o + 4u, 0, 0);
  const unsigned c = __builtin_amdgcn_raw_buffer_load_b32(rs, o2, 0, 0), d = __builtin_amdgcn_raw_buffer_load_b32(rs, o2 + 4u, 0, 0);
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float fa = (float)((a >> (8 * k)) & 0xffu), fb = (float)((b >> (8 * k)) & 0xffu);
    const float fc = (float)((c >> (8 * k)) & 0xffu), fd = (float)((d >> (8 * k)) & 0xffu);
    out[k] = (fa * (1.0f - dx) * (1.0f - dy) + fb * dx * (1.0f - dy) + fc * (1.0f - dx) * dy + fd * dx * dy);
  }
}

__device__ __forceinline__ void bilinear_rgb(const uchar4 *__restrict__ rgba, float px, float py, int W, float out[3]) {
  const int ix = (int)floorf(px), iy = (int)floorf(py);
  const float dx = px - (float)ix, dy = py - (float)iy;
  const uchar4 a = rgba[ix + iy * W], b = rgba[(ix + 1) + iy * W], c = rgba[ix + (iy + 1) * W],
               d = rgba[(ix + 1) + (iy + 1) * W];
  // ((a*(1-dx))*(1-dy)) in the reference's evaluation order: the products are NOT regrouped
  out[0] = ((float)a.x * (1.0f - dx) * (1.0f - dy) + (float)b.x * dx * (1.0f - dy) + (float)c.x * (1.0f - dx) * dy +
            (float)d.x * dx * dy);
  out[1] = ((float)a.y * (1.0f - dx) * (1.0f - dy) + (float)b.y * dx * (1.0f - dy) + (float)c.y * (1.0f - dx) * dy +
            (float)d.y * dx * dy);
  out[2] = ((float)a.z * (1.0f - dx) * (1.0f - dy) + (float)b.z * dx * (1.0f - dy) + (float)c.z * (1.0f - dx) * dy +
            (float)d.z * dx * dy);
}

// ComputeUpdatedVoxelInfo<hasColor>::compute on a packed voxel (lo, hi) whose camera-frame position pc = M_d * pm
// has been assembled by the caller.  Returns true if the voxel changed.
template <bool DEINT, bool SAME_CAM>
__device__ __forceinline__ bool update_voxel(unsigned &lo, unsigned &hi, const Vec4 &pc, const Vec4 &pm,
                                             const IntegrateParams &p, const float *inv_tab) {
  float eta, eta_mu, u, w;
  bool changed = false;
  {  // computeUpdatedVoxelDepthInfo
    if (!(pc.z >= kMinCamZ)) return false;
    u = p.fx_d * pc.x / pc.z + p.cx_d;
    w = p.fy_d * pc.y / pc.z + p.cy_d;
    if (!in_image(u, w, (float)(p.Wd - 2), (float)(p.Hd - 2))) return false;
    const float dm = p.depth[(int)(u + 0.5f) + (int)(w + 0.5f) * p.Wd];
    if (dm <= 0.0f) return false;
    eta = dm - pc.z;
    if (eta < -p.mu) return false;
    const float oldF = div_exact((float)(short)(lo & 0xffffu), 32767.0f, p.inv_32767);
    const int oldW = (int)((lo >> 16) & 0xffu);
    eta_mu = eta / p.mu;  // true IEEE division: the 3-instruction form is NOT exact for arbitrary mu
    float newF = fminf(1.0f, eta_mu);
    int newW = new_weight(p, dm);
    if (!DEINT) {
      newF = (float)oldW * oldF + (float)newW * newF;
      newW = oldW + newW;
      newF = div_exact(newF, (float)newW, inv_tab[newW]);
      newW = newW < p.max_w ? newW : p.max_w;
      const unsigned sdf = (unsigned)(unsigned short)float_to_sdf(newF);
      lo = (lo & 0xff000000u) | ((unsigned)newW << 16) | sdf;
      changed = true;
    } else if (oldW >= newW) {
      const int remW = oldW - newW;
      if (remW == 0) {
        lo = (lo & 0xff000000u) | 0x7fffu;
      } else {
        float F = div_exact((float)oldW * oldF - (float)newW * newF, (float)remW, inv_tab[remW]);
        F = fmaxf(-1.0f, fminf(1.0f, F));
        const unsigned sdf = (unsigned)(unsigned short)float_to_sdf(F);
        lo = (lo & 0xff000000u) | ((unsigned)remW << 16) | sdf;
      }
      changed = true;
    }
  }
  if ((eta > p.mu) || (fabsf(eta_mu) > 0.25f)) return changed;
  {  // computeUpdatedVoxelColorInfo
    if (!SAME_CAM) {
      const Vec4 pcr = mul(p.M_rgb, pm);
      u = p.fx_r * pcr.x / pcr.z + p.cx_r;
      w = p.fy_r * pcr.y / pcr.z + p.cy_r;
      if (!in_image(u, w, (float)(p.Wr - 2), (float)(p.Hr - 2))) return changed;
    }
    float m[3];
    bilinear_rgb(p.rgba, u, w, p.Wr, m);
    const unsigned oc[3] = {lo >> 24, hi & 0xffu, (hi >> 8) & 0xffu};
    const unsigned wc = (hi >> 16) & 0xffu;
    const float oldW = (float)wc;
    unsigned nc[3];
    unsigned new_wc;
    if (!DEINT) {
      float newW = oldW + 1.0f;
      const float inv_new = inv_tab[wc + 1];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const float oldC = div_exact((float)oc[k], 255.0f, p.inv_255);
        const float c = div_exact(m[k], 255.0f, p.inv_255);
        const float v = div_exact(oldC * oldW + c * 1.0f, newW, inv_new);
        nc[k] = (unsigned)(unsigned char)(v * 255.0f);
      }
      newW = fminf(newW, (float)p.max_w);
      new_wc = (unsigned)(unsigned char)newW;
    } else {
      if (wc < 1) return changed;
      const float remW = oldW - 1.0f;
      if (remW == 0.0f) {
        nc[0] = nc[1] = nc[2] = 0;
        new_wc = 0;
      } else {
        const float inv_rem = inv_tab[wc - 1];
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const float oldC = div_exact((float)oc[k], 255.0f, p.inv_255);
          const float c = div_exact(m[k], 255.0f, p.inv_255);
          float v = div_exact(oldC * oldW - c * 1.0f, remW, inv_rem);
          v = fmaxf(0.0f, fminf(1.0f, v));
          nc[k] = (unsigned)(unsigned char)(v * 255.0f);
        }
        new_wc = (unsigned)(unsigned char)remW;
      }
    }
    lo = (lo & 0x00ffffffu) | (nc[0] << 24);
    hi = (hi & 0xff000000u) | nc[1] | (nc[2] << 8) | (new_wc << 16);
    changed = true;
  }
  return changed;
}

// ---- two voxels at a time (fusion only) --------------------------------------------------------------------------
// SQ counters on MI355X: the kernel issues ~1060 VALU instructions per block-wave, i.e. 14 us of VALU issue per SIMD
// in a 24 us launch at V = 7.9k and ~90 % of the S-stress launch: it is bound by arithmetic, not by HBM.  The two
// voxels of a 16-byte chunk (x and x+1) run the same arithmetic on different data, and gfx950 has packed FP32
// (v_pk_fma/mul/add_f32: two lanes' worth per instruction), so the depth update is written on 2-vectors: one
// instruction stream for both voxels, every operation the same IEEE operation in the same order as the scalar code
// -- bit-identical results -- and the three divisions per voxel share 9-instruction packed sequences (div_ieee2).
// Per-voxel conditions become masks; the rare colour update and the de-integration keep the scalar path.
__device__ __forceinline__ f2 div_exact2(f2 a, float b, f2 y) {
  const f2 q = a * y;
  const f2 nb = {-b, -b};
  const f2 r = __builtin_elementwise_fma(nb, q, a);
  return __builtin_elementwise_fma(r, y, q);
}

// computeUpdatedVoxelColorInfo for one voxel (fusion); u, w = its projection into the colour image
// BUF: the texels come through a buffer resource (bilinear_rgb_buf)
template <bool SAME_CAM, bool BUF = false>
__device__ __forceinline__ bool fuse_colour(unsigned &lo, unsigned &hi, float u, float w, const Vec4 &pm,
                                            const IntegrateParams &p, const float *inv_tab) {
  if (!SAME_CAM) {
    const Vec4 pcr = mul(p.M_rgb, pm);
    u = p.fx_r * pcr.x / pcr.z + p.cx_r;
    w = p.fy_r * pcr.y / pcr.z + p.cy_r;
    if (!in_image(u, w, (float)(p.Wr - 2), (float)(p.Hr - 2))) return false;
  }
  float m[3];
  if constexpr (BUF) bilinear_rgb_buf(image_rsrc(p.rgba, p.Wr, p.Hr), u, w, p.Wr, m);
  else bilinear_rgb(p.rgba, u, w, p.Wr, m);
  const unsigned oc[3] = {lo >> 24, hi & 0xffu, (hi >> 8) & 0xffu};
  const unsigned wc = (hi >> 16) & 0xffu;
  const float oldW = (float)wc;
  unsigned nc[3];
  float newW = oldW + 1.0f;
  const float inv_new = inv_tab[wc + 1];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float oldC = div_exact((float)oc[k], 255.0f, p.inv_255);
    const float c = div_exact(m[k], 255.0f, p.inv_255);
    const float v = div_exact(oldC * oldW + c * 1.0f, newW, inv_new);
    nc[k] = (unsigned)(unsigned char)(v * 255.0f);
  }
  newW = fminf(newW, (float)p.max_w);
  const unsigned new_wc = (unsigned)(unsigned char)newW;
  lo = (lo & 0x00ffffffu) | (nc[0] << 24);
  hi = (hi & 0xff000000u) | nc[1] | (nc[2] << 8) | (new_wc << 16);
  return true;
}

// The colour path is what a wavefront pays most for: only voxels in the narrow band |eta / mu| <= 0.25 take it (about
// one in eight of the updated ones), but a wave64 executes it whenever ANY of its lanes does -- eight sparse executions of
// ~70 instructions per block.  So the one-camera fusion kernel queues the band voxels of a half block in LDS (ColQueue:
// projection + old colour), runs the colour update densely over the queue -- one voxel per lane, usually one pass --
// and hands every result back through LDS.  Same arithmetic per voxel, same bits.
// colour word: clr0 | clr1 << 8 | clr2 << 16 | w_color << 24
template <bool BUF>
__device__ __forceinline__ unsigned fuse_colour_word(unsigned pack, float u, float w, const IntegrateParams &p,
                                                     const float *inv_tab) {
  unsigned lo = pack << 24, hi = pack >> 8;
  const Vec4 unused = {0.0f, 0.0f, 0.0f, 1.0f};
  fuse_colour<true, BUF>(lo, hi, u, w, unused, p, inv_tab);
  return (lo >> 24) | ((hi & 0xffffffu) << 8);
}

// computeUpdatedVoxelColorInfo of the de-integration (update_voxel<true, true>'s colour part) on a queued colour word
template <bool BUF>
__device__ __forceinline__ unsigned defuse_colour_word(unsigned pack, float u, float w, const IntegrateParams &p,
                                                       const float *inv_tab) {
  const unsigned wc = pack >> 24;
  if (wc < 1) return pack;  // nothing was ever fused into this colour
  float m[3];
  if constexpr (BUF) bilinear_rgb_buf(image_rsrc(p.rgba, p.Wr, p.Hr), u, w, p.Wr, m);
  else bilinear_rgb(p.rgba, u, w, p.Wr, m);
  const float oldW = (float)wc, remW = oldW - 1.0f;
  if (remW == 0.0f) return 0u;  // colour 0, weight 0
  const float inv_rem = inv_tab[wc - 1];
  unsigned out = (unsigned)(unsigned char)remW << 24;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float oldC = div_exact((float)((pack >> (8 * k)) & 0xffu), 255.0f, p.inv_255);
    const float c = div_exact(m[k], 255.0f, p.inv_255);
    float v = div_exact(oldC * oldW - c * 1.0f, remW, inv_rem);
    v = fmaxf(0.0f, fminf(1.0f, v));
    out |= (unsigned)(unsigned char)(v * 255.0f) << (8 * k);
  }
  return out;
}

constexpr int kColQueue = 256;  // one slot per voxel of a half block

// ComputeUpdatedVoxelInfo<hasColor>::compute for the two voxels of one chunk: (vv.x, vv.y) at x, (vv.z, vv.w) at x + 1
// QUEUE: the colour update is not done here; `cmask` (bit h: voxel h is inside the narrow band) and the projections come
// back so that the caller can run the colour updates of a half block densely (see k_integrate)
template <bool SAME_CAM, bool QUEUE = false>
__device__ __forceinline__ bool fuse_pair(uint4 &vv, f2 pcx, f2 pcy, f2 pcz, const Vec4 &pm0, const Vec4 &pm1,
                                          const IntegrateParams &p, const float *inv_tab, unsigned *cmask = nullptr,
                                          f2 *u_out = nullptr, f2 *w_out = nullptr) {
  if constexpr (QUEUE) *cmask = 0;
  bool act0 = pcz.x >= kMinCamZ, act1 = pcz.y >= kMinCamZ;
  if (p.stop_max) {
    act0 = act0 && (int)((vv.x >> 16) & 0xffu) != p.max_w;
    act1 = act1 && (int)((vv.z >> 16) & 0xffu) != p.max_w;
  }
  if (!(act0 || act1)) return false;
  const f2 fx2 = {p.fx_d, p.fx_d}, fy2 = {p.fy_d, p.fy_d}, cx2 = {p.cx_d, p.cx_d}, cy2 = {p.cy_d, p.cy_d};
  const f2 u = div_ieee2(fx2 * pcx, pcz) + cx2;  // projParams.x * pt.x / pt.z + projParams.z
  const f2 w = div_ieee2(fy2 * pcy, pcz) + cy2;
  const float wmax = (float)(p.Wd - 2), hmax = (float)(p.Hd - 2);
  act0 = act0 && in_image(u.x, w.x, wmax, hmax);
  act1 = act1 && in_image(u.y, w.y, wmax, hmax);
  if (!(act0 || act1)) return false;
  const f2 half = {0.5f, 0.5f};
  const f2 ur = u + half, wr = w + half;
  const int i0 = act0 ? (int)ur.x + (int)wr.x * p.Wd : 0, i1 = act1 ? (int)ur.y + (int)wr.y * p.Wd : 0;
  f2 dm;
  dm.x = p.depth[i0];
  dm.y = p.depth[i1];
  const f2 eta = dm - pcz;
  act0 = act0 && !(dm.x <= 0.0f) && !(eta.x < -p.mu);
  act1 = act1 && !(dm.y <= 0.0f) && !(eta.y < -p.mu);
  if (!(act0 || act1)) return false;
  const f2 mu2 = {p.mu, p.mu};
  const f2 eta_mu = div_ieee2(eta, mu2);  // true IEEE division (the 3-instruction form is not exact for arbitrary mu)
  f2 sd;
  sd.x = (float)(short)(vv.x & 0xffffu);
  sd.y = (float)(short)(vv.z & 0xffffu);
  const f2 inv32767 = {p.inv_32767, p.inv_32767};
  const f2 oldF = div_exact2(sd, 32767.0f, inv32767);
  const int oldW0 = (int)((vv.x >> 16) & 0xffu), oldW1 = (int)((vv.z >> 16) & 0xffu);
  f2 newF;
  newF.x = fminf(1.0f, eta_mu.x);
  newF.y = fminf(1.0f, eta_mu.y);
  const int addW0 = new_weight(p, dm.x), addW1 = new_weight(p, dm.y);
  f2 oW, aW;
  oW.x = (float)oldW0; oW.y = (float)oldW1;
  aW.x = (float)addW0; aW.y = (float)addW1;
  f2 nf = oW * oldF + aW * newF;  // oldW * oldF + newW * newF: two products, one sum (no contraction)
  int nW0 = oldW0 + addW0, nW1 = oldW1 + addW1;
  f2 nWf, inv;
  nWf.x = (float)nW0; nWf.y = (float)nW1;
  inv.x = inv_tab[nW0]; inv.y = inv_tab[nW1];
  {  // div_exact with per-component divisors
    const f2 q = nf * inv;
    const f2 r = __builtin_elementwise_fma(-nWf, q, nf);
    nf = __builtin_elementwise_fma(r, inv, q);
  }
  nW0 = nW0 < p.max_w ? nW0 : p.max_w;
  nW1 = nW1 < p.max_w ? nW1 : p.max_w;
  const f2 scale = {32767.0f, 32767.0f};
  const f2 sf = nf * scale;  // floatToValue: (short)(x * 32767)
  if (act0) vv.x = (vv.x & 0xff000000u) | ((unsigned)nW0 << 16) | (unsigned)(unsigned short)(short)sf.x;
  if (act1) vv.z = (vv.z & 0xff000000u) | ((unsigned)nW1 << 16) | (unsigned)(unsigned short)(short)sf.y;
  // colour: only inside the narrow band around the surface
  const bool col0 = act0 && !((eta.x > p.mu) || (fabsf(eta_mu.x) > 0.25f));
  const bool col1 = act1 && !((eta.y > p.mu) || (fabsf(eta_mu.y) > 0.25f));
  if constexpr (QUEUE) {
    *cmask = (col0 ? 1u : 0u) | (col1 ? 2u : 0u);
    *u_out = u;
    *w_out = w;
  } else {
    if (col0) fuse_colour<SAME_CAM>(vv.x, vv.y, u.x, w.x, pm0, p, inv_tab);
    if (col1) fuse_colour<SAME_CAM>(vv.z, vv.w, u.y, w.y, pm1, p, inv_tab);
  }
  return true;
}

// ---- fuse_pair in two steps (plain one-camera fusion: weight 1, no stopIntegratingAtMaxW) --------------------------------
// A wave's serial chain per half block used to be  project chunk 0 -> read depth -> wait -> update -> project chunk 1 ->
// read depth -> wait -> update.  Split, both chunks are projected and BOTH pairs of depth pixels requested before
// either update waits for its pair: one exposed round trip per half block instead of two (20.7 -> 19.9 us per launch on
// the bench scene; the waves are parked on memory for half of their lifetime, SQ_WAIT_ANY, so the chain is what counts).
struct PairProj {
  f2 u, w, pcz, dm;
  bool act0, act1;
};

// Projection of the two voxels and the request for their depth pixels.  Branch-free: a voxel that fails a test keeps
// computing (NaN included), is masked out by act0 / act1, and its read cannot leave the buffer.
__device__ __forceinline__ void pair_project(PairProj &q, f2 pcx, f2 pcy, f2 pcz, const IntegrateParams &p,
                                             __amdgpu_buffer_rsrc_t depth_rs) {
  const f2 fx2 = {p.fx_d, p.fx_d}, fy2 = {p.fy_d, p.fy_d}, cx2 = {p.cx_d, p.cx_d}, cy2 = {p.cy_d, p.cy_d};
  q.pcz = pcz;
  q.u = div_ieee2(fx2 * pcx, pcz) + cx2;
  q.w = div_ieee2(fy2 * pcy, pcz) + cy2;
  const float wmax = (float)(p.Wd - 2), hmax = (float)(p.Hd - 2);
  q.act0 = (pcz.x >= kMinCamZ) & in_image(q.u.x, q.w.x, wmax, hmax);
  q.act1 = (pcz.y >= kMinCamZ) & in_image(q.u.y, q.w.y, wmax, hmax);
  const f2 half = {0.5f, 0.5f};
  const f2 ur = q.u + half, wr = q.w + half;
  q.dm.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(depth_rs, pixel_offset((int)ur.x, (int)wr.x, p.Wd), 0, 0));
  q.dm.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(depth_rs, pixel_offset((int)ur.y, (int)wr.y, p.Wd), 0, 0));
}

// The depth update of the two voxels (fuse_pair from `eta` on; PLAIN: newW = 1); `cmask` as fuse_pair<., QUEUE> returns
// it.  DEINT: update_voxel<true, .>'s depth part on both voxels at once -- W' = W - w, F' = clamp((W F - w f) / W'),
// W' = 0 -> the empty voxel, a voxel with W < w is left alone; the narrow-band colour test does not depend on that.
template <bool DEINT, bool PLAIN>
__device__ __forceinline__ bool pair_update(uint4 &vv, const PairProj &q, const IntegrateParams &p, const float *inv_tab,
                                            unsigned &cmask) {
  cmask = 0;
  const f2 eta = q.dm - q.pcz;
  const bool act0 = q.act0 && !(q.dm.x <= 0.0f) && !(eta.x < -p.mu);
  const bool act1 = q.act1 && !(q.dm.y <= 0.0f) && !(eta.y < -p.mu);
  if (!(act0 || act1)) return false;
  const f2 mu2 = {p.mu, p.mu};
  const f2 eta_mu = div_ieee2(eta, mu2);
  f2 sd;
  sd.x = (float)(short)(vv.x & 0xffffu);
  sd.y = (float)(short)(vv.z & 0xffffu);
  const f2 inv32767 = {p.inv_32767, p.inv_32767};
  const f2 oldF = div_exact2(sd, 32767.0f, inv32767);
  const int oldW0 = (int)((vv.x >> 16) & 0xffu), oldW1 = (int)((vv.z >> 16) & 0xffu);
  f2 newF;
  newF.x = fminf(1.0f, eta_mu.x);
  newF.y = fminf(1.0f, eta_mu.y);
  f2 oW;
  oW.x = (float)oldW0; oW.y = (float)oldW1;
  const int addW0 = new_weight<PLAIN>(p, q.dm.x), addW1 = new_weight<PLAIN>(p, q.dm.y);
  f2 aW;
  aW.x = (float)addW0; aW.y = (float)addW1;
  const f2 scale = {32767.0f, 32767.0f};
  if constexpr (!DEINT) {
    // oldW * oldF + newW * newF (PLAIN: newW = 1 and that product is exact: it is newF)
    f2 nf = PLAIN ? oW * oldF + newF : oW * oldF + aW * newF;
    int nW0 = oldW0 + addW0, nW1 = oldW1 + addW1;
    f2 nWf, inv;
    nWf.x = (float)nW0; nWf.y = (float)nW1;
    inv.x = inv_tab[nW0]; inv.y = inv_tab[nW1];
    {  // div_exact with per-component divisors
      const f2 qq = nf * inv;
      const f2 r = __builtin_elementwise_fma(-nWf, qq, nf);
      nf = __builtin_elementwise_fma(r, inv, qq);
    }
    nW0 = nW0 < p.max_w ? nW0 : p.max_w;
    nW1 = nW1 < p.max_w ? nW1 : p.max_w;
    const f2 sf = nf * scale;
    if (act0) vv.x = (vv.x & 0xff000000u) | ((unsigned)nW0 << 16) | (unsigned)(unsigned short)(short)sf.x;
    if (act1) vv.z = (vv.z & 0xff000000u) | ((unsigned)nW1 << 16) | (unsigned)(unsigned short)(short)sf.y;
  } else {
    const int rW0 = oldW0 - addW0, rW1 = oldW1 - addW1;  // (< 0: the voxel holds less than this frame's weight -- left alone)
    f2 nf = oW * oldF - aW * newF;  // oldW * oldF - newW * newF: two products, one difference
    f2 rWf, inv;
    rWf.x = (float)rW0; rWf.y = (float)rW1;
    inv.x = inv_tab[rW0 > 0 ? rW0 : 0]; inv.y = inv_tab[rW1 > 0 ? rW1 : 0];
    {
      const f2 qq = nf * inv;
      const f2 r = __builtin_elementwise_fma(-rWf, qq, nf);
      nf = __builtin_elementwise_fma(r, inv, qq);
    }
    nf.x = fmaxf(-1.0f, fminf(1.0f, nf.x));
    nf.y = fmaxf(-1.0f, fminf(1.0f, nf.y));
    const f2 sf = nf * scale;
    // W' = 0: sdf 32767, weight 0 (the quotient above is then 0 / 0 and not used)
    const unsigned s0 = rW0 == 0 ? 0x7fffu : (unsigned)(unsigned short)(short)sf.x;
    const unsigned s1 = rW1 == 0 ? 0x7fffu : (unsigned)(unsigned short)(short)sf.y;
    if (act0 && rW0 >= 0) vv.x = (vv.x & 0xff000000u) | ((unsigned)rW0 << 16) | s0;
    if (act1 && rW1 >= 0) vv.z = (vv.z & 0xff000000u) | ((unsigned)rW1 << 16) | s1;
  }
  // upstream skips the colour when `eta > mu || fabs(eta / mu) > 0.25`.  With mu > 0, eta > mu makes the exact quotient
  // exceed 1, so its rounding is >= 1 > 0.25: the first test never decides anything the second does not already
  const bool col0 = act0 && !(fabsf(eta_mu.x) > 0.25f);
  const bool col1 = act1 && !(fabsf(eta_mu.y) > 0.25f);
  cmask = (col0 ? 1u : 0u) | (col1 ? 2u : 0u);
  return true;
}

constexpr int kMaxGroup = 8;
// 8192 waves = one full residency wave of the 256 CUs (8 per SIMD), as 512 workgroups of 16 waves: two per CU.  (2048
// workgroups of 4 waves: 0.35 us slower per launch -- four times the workgroups to dispatch and tables to fill.  A
// workgroup barrier between the entry gathers and the block loads, so that no wave's small dependent loads queue behind
// its neighbours' kilobytes, was measured too, with 4 and with 16 waves: +0.6 ... 0.9 us.)
constexpr int kWgWaves = 16;
constexpr int kIntegrateGrid = 8192 / kWgWaves;

// SAME_CAM: the RGB camera is the depth camera (identity calib, as the reference sets it up): the colour update
// reuses the depth projection, and the kernel carries one matrix instead of two (the scalar register file does not
// hold both without spilling to VGPR lanes).
// PLAIN: none of the optional features is in use -- stored list, shards, dirty marks, stopIntegratingAtMaxW, depth
// weighting -- i.e. the per-frame fusion of the reference's configuration.  Compiled without them the kernel carries
// fewer live arguments (36 scalar registers spilled to VGPR lanes instead of 72, a shorter preamble) and can afford the
// split update (pair_project / pair_update): 22.9 -> 21.4 us for the specialisation, -> 19.9 us with the split.
// Queueing the frame's visible list on the ring (ProcessFrame) = one bit per visible resident block in that block's ring word,
// and its last_seen stamp.  The words lie 64 bytes apart (one line per block), in slot order -- random with respect to the
// order of the list.  Measured on the S-stress map (V = 262 k, 2.1 GB through the launch; profiles/experiments/
// push_variants.sh, profiles/r04_push_variants.json): no push 451 us, the last_seen store alone 456, the ring bit as a
// device-scope atomic from the gathering lane 550 (the product up to round 3: 0.49 of the HBM peak where the kernel without
// push reaches 0.60), as a plain load-OR-store 522, either of them issued behind the block's stores: the same; with the
// slots in list order the push costs nothing at all.  What costs is a quarter of a million 64-byte read-modify-writes at
// random places of a 16 MB array in the middle of a 4.6 TB/s stream -- and, in a block wave's prologue, the wait for them.
// So above kPushJobMin visible blocks (the voxels no longer fit the Infinity Cache) the block waves do not push: kPushWgs
// extra workgroups at the END of the grid (they start as block workgroups retire) walk the visible list once more, four
// entries per lane in flight -- entry (a cache hit: a block wave has just read it), ring word, OR, store -- where a lane
// that waits holds nothing else up: 506 us = 0.53.  Below it -- every per-frame launch of a real sequence -- the gathering
// lane's atomic stays: on the bench scene (V = 7.9 k) the trailing workgroups make the launch 0.45 us LONGER (20.0 -> 20.45 us,
// three alternations), because there the launch ends with its last block wave and they queue behind it.  Same bits, same
// stamps either way.
// kPushJobMin = 65536 is dslam_engine::push_job_min (dslam_internal.h): a launch parameter, so that the parity test can lower it
constexpr int kPushWgs = 64;
constexpr int kPushPer = 4;   // entries per lane whose loads travel together (id -> entry -> ring word: three round trips per batch)
__device__ __forceinline__ void push_visible_list_job(const IntegrateParams &p, int wg) {
  if (!p.push_words) return;
  const int nvis = p.rc->no_visible;
  if (nvis < p.push_job_min) return;   // (the block waves push)
  const unsigned long long bit = 1ull << (p.push_bit & 63);
  constexpr int kStride = kPushWgs * kWgWaves * 64;
  for (int i0 = wg * (kWgWaves * 64) + (int)threadIdx.x; i0 < nvis; i0 += kStride * kPushPer) {
    int id[kPushPer], ptr[kPushPer];
    unsigned long long old[kPushPer];
#pragma unroll
    for (int q = 0; q < kPushPer; q++) id[q] = i0 + q * kStride < nvis ? p.visible_ids[i0 + q * kStride] : -1;
#pragma unroll
    for (int q = 0; q < kPushPer; q++) ptr[q] = id[q] >= 0 ? load_entry(p.hash, id[q]).ptr : -1;
#pragma unroll
    for (int q = 0; q < kPushPer; q++)
      old[q] = ptr[q] >= 0 ? p.masks[((size_t)ptr[q] * 2 + p.push_ring) * p.push_words + (p.push_bit >> 6)] : 0ull;
#pragma unroll
    for (int q = 0; q < kPushPer; q++)
      if (ptr[q] >= 0) {   // (one lane per block and launch: no race)
        p.masks[((size_t)ptr[q] * 2 + p.push_ring) * p.push_words + (p.push_bit >> 6)] = old[q] | bit;
        p.last_seen[ptr[q]] = p.push_frame;
      }
  }
}

template <bool DEINT, bool SAME_CAM, bool PLAIN = false, bool DIAG = false, bool STREAM = false>
__global__ __launch_bounds__(kWgWaves * 64, 8) void k_integrate(IntegrateParams p) {
  constexpr bool stream = STREAM;   // (cache policy of the voxel chunks: vox_load2)
  static_assert(!PLAIN || (!DEINT && SAME_CAM && DSLAM_PACKED && DSLAM_COLOUR_QUEUE), "PLAIN is the queued one-camera fusion");
  static_assert(!DIAG || PLAIN, "the per-wave timeline exists for the plain fusion kernel");
  if constexpr (PLAIN) {
    if ((int)blockIdx.x >= kIntegrateGrid) { push_visible_list_job(p, (int)blockIdx.x - kIntegrateGrid); return; }
  }
  __shared__ float inv_tab[kInvTab];
  [[maybe_unused]] unsigned long long diag_entry = 0, diag_cyc = 0;
  [[maybe_unused]] bool diag_first = true;  // a wave records its first block
  if constexpr (DIAG) { diag_entry = wall_clock64(); diag_cyc = clock64(); }
#define DSLAM_STAMP(k) do { if constexpr (DIAG) { if (lane == 0 && diag_first) p.dbg_waves[(size_t)wave * 16 + (k)] = wall_clock64(); } } while (0)
  // the one-camera fusion variant runs its colour updates densely from a per-wave LDS queue (fuse_colour_word)
  constexpr bool kQueueColour = SAME_CAM && DSLAM_PACKED && DSLAM_COLOUR_QUEUE;
  // the split update (pair_project / pair_update): plain fusion, and the one-camera de-integration (41.6 -> 34.2 us per
  // launch against the voxel-by-voxel form it replaces; the general fusion variant has no registers for it)
  constexpr bool kPairPath = PLAIN || (DEINT && kQueueColour);
  // The queue of one wave.  The data of a queued voxel sits at the voxel's OWN place (chunk-voxel k of lane l: k * 64 + l:
  // no address arithmetic for its owner, neither to queue it nor to fetch the result), `list` holds the places in queue
  // order, and the result comes back at the same place.  4.25 KiB per wave: 70 KiB per 16-wave workgroup with the table (two workgroups per CU).
  struct ColQueue {
    float u[kColQueue], w[kColQueue];
    unsigned c[kColQueue], r[kColQueue];
    unsigned char list[kColQueue];
  };
  __shared__ ColQueue col_q[kQueueColour ? kWgWaves : 1];
  for (int i = threadIdx.x; i < kInvTab; i += kWgWaves * 64) inv_tab[i] = recip_table_entry(i);  // = RN(1 / i)
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * (kWgWaves * 64) + threadIdx.x) >> 6));
  constexpr int n_waves = kIntegrateGrid * kWgWaves;  // (the launch below uses exactly this grid)
  if constexpr (DIAG) {
    if (lane == 0) { p.dbg_waves[(size_t)wave * 16] = diag_entry; p.dbg_waves[(size_t)wave * 16 + 12] = diag_cyc; p.dbg_waves[(size_t)wave * 16 + 1] = wall_clock64(); }
  }
  // With at most one block per wave (G = 1: the bench's 7.9 k blocks on 8192 waves) wave w takes list entry w, so that
  // id is requested BEFORE the list length is known and the two loads travel together.  (`zero` is opaque to the
  // compiler: with a provably uniform address it would make this a scalar load and wait for it on the spot.)
  // (PLAIN only: the other variants have no register to carry it in)
  [[maybe_unused]] int id_spec = 0;
  if (PLAIN && p.spec_ids && lane == 0) {
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    id_spec = p.visible_ids[wave + zero];
  }
  const int nvis = p.rc->no_visible;
  if (p.timer_slot && blockIdx.x == 0 && threadIdx.x == 0) *p.timer_slot = nvis;
  DSLAM_STAMP(2);
  // entries per wave and round: the largest group that still gives EVERY wave of the grid a group (floor, not ceil: with
  // 11 k visible blocks on 8192 waves, one full round of single blocks plus a partial second one beats 5.5 k waves of two
  // blocks each -- 1.5-2.7 % on the re-integration batch, whose launches see that many)
  int G = nvis / n_waves;
  G = G < 1 ? 1 : (G > kMaxGroup ? kMaxGroup : G);

  // lane-constant voxel coordinates inside a block
  const int vx0 = (lane & 3) * 2, vy = (lane >> 2) & 7, vz0 = lane >> 5;

  for (int base = wave * G; base < nvis; base += n_waves * G) {
    // lanes 0..G-1 gather the group's hash entries (one 16-byte load each) and do everything that concerns a block as a
    // whole -- ring push, dirty mark, shard test -- so that none of those kernel arguments is used inside the voxel
    // loop below: the kernel is short of scalar registers (80 at 8 waves per SIMD, the rest spills to VGPR lanes), and
    // this way the spill code sits in the per-block prologue instead of in every 16-byte chunk (23.0 -> 22.1 us).
    // (Reading these arguments with vector loads from the argument segment, so that they need no scalar registers at
    // all, leaves 28 spills instead of 72 and is SLOWER, 22.8 us: the loads sit on every wave's critical path.)
    int e_ptr = -2, e_px = 0, e_py = 0, e_pz = 0;
    if (lane < G && base + lane < nvis) {
      const HashEntry e = load_entry(p.hash, (PLAIN && p.spec_ids && G == 1 && base == wave) ? id_spec : p.visible_ids[base + lane]);
      e_ptr = e.ptr; e_px = e.pos[0]; e_py = e.pos[1]; e_pz = e.pos[2];
      if (!PLAIN && p.expect_pos) {  // a list stored with a keyframe: the entry must still hold the block it held then
        const short4 ep = p.expect_pos[base + lane];
        if (ep.x != e.pos[0] || ep.y != e.pos[1] || ep.z != e.pos[2]) e_ptr = -2;
      }
      if (e_ptr >= 0) {
        const int ptr = e_ptr;
        if (p.push_words && (!PLAIN || nvis < p.push_job_min)) {  // queue this block on the visible-list ring (one lane per block: no race)
          unsigned long long *word = &p.masks[((size_t)ptr * 2 + p.push_ring) * p.push_words + (p.push_bit >> 6)];
          // (PLAIN: an atomic OR whose result nobody reads -- nothing to wait for in front of the block loads)
          int bit = p.push_bit & 63;
          // (STREAM: the shift stays here -- hoisted out of the loops, the 64-bit mask is the one value the streaming variants
          // have no register for, and they would be the only kernels of the library with scratch memory)
          if constexpr (STREAM) asm volatile("" : "+v"(bit));
          if constexpr (PLAIN) __hip_atomic_fetch_or(word, 1ull << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *word |= 1ull << bit;
          p.last_seen[ptr] = p.push_frame;
        }
        if constexpr (!PLAIN) {
          // (before the shard test: every rank of a sharded batch ends up with the same set of marks)
          if (p.dirty) p.dirty[ptr] = 1;
          if (p.num_shards > 1) {
            int cb = p.chunk_blocks, ns = p.num_shards;
            if constexpr (STREAM) asm volatile("" : "+s"(cb), "+s"(ns));   // (the divisions' reciprocals stay here: see the ring push above)
            if (((ptr / cb) % ns) != p.shard) e_ptr = -3;
          }
          if (p.shard_count >= 0 && (ptr < p.shard_first || ptr >= p.shard_first + p.shard_count)) e_ptr = -3;
        }
      }
    }
    DSLAM_STAMP(3);
    for (int k = 0; k < G; k++) {
      const int ptr = __builtin_amdgcn_readlane(e_ptr, k);
      if (ptr < 0) continue;
      const int gx = __builtin_amdgcn_readlane(e_px, k) * kBlock;
      const int gy = __builtin_amdgcn_readlane(e_py, k) * kBlock;
      const int gz = __builtin_amdgcn_readlane(e_pz, k) * kBlock;
      uint4 *blk = p.voxels16 + (size_t)ptr * (kBlock3 / 2);

      // pc = M_d * (x, y, z, 1) = ((m0*x + m4*y) + m8*z) + m12 per component.  The products depend only on the
      // lane's 2 x values, its y, and the 4 z values of its loads, so they are formed once per block; what is
      // left per voxel are the additions, in the reference's order (bit-identical to the full mat-vec).
      const float fy = (float)(gy + vy) * p.voxel_size;
      float fxv[2], pxy[2][3];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        fxv[h] = (float)(gx + vx0 + h) * p.voxel_size;
        pxy[h][0] = p.M_d.m[0] * fxv[h] + p.M_d.m[4] * fy;
        pxy[h][1] = p.M_d.m[1] * fxv[h] + p.M_d.m[5] * fy;
        pxy[h][2] = p.M_d.m[2] * fxv[h] + p.M_d.m[6] * fy;
      }
      // the block is processed as two halves of 2 KiB (2 x dwordx4 per lane in flight each): 16 fewer live VGPRs
      // than holding all four chunks, which is what lets the kernel run 8 waves per SIMD without spilling
#pragma unroll 1
      for (int half = 0; half < 2; half++) {
      uint4 v[2];
      if constexpr (stream) {
        v[0] = load_nt(blk + (half * 2) * 64 + lane);
        v[1] = load_nt(blk + (half * 2 + 1) * 64 + lane);
      } else {
        v[0] = blk[(half * 2) * 64 + lane];
        v[1] = blk[(half * 2 + 1) * 64 + lane];
      }
      bool chs[2] = {false, false};
      [[maybe_unused]] int q_n = 0;                    // queued colour updates of this half (wave-uniform)
      [[maybe_unused]] unsigned cms[2] = {0u, 0u};     // per chunk: which of the lane's two voxels are queued
      // queue the narrow-band voxels of chunk jj (cm: bit h = voxel h) with their projections
      [[maybe_unused]] auto queue_colour = [&](int jj, unsigned cm, f2 uo, f2 wo) {
        cms[jj] = cm;
        if (!__ballot(cm != 0u)) return;  // (most chunks of a block far from the surface queue nothing)
        ColQueue &Q = col_q[threadIdx.x >> 6];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const bool c = (cm >> h) & 1u;
          const unsigned long long bm = __ballot(c);
          if (c) {
            const unsigned lo = h ? v[jj].z : v[jj].x, hi = h ? v[jj].w : v[jj].y;
            const int own = (jj * 2 + h) * 64 + lane;
            const int slot = q_n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
            Q.u[own] = h ? uo.y : uo.x;
            Q.w[own] = h ? wo.y : wo.x;
            Q.c[own] = __builtin_amdgcn_perm(hi, lo, 0x06050403u);  // clr0 | clr1 << 8 | clr2 << 16 | w_color << 24
            Q.list[slot] = (unsigned char)own;
          }
          q_n += __popcll(bm);
        }
      };
      if constexpr (kPairPath) {
        // PLAIN: both chunks projected and their depth pixels requested, then both updated (see pair_project).  The
        // de-integration variant carries the optional features and has no registers to hold two projections: its chunks
        // go one after the other -- packed arithmetic and the colour queue, but one depth wait per chunk.
        PairProj pq[PLAIN ? 2 : 1];
        const __amdgpu_buffer_rsrc_t depth_rs = image_rsrc(p.depth, p.Wd, p.Hd);
        auto project = [&](int jj, PairProj &q) {
          const float fz = (float)(gz + (half * 2 + jj) * 2 + vz0) * p.voxel_size;
          const float az0 = p.M_d.m[8] * fz, az1 = p.M_d.m[9] * fz, az2 = p.M_d.m[10] * fz;
          const f2 a0 = {az0, az0}, a1 = {az1, az1}, a2 = {az2, az2};
          const f2 t0 = {p.M_d.m[12], p.M_d.m[12]}, t1 = {p.M_d.m[13], p.M_d.m[13]}, t2 = {p.M_d.m[14], p.M_d.m[14]};
          const f2 px = {pxy[0][0], pxy[1][0]}, py = {pxy[0][1], pxy[1][1]}, pz = {pxy[0][2], pxy[1][2]};
          pair_project(q, (px + a0) + t0, (py + a1) + t1, (pz + a2) + t2, p, depth_rs);
        };
        if constexpr (PLAIN) {
#pragma unroll
          for (int jj = 0; jj < 2; jj++) project(jj, pq[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
          if constexpr (!PLAIN) project(jj, pq[0]);
          PairProj &q = pq[PLAIN ? jj : 0];
          unsigned cm;
          chs[jj] = pair_update<DEINT, PLAIN>(v[jj], q, p, inv_tab, cm);
          queue_colour(jj, cm, q.u, q.w);
          DSLAM_STAMP(4 + half * 4 + jj);
        }
      } else
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
        const int j = half * 2 + jj;
        const float fz = (float)(gz + j * 2 + vz0) * p.voxel_size;
        const float az0 = p.M_d.m[8] * fz, az1 = p.M_d.m[9] * fz, az2 = p.M_d.m[10] * fz;
        bool ch = false;
        if constexpr (!DEINT && DSLAM_PACKED) {
          // both voxels of the chunk at once (packed FP32): pc = ((pxy + az) + m12..14), as in the scalar form
          const f2 a0 = {az0, az0}, a1 = {az1, az1}, a2 = {az2, az2};
          const f2 t0 = {p.M_d.m[12], p.M_d.m[12]}, t1 = {p.M_d.m[13], p.M_d.m[13]}, t2 = {p.M_d.m[14], p.M_d.m[14]};
          const f2 px = {pxy[0][0], pxy[1][0]}, py = {pxy[0][1], pxy[1][1]}, pz = {pxy[0][2], pxy[1][2]};
          const Vec4 pm0 = {fxv[0], fy, fz, 1.0f}, pm1 = {fxv[1], fy, fz, 1.0f};
          if constexpr (kQueueColour) {
            unsigned cm;
            f2 uo, wo;
            ch = fuse_pair<SAME_CAM, true>(v[jj], (px + a0) + t0, (py + a1) + t1, (pz + a2) + t2, pm0, pm1, p, inv_tab, &cm, &uo, &wo);
            queue_colour(jj, cm, uo, wo);
          } else {
            ch = fuse_pair<SAME_CAM>(v[jj], (px + a0) + t0, (py + a1) + t1, (pz + a2) + t2, pm0, pm1, p, inv_tab);
          }
        } else {
#pragma unroll
          for (int h = 0; h < 2; h++) {
            unsigned &lo = h ? v[jj].z : v[jj].x;
            unsigned &hi = h ? v[jj].w : v[jj].y;
            Vec4 pc, pm;
            pc.x = (pxy[h][0] + az0) + p.M_d.m[12];
            pc.y = (pxy[h][1] + az1) + p.M_d.m[13];
            pc.z = (pxy[h][2] + az2) + p.M_d.m[14];
            pc.w = 1.0f;
            pm.x = fxv[h]; pm.y = fy; pm.z = fz; pm.w = 1.0f;
            ch |= update_voxel<DEINT, SAME_CAM>(lo, hi, pc, pm, p, inv_tab);
          }
        }
        chs[jj] = ch;
      }
      if constexpr (kQueueColour) {
        if (q_n > 0) {
          // the queued colour updates, one per lane; every result goes back to the place of its voxel
          ColQueue &Q = col_q[threadIdx.x >> 6];
          // (the queue passes data between the lanes of ONE wave through LDS: program order + lockstep make that work; the
          // wave barriers say so to the compiler, which may otherwise move the LDS accesses of different lanes' data)
          __builtin_amdgcn_wave_barrier();
          for (int i = lane; i < q_n; i += 64) {
            const int o = Q.list[i];
            // (texels through a buffer resource where the variant has scalar registers to spare for a second one)
            if constexpr (DEINT) Q.r[o] = defuse_colour_word<false>(Q.c[o], Q.u[o], Q.w[o], p, inv_tab);
            else Q.r[o] = fuse_colour_word<PLAIN>(Q.c[o], Q.u[o], Q.w[o], p, inv_tab);
          }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int jj = 0; jj < 2; jj++)
#pragma unroll
            for (int h = 0; h < 2; h++)
              if ((cms[jj] >> h) & 1u) {
                const unsigned word = Q.r[(jj * 2 + h) * 64 + lane];
                unsigned &lo = h ? v[jj].z : v[jj].x;
                unsigned &hi = h ? v[jj].w : v[jj].y;
                lo = __builtin_amdgcn_perm(word, lo, 0x04020100u);  // (lo & 0x00ffffff) | word << 24
                hi = __builtin_amdgcn_perm(hi, word, 0x07030201u);  // (hi & 0xff000000) | word >> 8
              }
        }
      }
      DSLAM_STAMP(6 + half * 4);
#pragma unroll
      for (int jj = 0; jj < 2; jj++)
        if (chs[jj]) {
          if constexpr (stream) store_nt(blk + (half * 2 + jj) * 64 + lane, v[jj]);
          else blk[(half * 2 + jj) * 64 + lane] = v[jj];
        }
      DSLAM_STAMP(7 + half * 4);
      }
      if constexpr (DIAG) {
        if (lane == 0 && diag_first) {
          unsigned hw, xcc;
          asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
          asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
          p.dbg_waves[(size_t)wave * 16 + 13] = clock64();
          p.dbg_waves[(size_t)wave * 16 + 14] = (unsigned long long)ptr;
          p.dbg_waves[(size_t)wave * 16 + 15] = ((unsigned long long)xcc << 32) | hw;
        }
        diag_first = false;
      }
    }
  }
#undef DSLAM_STAMP
}

static void fill_params(IntegrateParams &ip, dslam_engine *e, dslam_scene *s, const dslam_view *v,
                        const dslam_render_state *r, const float *M_d, const float *intr_d, const float *M_rgb,
                        const float *intr_rgb) {
  ip.visible_ids = r->visible_ids; ip.rc = r->counters; ip.hash = s->hash;
  ip.voxels16 = reinterpret_cast<uint4 *>(s->voxels);
  ip.depth = v->depth; ip.rgba = v->rgba_src;
  ip.Wd = v->w_d; ip.Hd = v->h_d; ip.Wr = v->w_rgb; ip.Hr = v->h_rgb;
  memcpy(ip.M_d.m, M_d, 64);
  memcpy(ip.M_rgb.m, M_rgb ? M_rgb : M_d, 64);
  const float *kr = intr_rgb ? intr_rgb : intr_d;
  ip.fx_d = intr_d[0]; ip.fy_d = intr_d[1]; ip.cx_d = intr_d[2]; ip.cy_d = intr_d[3];
  ip.fx_r = kr[0]; ip.fy_r = kr[1]; ip.cx_r = kr[2]; ip.cy_r = kr[3];
  ip.voxel_size = s->p.voxel_size; ip.mu = s->p.mu; ip.max_w = s->p.max_w;
  ip.inv_32767 = 1.0f / 32767.0f; ip.inv_255 = 1.0f / 255.0f;
  ip.same_cam = (memcmp(ip.M_d.m, ip.M_rgb.m, 64) == 0 && kr[0] == intr_d[0] && kr[1] == intr_d[1] && kr[2] == intr_d[2] &&
                 kr[3] == intr_d[3] && v->w_rgb == v->w_d && v->h_rgb == v->h_d) ? 1 : 0;
  ip.stop_max = s->p.stop_integrating_at_max_w;
  ip.depth_weighting = e->wp.depth_weighting; ip.max_new_w = e->wp.max_new_w; ip.max_distance = e->wp.max_distance;
  ip.shard = s->shard; ip.num_shards = s->num_shards; ip.chunk_blocks = s->chunk_blocks;
  ip.shard_first = s->shard_first; ip.shard_count = s->shard_count;
  ip.dirty = s->dirty_tracking ? s->dirty : nullptr;
  ip.expect_pos = nullptr;
  ip.spec_ids = 0;
}

// stream: the launch is expected to be larger than the Infinity Cache (vox_load2)
static int launch_integrate_params(dslam_engine *e, IntegrateParams &ip, bool deintegrate, bool stream) {
  ip.timer_slot = nullptr;
  ip.dbg_waves = nullptr;
  // Timed launches (bench roofline) attach their two events to the dispatch packet itself (hipExtLaunchKernelGGL), so
  // the elapsed time is the kernel's own start-to-end interval -- what rocprofv3 reports -- not the interval between
  // two separately recorded stream events, which also contains ~3 us of packet processing.
  const bool timed = e->timer_enabled && e->ev_used + 2 <= e->ev_pool.size();
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (timed) {
    ip.timer_slot = e->timer_counts_dev + (e->ev_used / 2);
    ev0 = e->ev_pool[e->ev_used];
    ev1 = e->ev_pool[e->ev_used + 1];
    e->ev_used += 2;
  }
  const dim3 grid(kIntegrateGrid), block(kWgWaves * 64);
  const dim3 grid_plain(kIntegrateGrid + kPushWgs);   // (+ the workgroups that queue the visible list on the ring)
  if (deintegrate) {
    // (streaming instantiations exist for the one-camera de-integration and the plain one-camera fusion)
    if (ip.same_cam && stream) e->stream_launches++;
    if (ip.same_cam && stream) hipExtLaunchKernelGGL((k_integrate<true, true, false, false, true>), grid, block, 0, e->stream, ev0, ev1, 0, ip);
    else if (ip.same_cam) hipExtLaunchKernelGGL((k_integrate<true, true>), grid, block, 0, e->stream, ev0, ev1, 0, ip);
    else hipExtLaunchKernelGGL((k_integrate<true, false>), grid, block, 0, e->stream, ev0, ev1, 0, ip);
  } else {
    const bool plain = DSLAM_PACKED && DSLAM_COLOUR_QUEUE && ip.same_cam && !ip.expect_pos && ip.num_shards <= 1 &&
                       ip.shard_count < 0 && !ip.dirty && !ip.stop_max && !ip.depth_weighting;
    // diagnostics: the per-wave timeline of one launch, well into the run (DSLAM_DBG_INTEGRATE=<file>)
    static const char *dbg_file = getenv("DSLAM_DBG_INTEGRATE");
    static int dbg_calls = 0;
    if (dbg_file && plain && ++dbg_calls == 60) {
      constexpr size_t kTraceBytes = (size_t)kIntegrateGrid * kWgWaves * 16 * sizeof(unsigned long long);
      unsigned long long *trace_dev = nullptr;
      DSLAM_HIP(hipMalloc((void **)&trace_dev, kTraceBytes));
      DSLAM_HIP(hipMemsetAsync(trace_dev, 0, kTraceBytes, e->stream));
      ip.dbg_waves = trace_dev;
      hipExtLaunchKernelGGL((k_integrate<false, true, DSLAM_PACKED && DSLAM_COLOUR_QUEUE, DSLAM_PACKED && DSLAM_COLOUR_QUEUE>), grid_plain, block, 0, e->stream, ev0, ev1, 0, ip);
      DSLAM_HIP(hipGetLastError());
      DSLAM_HIP(hipStreamSynchronize(e->stream));
      std::vector<unsigned long long> h(kTraceBytes / sizeof(unsigned long long));
      DSLAM_HIP(hipMemcpy(h.data(), trace_dev, kTraceBytes, hipMemcpyDeviceToHost));
      if (FILE *f = fopen(dbg_file, "wb")) { fwrite(h.data(), 1, kTraceBytes, f); fclose(f); }
      (void)hipFree(trace_dev);
      return DSLAM_OK;
    }
    if (plain && stream) e->stream_launches++;
    if (plain && stream) hipExtLaunchKernelGGL((k_integrate<false, true, DSLAM_PACKED && DSLAM_COLOUR_QUEUE, false, true>), grid_plain, block, 0, e->stream, ev0, ev1, 0, ip);
    else if (plain) hipExtLaunchKernelGGL((k_integrate<false, true, DSLAM_PACKED && DSLAM_COLOUR_QUEUE>), grid_plain, block, 0, e->stream, ev0, ev1, 0, ip);
    else if (ip.same_cam) hipExtLaunchKernelGGL((k_integrate<false, true>), grid, block, 0, e->stream, ev0, ev1, 0, ip);
    else hipExtLaunchKernelGGL((k_integrate<false, false>), grid, block, 0, e->stream, ev0, ev1, 0, ip);
  }
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

int launch_integrate(dslam_engine *e, dslam_scene *s, const dslam_view *v, const dslam_render_state *r,
                     const float *M_d, const float *intr_d, const float *M_rgb, const float *intr_rgb,
                     bool deintegrate, int push_ring) {
  int rc = ensure_view_depth(e, v);
  if (rc) return rc;
  IntegrateParams ip;
  fill_params(ip, e, s, v, r, M_d, intr_d, M_rgb, intr_rgb);
  ip.masks = s->masks; ip.last_seen = s->last_seen; ip.push_words = 0; ip.push_ring = 0; ip.push_bit = 0; ip.push_frame = 0;
  ip.push_job_min = e->push_job_min;
  ip.spec_ids = r->n_local >= kIntegrateGrid * kWgWaves ? 1 : 0;  // (a visible list has room for every voxel-block slot)
  if (push_ring >= 0) {
    if ((rc = prepare_push_visible_list(e, s, push_ring, &ip.push_bit, &ip.push_frame))) return rc;
    ip.push_words = s->history_words; ip.push_ring = push_ring;
  }
  // (the visible count as the host last heard of it: dslam_render_state::vis_hint)
  const bool stream = r->vis_hint && __atomic_load_n(r->vis_hint, __ATOMIC_RELAXED) >= e->push_job_min;
  return launch_integrate_params(e, ip, deintegrate, stream);
}

int launch_integrate_list(dslam_engine *e, dslam_scene *s, const dslam_view *v, const void *count_header, const int *ids,
                          const short4 *expect_pos, const float *M_d, const float *intr_d, const float *M_rgb,
                          const float *intr_rgb, bool deintegrate) {
  int rc = ensure_view_depth(e, v);
  if (rc) return rc;
  IntegrateParams ip;
  // (fill_params only takes the list pointer and the counter block from the render state)
  dslam_render_state list_view;
  list_view.visible_ids = const_cast<int *>(ids);
  list_view.counters = reinterpret_cast<RenderCounters *>(const_cast<void *>(count_header));
  fill_params(ip, e, s, v, &list_view, M_d, intr_d, M_rgb, intr_rgb);
  ip.expect_pos = expect_pos;
  ip.masks = s->masks; ip.last_seen = s->last_seen; ip.push_words = 0; ip.push_ring = 0; ip.push_bit = 0; ip.push_frame = 0;
  ip.push_job_min = e->push_job_min;
  return launch_integrate_params(e, ip, deintegrate, false);
}

// =========================================================================================================================
// Block-major re-integration batch (dslam_reintegrate_batch)
// =========================================================================================================================
// Reference: DenseSlam::OnlineCorrection's loop (DenseSlam.cpp:389-403): for every corrected keyframe k, DeProcessFrame at
// its old pose, then ProcessFrame at the new one.  Run as written that is 2 K launches of k_integrate which each stream
// every visible block through HBM again: 32 keyframes visit ~512 k blocks (4.2 GB) of which 34 k are distinct (0.28 GB).
// A voxel's value depends only on its own history, so the batch is re-ordered BLOCK-major: the allocation passes of all K
// re-fusions run first, in keyframe order (they do not read voxels); then ONE launch takes every block the batch touches,
// loads its 4 KiB once, applies -- in keyframe order -- the de- and re-updates of every keyframe whose list names it, and
// stores it once.  Per voxel the sequence of operations is the one of the loop above: the result is bit-identical.
//   opmask[slot]   bit 2k: de-integration of keyframe k applies to the block in this slot (its stored fusion-time list
//                  names an entry that still holds the block, and the block existed before re-fusion k allocated);
//                  bit 2k + 1: the re-fusion of keyframe k lists it
//   ops[2k], ops[2k + 1]   pose and images of the two operations
// Blocks differ in how many operations they take: they are ordered longest first and dealt round-robin over the waves.
struct BatchOp {
  Mat4 M;                // world -> camera of this operation (old pose: de-integration, new pose: re-fusion)
  const float *depth;    // the keyframe's depth image in metres: written by the allocation pass of its re-fusion (k_mark derives
                         // it from the int16 image exactly as UpdateView does) into the batch's scratch, one image per keyframe
  const uchar4 *rgba;
  int push_bit, push_frame;  // re-fusions: the ring bit and frame stamp ProcessFrame(isDefusion) queues the block with
  int pad[2];
};

struct BatchParams {
  IntegrateParams ip;    // what does not change over the batch (intrinsics, mu, sizes, weights, shard, rings ...)
  const BatchOp *ops;
  const unsigned long long *opmask;
  const int *slot_entry;  // per slot: a hash entry that holds it (its block position)
  const int *cls_list;    // [kBatchClasses][n_local]: the slots with operations, by class of their operation count
  const int *cls_count;   // [kBatchClasses]
  int n_local;
  int n_ops;
  int grid_waves;   // waves of the launch (units are dealt round-robin over them)
};

// What an operation needs of its BatchOp, in SCALAR registers.  The table sits in LDS (a scalar load per operation costs a
// round trip of its own); what comes out of LDS is a vector register, and a buffer resource built from vector registers is
// "divergent" to the compiler: round 3's kernel wrapped every depth / texel load of an operation in a waterfall loop
// (v_readfirstlane x 4, two compares, an exec-mask loop around the load -- eight of them per block-operation, found in the
// ISA), and multiplied by pose terms it re-read from LDS at every use.  Here every value is made uniform once per
// operation (v_readfirstlane: 14 floats, two pointers), the resources are scalar, the pose terms scalar operands.
struct OpScalars {
  float m0, m1, m2, m4, m5, m6, m8, m9, m10, m12, m13, m14;
  __amdgpu_buffer_rsrc_t depth_rs, rgba_rs;
};
__device__ __forceinline__ float uniform_f(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ const void *uniform_ptr(const void *ptr) {
  const unsigned long long a = (unsigned long long)ptr;
  // (readfirstlane returns int: the low half must be widened as UNSIGNED, or a set bit 31 smears over the high half)
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)), lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)a);
  return (const void *)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}
__device__ __forceinline__ OpScalars op_scalars(const BatchOp &op, int Wd, int Hd, int Wr, int Hr) {
  OpScalars o;
  o.m0 = uniform_f(op.M.m[0]); o.m1 = uniform_f(op.M.m[1]); o.m2 = uniform_f(op.M.m[2]);
  o.m4 = uniform_f(op.M.m[4]); o.m5 = uniform_f(op.M.m[5]); o.m6 = uniform_f(op.M.m[6]);
  o.m8 = uniform_f(op.M.m[8]); o.m9 = uniform_f(op.M.m[9]); o.m10 = uniform_f(op.M.m[10]);
  o.m12 = uniform_f(op.M.m[12]); o.m13 = uniform_f(op.M.m[13]); o.m14 = uniform_f(op.M.m[14]);
  o.depth_rs = image_rsrc(uniform_ptr(op.depth), Wd, Hd);
  o.rgba_rs = image_rsrc(uniform_ptr(op.rgba), Wr, Hr);
  return o;
}

// fuse_colour_word / defuse_colour_word with the texels through an operation's own resource
template <bool DEINT>
__device__ __forceinline__ unsigned batch_colour_word(unsigned pack, float u, float w, __amdgpu_buffer_rsrc_t rgba_rs,
                                                      const IntegrateParams &p, const float *inv_tab) {
  float m[3];
  if constexpr (DEINT) {
    const unsigned wc = pack >> 24;
    if (wc < 1) return pack;  // nothing was ever fused into this colour
    bilinear_rgb_buf(rgba_rs, u, w, p.Wr, m);
    const float oldW = (float)wc, remW = oldW - 1.0f;
    if (remW == 0.0f) return 0u;  // colour 0, weight 0
    const float inv_rem = inv_tab[wc - 1];
    unsigned out = (unsigned)(unsigned char)remW << 24;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float oldC = div_exact((float)((pack >> (8 * k)) & 0xffu), 255.0f, p.inv_255);
      const float c = div_exact(m[k], 255.0f, p.inv_255);
      float v = div_exact(oldC * oldW - c * 1.0f, remW, inv_rem);
      v = fmaxf(0.0f, fminf(1.0f, v));
      out |= (unsigned)(unsigned char)(v * 255.0f) << (8 * k);
    }
    return out;
  } else {
    bilinear_rgb_buf(rgba_rs, u, w, p.Wr, m);
    const unsigned wc = pack >> 24;
    const float oldW = (float)wc;
    float newW = oldW + 1.0f;
    const float inv_new = inv_tab[wc + 1];
    unsigned out = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float oldC = div_exact((float)((pack >> (8 * k)) & 0xffu), 255.0f, p.inv_255);
      const float c = div_exact(m[k], 255.0f, p.inv_255);
      const float v = div_exact(oldC * oldW + c * 1.0f, newW, inv_new);
      out |= (unsigned)(unsigned char)(v * 255.0f) << (8 * k);
    }
    newW = fminf(newW, (float)p.max_w);
    return out | ((unsigned)(unsigned char)newW << 24);
  }
}

// The colour queue of k_integrate, one per wave, with room for a whole block (kHalves = 2: the narrow-band voxels of BOTH
// halves of an operation are queued and run in ONE dense pass -- two half-empty passes of ~120 instructions per block-
// operation were a quarter of the kernel's arithmetic).  A voxel's data sits at its own place (half * 256 + chunk-voxel
// k * 64 + lane), `list` holds the places in queue order, the result comes back in `c`.
template <int kHalves>
struct BatchColQueue {
  float u[256 * kHalves], w[256 * kHalves];
  unsigned c[256 * kHalves];
  unsigned short list[256 * kHalves];
};

// The depth updates of one operation on one half block (two 16-byte chunks per lane): both chunks are projected and their
// four depth pixels requested before the first update waits; the narrow-band voxels are queued (place = qbase + ...).
template <bool DEINT, bool UNIT_W, int kHalves>
__device__ __forceinline__ void batch_op_half(uint4 (&v)[2], bool (&chs)[2], int gz0, const float (&pxy)[2][3], const OpScalars &os,
                                              const IntegrateParams &p, const float *inv_tab, BatchColQueue<kHalves> &Q, int lane,
                                              int qbase, int &q_n, unsigned &cms) {
  PairProj pq[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const float fz = (float)(gz0 + j * 2) * p.voxel_size;   // (gz0: the z of this lane's voxels in the half's first chunk)
    const float az0 = os.m8 * fz, az1 = os.m9 * fz, az2 = os.m10 * fz;
    const f2 a0 = {az0, az0}, a1 = {az1, az1}, a2 = {az2, az2};
    const f2 t0 = {os.m12, os.m12}, t1 = {os.m13, os.m13}, t2 = {os.m14, os.m14};
    const f2 px = {pxy[0][0], pxy[1][0]}, py = {pxy[0][1], pxy[1][1]}, pz = {pxy[0][2], pxy[1][2]};
    pair_project(pq[j], (px + a0) + t0, (py + a1) + t1, (pz + a2) + t2, p, os.depth_rs);
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const PairProj &q = pq[j];
    unsigned cm;
    chs[j] |= pair_update<DEINT, UNIT_W>(v[j], q, p, inv_tab, cm);
    cms |= cm << (2 * j);
    if (__ballot(cm != 0u)) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const bool c = (cm >> h) & 1u;
        const unsigned long long bm = __ballot(c);
        if (c) {
          const unsigned lo = h ? v[j].z : v[j].x, hi = h ? v[j].w : v[j].y;
          const int own = qbase + (j * 2 + h) * 64 + lane;
          const int slot = q_n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0u));
          Q.u[own] = h ? q.u.y : q.u.x;
          Q.w[own] = h ? q.w.y : q.w.x;
          Q.c[own] = __builtin_amdgcn_perm(hi, lo, 0x06050403u);
          Q.list[slot] = (unsigned short)own;
        }
        q_n += __popcll(bm);
      }
    }
  }
}

// the queued colour updates, one per lane and round; every result goes back to the place of its voxel
template <bool DEINT, int kHalves>
__device__ __forceinline__ void batch_colour_pass(int q_n, const OpScalars &os, const IntegrateParams &p, const float *inv_tab,
                                                  BatchColQueue<kHalves> &Q, int lane) {
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < q_n; i += 64) {
    const int o = Q.list[i];
    Q.c[o] = batch_colour_word<DEINT>(Q.c[o], Q.u[o], Q.w[o], os.rgba_rs, p, inv_tab);   // (read and written by the same lane)
  }
  __builtin_amdgcn_wave_barrier();
}
template <int kHalves>
__device__ __forceinline__ void batch_colour_fetch(uint4 (&v)[2], unsigned cms, const BatchColQueue<kHalves> &Q, int qbase, int lane) {
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int h = 0; h < 2; h++)
      if ((cms >> (2 * j + h)) & 1u) {
        const unsigned word = Q.c[qbase + (j * 2 + h) * 64 + lane];
        unsigned &lo = h ? v[j].z : v[j].x;
        unsigned &hi = h ? v[j].w : v[j].y;
        lo = __builtin_amdgcn_perm(word, lo, 0x04020100u);
        hi = __builtin_amdgcn_perm(hi, word, 0x07030201u);
      }
}

// one operation on the unit a wave holds (kHalves halves of a block)
template <bool DEINT, bool UNIT_W, int kHalves>
__device__ __forceinline__ void batch_op(uint4 (&v)[kHalves][2], bool (&chs)[kHalves][2], int gz0, const float (&pxy)[2][3],
                                         const OpScalars &os, const IntegrateParams &p, const float *inv_tab,
                                         BatchColQueue<kHalves> &Q, int lane) {
  int q_n = 0;
  unsigned cms[kHalves];
#pragma unroll
  for (int hf = 0; hf < kHalves; hf++) {
    cms[hf] = 0;
    batch_op_half<DEINT, UNIT_W, kHalves>(v[hf], chs[hf], gz0 + hf * 4, pxy, os, p, inv_tab, Q, lane, hf * 256, q_n, cms[hf]);
  }
  if (q_n > 0) {
    batch_colour_pass<DEINT, kHalves>(q_n, os, p, inv_tab, Q, lane);
#pragma unroll
    for (int hf = 0; hf < kHalves; hf++) batch_colour_fetch<kHalves>(v[hf], cms[hf], Q, hf * 256, lane);
    __builtin_amdgcn_wave_barrier();   // (the queue is refilled by the next operation)
  }
}

// (-DDSLAM_BATCH_WAVES / -DDSLAM_BATCH_MIN_WAVES / -DDSLAM_BATCH_UNSHARDED_HALVES: occupancy experiments of round 4)
#ifndef DSLAM_BATCH_WAVES
#define DSLAM_BATCH_WAVES 8
#endif
#ifndef DSLAM_BATCH_MIN_WAVES
#define DSLAM_BATCH_MIN_WAVES 1
#endif
#ifndef DSLAM_BATCH_UNSHARDED_HALVES
#define DSLAM_BATCH_UNSHARDED_HALVES 2
#endif
constexpr int kBatchWgWaves = DSLAM_BATCH_WAVES;
// 4096 waves = what is resident at once (4 per SIMD at this register count).  Units are dealt round-robin over the list
// ordered longest first -- NOT fetched from a device counter: ~77 k fetches from one address serialise at ~12 ns each,
// which was 0.9 ms of a 1.2 ms launch and the whole launch of a rank that owns an eighth of the blocks.
constexpr int kBatchGrid = 512;

// kHalves = 2: the unit of work is a block -- the operation's set-up (its scalars, the block's xy terms) and its colour pass
// are paid once per block.  kHalves = 1: half a block per unit, for a rank that owns a fraction of the blocks -- the launch
// then ends with the longest chain of operations on one unit, and a half block's is half as long (rank 0 of 8: 0.46
// against 0.52 ms in round 3).
template <bool UNIT_W, int kHalves>
__global__ __launch_bounds__(kBatchWgWaves * 64, DSLAM_BATCH_MIN_WAVES) void k_reintegrate_blocks(BatchParams bp) {
  __shared__ float inv_tab[kInvTab];
  __shared__ BatchColQueue<kHalves> col_q[kBatchWgWaves];
  __shared__ BatchOp s_ops[64];   // the batch's operations: read per operation from LDS, not with a ~1 us scalar load each
  __shared__ int s_cum[9];        // blocks in the classes in front of class c (k_batch_assemble's lists)
  if (threadIdx.x == 0) {
    int run = 0;
    for (int c = 0; c < 8; c++) { s_cum[c] = run; run += bp.cls_count[c]; }
    s_cum[8] = run;
  }
  for (int i = threadIdx.x; i < kInvTab; i += kBatchWgWaves * 64) inv_tab[i] = recip_table_entry(i);
  for (int i = threadIdx.x; i < bp.n_ops * (int)(sizeof(BatchOp) / 4); i += kBatchWgWaves * 64)
    reinterpret_cast<unsigned *>(s_ops)[i] = reinterpret_cast<const unsigned *>(bp.ops)[i];
  __syncthreads();
  const IntegrateParams &p0 = bp.ip;
  const int lane = threadIdx.x & 63;
  BatchColQueue<kHalves> &Q = col_q[threadIdx.x >> 6];
  const int n = __builtin_amdgcn_readfirstlane(s_cum[8]) * (2 / kHalves);
  const int vx0 = (lane & 3) * 2, vy = (lane >> 2) & 7, vz0 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * (kBatchWgWaves * 64) + threadIdx.x) >> 6));
  for (int i = wave; i < n; i += bp.grid_waves) {
    const int bi = kHalves == 2 ? i : (i >> 1);
    int cls = 0;
#pragma unroll
    for (int c = 1; c < 8; c++) cls += bi >= s_cum[c] ? 1 : 0;
    const int ptr = __builtin_amdgcn_readfirstlane(bp.cls_list[(size_t)cls * bp.n_local + (bi - s_cum[cls])]);
    const int half = kHalves == 2 ? 0 : (i & 1);
    const unsigned long long mask = bp.opmask[ptr];
    // (before the shard test, as in k_integrate: every rank of a sharded batch ends up with the same marks and the same
    // rings) the block joins the list of every re-fusion that names it on the defusion ring
    if (lane == 0 && half == 0) {
      if (p0.dirty) p0.dirty[ptr] = 1;
      if (p0.push_words) {
        // (one read-modify-write per ring word, not one per re-fusion: a chain of ~16 dependent round trips per block on
        // one lane, for every block on every rank, was 0.3 ms of the launch)
        unsigned long long *ring = p0.masks + ((size_t)ptr * 2 + p0.push_ring) * p0.push_words;
        const unsigned long long fmask = mask & 0xAAAAAAAAAAAAAAAAull;
        int frame = -1;
        for (int w = 0; w < p0.push_words; w++) {
          unsigned long long acc = 0;
          for (unsigned long long m = fmask; m; m &= m - 1) {
            const BatchOp &op = s_ops[__ffsll((long long)m) - 1];
            if ((op.push_bit >> 6) == w) acc |= 1ull << (op.push_bit & 63);
            frame = op.push_frame;   // (ascending: the last re-fusion that lists the block)
          }
          if (acc) ring[w] |= acc;
        }
        if (frame >= 0) p0.last_seen[ptr] = frame;
      }
    }
    if (p0.num_shards > 1 && ((ptr / p0.chunk_blocks) % p0.num_shards) != p0.shard) continue;
    if (p0.shard_count >= 0 && (ptr < p0.shard_first || ptr >= p0.shard_first + p0.shard_count)) continue;
    const HashEntry e = load_entry(p0.hash, bp.slot_entry[ptr]);
    const int gx = __builtin_amdgcn_readfirstlane((int)e.pos[0]) * kBlock, gy = __builtin_amdgcn_readfirstlane((int)e.pos[1]) * kBlock,
              gz = __builtin_amdgcn_readfirstlane((int)e.pos[2]) * kBlock;
    uint4 *blk = p0.voxels16 + (size_t)ptr * (kBlock3 / 2) + half * 128;
    uint4 v[kHalves][2];
    bool chs[kHalves][2];
#pragma unroll
    for (int hf = 0; hf < kHalves; hf++) {
      v[hf][0] = blk[hf * 128 + lane];
      v[hf][1] = blk[hf * 128 + 64 + lane];
      chs[hf][0] = false; chs[hf][1] = false;
    }
    const int gz0 = gz + half * 4 + vz0;
    const float fyv = (float)(gy + vy) * p0.voxel_size;
    float fxv[2];
    fxv[0] = (float)(gx + vx0) * p0.voxel_size;
    fxv[1] = (float)(gx + vx0 + 1) * p0.voxel_size;
    const unsigned mlo = (unsigned)mask, mhi = (unsigned)(mask >> 32);
    for (int part = 0; part < 2; part++) {
      for (unsigned m = part ? mhi : mlo; m; m &= m - 1) {
        const int bit = __builtin_amdgcn_readfirstlane(part * 32 + __ffs((int)m) - 1);
        const OpScalars os = op_scalars(s_ops[bit], p0.Wd, p0.Hd, p0.Wr, p0.Hr);
        float pxy[2][3];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          pxy[h][0] = os.m0 * fxv[h] + os.m4 * fyv;
          pxy[h][1] = os.m1 * fxv[h] + os.m5 * fyv;
          pxy[h][2] = os.m2 * fxv[h] + os.m6 * fyv;
        }
        if (bit & 1) batch_op<false, UNIT_W, kHalves>(v, chs, gz0, pxy, os, p0, inv_tab, Q, lane);   // re-fusion at the new pose
        else batch_op<true, UNIT_W, kHalves>(v, chs, gz0, pxy, os, p0, inv_tab, Q, lane);            // de-integration at the old one
      }
    }
#pragma unroll
    for (int hf = 0; hf < kHalves; hf++) {
      if (chs[hf][0]) blk[hf * 128 + lane] = v[hf][0];
      if (chs[hf][1]) blk[hf * 128 + 64 + lane] = v[hf][1];
    }
  }
}

// Which operations touch which block.  One workgroup row per operation (blockIdx.y = 2k / 2k + 1), lanes over that
// operation's list: the keyframe's stored fusion-time list (de-integration; entries that no longer hold that block, or
// whose block the batch itself allocated at or after re-fusion k, are skipped -- what dslam_deprocess_frame_stored would
// have found at that point of the sequence) or the list the re-fusion's allocation pass left (kept in the batch's scratch).
struct BatchListRef {
  const RenderCounters *count;   // header of the list (no_visible = its length)
  const int *ids;
  const short4 *pos;             // stored lists: the block each entry held at fusion time; null: a fresh list
};
__global__ __launch_bounds__(256) void k_batch_mark(const BatchListRef *__restrict__ lists, const HashEntry *__restrict__ hash,
                                                    const int *__restrict__ born, unsigned char *marks, int *slot_entry) {
  const int op = blockIdx.y, k = op >> 1;
  const BatchListRef L = lists[op];
  const int n = L.count->no_visible;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int t = L.ids[i];
    const HashEntry e = load_entry(hash, t);
    if (e.ptr < 0) continue;
    if (L.pos) {
      const short4 ep = L.pos[i];
      const int b = born[e.ptr];   // (requested with the position)
      if (ep.x != e.pos[0] || ep.y != e.pos[1] || ep.z != e.pos[2]) continue;
      if (b > k) continue;   // allocated by re-fusion k or a later one: did not exist when keyframe k was de-integrated
    }
    // One byte per (block, operation), 64 of them next to each other: plain stores.  (A 64-bit mask per block built with
    // atomicOr -- ~0.5 M device-scope atomics per batch, one per list entry -- took 76 us; these stores and the pass that
    // gathers them take ~20.)
    marks[(size_t)e.ptr * 64 + op] = 1;
    slot_entry[e.ptr] = t;   // (whoever lists the block names the entry that holds it)
  }
}

// The marks of every block slot gathered into its operation mask; the blocks with operations are listed by CLASS of their
// operation count (kBatchClasses lists, most operations first): the block launch deals its units in that order, so that a
// block with 60 operations is not what the launch ends with.  Four lanes per slot (16 marks each); the marks are cleared on
// the way, the next batch finds them zero.
constexpr int kBatchClasses = 8;
constexpr int kAsmIter = 16;   // 16-byte pieces per thread: a workgroup gathers 1024 slots and updates each class counter ONCE
__global__ __launch_bounds__(256) void k_batch_assemble(uint4 *marks16, int n_slots, unsigned long long *opmask, int *cls_list,
                                                        int *cls_count) {
  // (one counter update per workgroup and class: updates of one address from all over the device serialise at ~12 ns each --
  // per wave and class they were 100 us of this launch)
  __shared__ int s_cnt[kBatchClasses], s_base[kBatchClasses];
  __shared__ int s_ops_total;   // (diagnostics: the batch's block-operations, cls_count[kBatchClasses])
  if (threadIdx.x < kBatchClasses) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_ops_total = 0;
  __syncthreads();
  const int quarter = threadIdx.x & 3;
  int cls[kAsmIter], pos[kAsmIter];
#pragma unroll
  for (int it0 = 0; it0 < kAsmIter; it0 += 4) {
    uint4 m[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int gid = (blockIdx.x * kAsmIter + it0 + q) * 256 + threadIdx.x;
      m[q] = (gid >> 2) < n_slots ? marks16[gid] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int gid = (blockIdx.x * kAsmIter + it0 + q) * 256 + threadIdx.x;
      const int slot = gid >> 2;
      unsigned m16 = 0;
      if (m[q].x | m[q].y | m[q].z | m[q].w) {
        const unsigned w[4] = {m[q].x, m[q].y, m[q].z, m[q].w};
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int bt = 0; bt < 4; bt++) m16 |= ((w[j] >> (8 * bt)) & 0xffu) ? (1u << (j * 4 + bt)) : 0u;
        marks16[gid] = make_uint4(0, 0, 0, 0);
      }
      // the four quarters of a slot sit in neighbouring lanes
      unsigned lo = quarter < 2 ? (m16 << (16 * quarter)) : 0u, hi = quarter >= 2 ? (m16 << (16 * (quarter - 2))) : 0u;
      lo |= __shfl_xor(lo, 1, 64); hi |= __shfl_xor(hi, 1, 64);
      lo |= __shfl_xor(lo, 2, 64); hi |= __shfl_xor(hi, 2, 64);
      const unsigned long long mask = ((unsigned long long)hi << 32) | lo;
      cls[it0 + q] = -1; pos[it0 + q] = 0;
      if (quarter == 0 && mask != 0ull) {
        opmask[slot] = mask;
        const int c = (64 - __popcll(mask)) >> 3;   // 0: 57..64 operations ... 7: 1..8
        cls[it0 + q] = c;
        pos[it0 + q] = atomicAdd(&s_cnt[c], 1);
        atomicAdd(&s_ops_total, __popcll(mask));
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < kBatchClasses) {
    const int c = s_cnt[threadIdx.x];
    s_base[threadIdx.x] = c ? atomicAdd(&cls_count[threadIdx.x], c) : 0;
  }
  if (threadIdx.x == 0 && s_ops_total) atomicAdd(&cls_count[kBatchClasses], s_ops_total);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kAsmIter; it++)
    if (cls[it] >= 0) cls_list[(size_t)cls[it] * n_slots + s_base[cls[it]] + pos[it]] = ((blockIdx.x * kAsmIter + it) * 256 + threadIdx.x) >> 2;
}

int launch_batch_ops(dslam_engine *e, const void *lists_dev, int n_ops, const dslam_scene *s, const int *born, unsigned char *marks,
                     unsigned long long *opmask, int *slot_entry, int *cls_list, int *cls_count) {
  const int L = s->p.num_local_blocks;
  hipLaunchKernelGGL(k_batch_mark, dim3(32, n_ops), dim3(256), 0, e->stream, reinterpret_cast<const BatchListRef *>(lists_dev), s->hash,
                     born, marks, slot_entry);
  hipLaunchKernelGGL(k_batch_assemble, dim3((L * 4 + 256 * kAsmIter - 1) / (256 * kAsmIter)), dim3(256), 0, e->stream, reinterpret_cast<uint4 *>(marks), L, opmask,
                     cls_list, cls_count);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

int launch_reintegrate_blocks(dslam_engine *e, dslam_scene *s, int w_d, int h_d, int w_rgb, int h_rgb, const float *intr,
                              const void *ops_dev, const unsigned long long *opmask, const int *slot_entry,
                              const int *cls_list, const int *cls_count, int push_ring, int n_ops) {
  BatchParams bp;
  IntegrateParams &ip = bp.ip;
  memset(&ip, 0, sizeof(ip));
  ip.hash = s->hash;
  ip.voxels16 = reinterpret_cast<uint4 *>(s->voxels);
  ip.Wd = w_d; ip.Hd = h_d; ip.Wr = w_rgb; ip.Hr = h_rgb;
  ip.fx_d = intr[0]; ip.fy_d = intr[1]; ip.cx_d = intr[2]; ip.cy_d = intr[3];
  ip.fx_r = intr[0]; ip.fy_r = intr[1]; ip.cx_r = intr[2]; ip.cy_r = intr[3];
  ip.voxel_size = s->p.voxel_size; ip.mu = s->p.mu; ip.max_w = s->p.max_w;
  ip.inv_32767 = 1.0f / 32767.0f; ip.inv_255 = 1.0f / 255.0f;
  ip.same_cam = 1;
  ip.stop_max = 0;
  ip.depth_weighting = e->wp.depth_weighting; ip.max_new_w = e->wp.max_new_w; ip.max_distance = e->wp.max_distance;
  ip.shard = s->shard; ip.num_shards = s->num_shards; ip.chunk_blocks = s->chunk_blocks;
  ip.shard_first = s->shard_first; ip.shard_count = s->shard_count;
  ip.dirty = s->dirty_tracking ? s->dirty : nullptr;
  ip.masks = s->masks; ip.last_seen = s->last_seen;
  ip.push_words = push_ring >= 0 ? s->history_words : 0;
  ip.push_ring = push_ring >= 0 ? push_ring : 0;
  bp.ops = reinterpret_cast<const BatchOp *>(ops_dev);
  bp.opmask = opmask; bp.slot_entry = slot_entry; bp.cls_list = cls_list; bp.cls_count = cls_count; bp.n_local = s->p.num_local_blocks;
  bp.n_ops = n_ops;
  const bool sharded = ip.num_shards > 1 || ip.shard_count >= 0;
  static const int grid_wgs = getenv("DSLAM_BATCH_GRID") ? atoi(getenv("DSLAM_BATCH_GRID")) : kBatchGrid;   // (experiments: occupancy of the block launch)
  bp.grid_waves = grid_wgs * kBatchWgWaves;
  const dim3 grid(grid_wgs), block(kBatchWgWaves * 64);
  if (ip.depth_weighting) {
    if (sharded) hipLaunchKernelGGL((k_reintegrate_blocks<false, 1>), grid, block, 0, e->stream, bp);
    else hipLaunchKernelGGL((k_reintegrate_blocks<false, DSLAM_BATCH_UNSHARDED_HALVES>), grid, block, 0, e->stream, bp);
  } else {
    if (sharded) hipLaunchKernelGGL((k_reintegrate_blocks<true, 1>), grid, block, 0, e->stream, bp);
    else hipLaunchKernelGGL((k_reintegrate_blocks<true, DSLAM_BATCH_UNSHARDED_HALVES>), grid, block, 0, e->stream, bp);
  }
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

// the render state's visible list, with the block position of every entry, into a keyframe's list slot
__global__ __launch_bounds__(256) void k_store_visible_list(const int *__restrict__ ids, const RenderCounters *rc,
                                                            const HashEntry *__restrict__ hash, RenderCounters *header,
                                                            int *__restrict__ out_ids, short4 *__restrict__ out_pos, int capacity) {
  int n = rc->no_visible;
  n = n < capacity ? n : capacity;
  if (blockIdx.x == 0 && threadIdx.x == 0) { RenderCounters h = {}; h.no_visible = n; *header = h; }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int t = ids[i];
    const HashEntry e = load_entry(hash, t);
    out_ids[i] = t;
    out_pos[i] = make_short4(e.pos[0], e.pos[1], e.pos[2], 0);
  }
}

// the block positions of lists that were written without them (the lists of the re-integration batch's allocation passes:
// an entry keeps its block from its allocation to the end of the batch, so one launch at the end serves all passes)
__global__ __launch_bounds__(256) void k_store_list_positions(const BatchListRef *__restrict__ jobs, const HashEntry *__restrict__ hash) {
  const BatchListRef L = jobs[blockIdx.y];
  const int n = L.count->no_visible;
  short4 *out = const_cast<short4 *>(L.pos);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const HashEntry e = load_entry(hash, L.ids[i]);
    out[i] = make_short4(e.pos[0], e.pos[1], e.pos[2], 0);
  }
}
int launch_store_list_positions(dslam_engine *e, const dslam_scene *s, const void *jobs_dev, int n_jobs) {
  if (n_jobs <= 0) return DSLAM_OK;
  hipLaunchKernelGGL(k_store_list_positions, dim3(32, n_jobs), dim3(256), 0, e->stream, reinterpret_cast<const BatchListRef *>(jobs_dev), s->hash);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

int launch_store_visible_list(dslam_engine *e, const dslam_scene *s, const dslam_render_state *r, void *header, int *ids,
                              short4 *pos, int capacity) {
  hipLaunchKernelGGL(k_store_visible_list, dim3(64), dim3(256), 0, e->stream, r->visible_ids, r->counters, s->hash,
                     reinterpret_cast<RenderCounters *>(header), ids, pos, capacity);
  DSLAM_HIP(hipGetLastError());
  return DSLAM_OK;
}

}  // namespace dslam
