// warp.hip - STN homography warp of the court template and POI projection.
//
// Replaces, for Reconstructor.warp / transform_poi (models/reconstructor.py:109-130), the
// Kornia chain create_meshgrid -> transform_points -> convert_points_from_homogeneous ->
// F.grid_sample(padding 'zeros', align_corners=False), fused with predict()'s
// `* mask_classes` and `.type(torch.int32)` (models/reconstructor.py:223,240).
//
// Compulsory traffic per batch: B*h*w*4 B of output + one template image (the template is one image
// replicated over the batch: utils/dataset.py:59) + 36 B of theta per frame.  The coordinate arithmetic
// uses individually rounded fp32 operations (__fmul_rn/__fadd_rn, no FMA contraction except where the
// reference's own kernel fuses) in exactly the order of oracle/warp_ref.py, so nearest-mode results are
// integer-identical to the oracle.  Work decomposition and the instruction-count choices: see the comment
// in front of warp2_body.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

// waves a launch should put on the chip before a thread takes more rows (experiment knob of profiles/build_variant.py)
#ifndef SFH_WARP_MIN_WAVES
#define SFH_WARP_MIN_WAVES 16384
#endif

namespace {

struct Homog {
  float t[9];
};

__device__ __forceinline__ void apply_h(const Homog& H, float x, float y, float& u, float& v) {
  const float X = __fadd_rn(__fadd_rn(__fmul_rn(H.t[0], x), __fmul_rn(H.t[1], y)), H.t[2]);
  const float Y = __fadd_rn(__fadd_rn(__fmul_rn(H.t[3], x), __fmul_rn(H.t[4], y)), H.t[5]);
  const float Z = __fadd_rn(__fadd_rn(__fmul_rn(H.t[6], x), __fmul_rn(H.t[7], y)), H.t[8]);
  const float s = (fabsf(Z) > 1e-8f) ? __fdiv_rn(1.0f, __fadd_rn(Z, 1e-8f)) : 1.0f;
  u = __fmul_rn(s, X);
  v = __fmul_rn(s, Y);
}

__device__ __forceinline__ float norm_axis(int i, int n) {
  // create_meshgrid: (i/(n-1) - 0.5) * 2
  return __fmul_rn(__fsub_rn(__fdiv_rn((float)i, (float)(n - 1)), 0.5f), 2.0f);
}

__device__ __forceinline__ float unnorm(float c, int size) {
  // ATen CPU grid sampler, align_corners=False: fma(fl(c + 1), size/2, -0.5) - the rounding
  // that reproduces torch's F.grid_sample bit for bit (see oracle/warp_ref.py:unnormalize).
  return __builtin_fmaf(__fadd_rn(c, 1.0f), 0.5f * (float)size, -0.5f);
}

__device__ __forceinline__ float fetch(const float* __restrict__ tm, float fx, float fy, int wt, int ht) {
  // fx, fy are integral-valued floats (or NaN/inf): in range -> template value, else 0
  if (fx >= 0.f && fx <= (float)(wt - 1) && fy >= 0.f && fy <= (float)(ht - 1))
    return tm[(int)fy * wt + (int)fx];
  return 0.f;
}

// ---------------------------------------------------------------------------------------------
// Second-generation forward kernel: the same arithmetic bit for bit in about half the vector instructions.
// The kernel is VALU-issue-bound (profiles/r02_warp_*.txt), so every choice below removes vector
// instructions per pixel:
//
// * 1/(Z + 1e-8): hardware reciprocal (1 ulp) + two Newton steps in FMA arithmetic = the correctly
//   rounded quotient for every |z| in [2^-64, 2^64] (exhaustive sweep on the GPU: sfh_selftest_warp_arith);
//   i/(n-1) of create_meshgrid: q0 = i*r and one FMA residual step with the correctly rounded r = 1/(n-1)
//   (exhaustive for every n <= 16384).  A wave-uniform test on theta selects this path; huge or non-finite
//   matrices take the IEEE divisions.
// * A wave covers 64*J consecutive pixels of RPT rows, lane -> pixel lane + 64*j: every store is one
//   coalesced 256-byte wave instruction, the taps of one instruction fall into one or two cache lines,
//   and t0*xn, t3*xn, t6*xn are row-invariant - a thread keeps them in registers over its rows, so a
//   homogeneous coordinate costs two additions per pixel.  The row constants (t1*yn, t4*yn, t7*yn) are
//   computed once by lane rr of the wave and broadcast through v_readlane (scalar operands afterwards).
// * (NOLOAD is a measurement variant of profiles/micro/warp_variants.hip: taps are not fetched.)
// * Taps go through a buffer descriptor on the template: an invalid tap gets an out-of-range offset and
//   the hardware returns 0 (grid_sample's zeros padding) - no divergent branches, all taps of a row in
//   flight together.
// * When |t8| - |t6| - |t7| > 0 by a margin, |Z| > 1e-8 on the whole frame and the select between
//   1/(Z + 1e-8) and 1 disappears (wave-uniform).
// Packed fp32 instructions (v_pk_add/mul/fma_f32, two rows per instruction) were measured SLOWER
// (4.31 vs 4.72 TB/s at 640x360 B=1024): they issue at half rate here.

// LEVEL 0: IEEE divisions (any theta); 1: fast reciprocal; 2: fast reciprocal and |Z| > 1e-8 everywhere
template <int LEVEL>
__device__ __forceinline__ float recip_rn(float z) {
  if (LEVEL == 0) return __fdiv_rn(1.0f, z);
  // caller guarantees |z| <= 2^60 (a tiny or zero z gives a result the caller discards)
  float r = __builtin_amdgcn_rcpf(z);
  float e = __builtin_fmaf(-z, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  e = __builtin_fmaf(-z, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}

// i / d for integral 0 <= i <= d < 2^14, rd = __fdiv_rn(1, d)
__device__ __forceinline__ float div_small(float i, float d, float rd) {
  const float q0 = __fmul_rn(i, rd);
  return __builtin_fmaf(__builtin_fmaf(-q0, d, i), rd, q0);
}

template <bool SMALL>
__device__ __forceinline__ float norm_axis2(int i, int n, float rd) {
  const float q = SMALL ? div_small((float)i, (float)(n - 1), rd) : __fdiv_rn((float)i, (float)(n - 1));
  return __fmul_rn(__fsub_rn(q, 0.5f), 2.0f);
}

__device__ __forceinline__ float tap_ld(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0));
}

constexpr unsigned kTapOOB = 0xFFFFFFFCu;   // beyond any descriptor's num_records: the load returns 0

// rx, ry integral-valued floats -> byte offset of the tap or kTapOOB.  LEVEL 0 tolerates NaN / inf.
template <int LEVEL>
__device__ __forceinline__ unsigned tap_off(float rx, float ry, int wt, int ht) {
  if (LEVEL == 0) {
    const bool ok = (rx >= 0.f) & (rx <= (float)(wt - 1)) & (ry >= 0.f) & (ry <= (float)(ht - 1));
    const float fi = __builtin_fmaf(ry, (float)wt, rx);             // exact: < 2^24 (checked by the launcher)
    return ok ? ((unsigned)(int)fi << 2) : kTapOOB;
  }
  // finite coordinates: v_cvt_i32_f32 saturates, one unsigned compare per axis
  const int ix = (int)rx, iy = (int)ry;
  const bool ok = ((unsigned)ix < (unsigned)wt) & ((unsigned)iy < (unsigned)ht);
  return ok ? (__umul24((unsigned)iy, (unsigned)wt) + (unsigned)ix) << 2 : kTapOOB;   // valid => iy, wt < 2^24
}

// OUT: 0 = int32 mask only (predict), 1 = float only (forward / training), 2 = both
template <int MODE, int J, int RPT, int OUT, int LEVEL, bool SMALL, bool NOLOAD = false>
__device__ __forceinline__ void warp2_body(const float (&t)[9], int b, int lane, int c0, int r0,
                                           const float* __restrict__ tmpl, long tmpl_bstride,
                                           int ht, int wt, int h, int w, float rdw, float rdh, float out_scale,
                                           float* __restrict__ out_f, int32_t* __restrict__ out_i) {
  float a0[J], a3[J], a6[J];
  unsigned coff[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = c0 + 64 * j;
    const float xn = norm_axis2<SMALL>(c < w ? c : w - 1, w, rdw);
    a0[j] = __fmul_rn(t[0], xn);
    a3[j] = __fmul_rn(t[3], xn);
    a6[j] = __fmul_rn(t[6], xn);
    coff[j] = c < w ? (unsigned)c * 4u : kTapOOB;
  }
  // lane rr holds the row constants of row r0 + rr
  const float ynl = norm_axis2<SMALL>(r0 + (lane & (RPT - 1)), h, rdh);
  const float c1l = __fmul_rn(t[1], ynl), c4l = __fmul_rn(t[4], ynl), c7l = __fmul_rn(t[7], ynl);
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(tmpl + (long)b * tmpl_bstride), 0, ht * wt * 4, 0x00020000);
  const int nrows = (h - r0 < RPT) ? h - r0 : RPT;
  const long rowbase = ((long)b * h + r0) * w;
  const __amdgpu_buffer_rsrc_t rof = __builtin_amdgcn_make_buffer_rsrc(
      OUT != 0 ? out_f + rowbase : nullptr, 0, OUT != 0 ? nrows * w * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t roi = __builtin_amdgcn_make_buffer_rsrc(
      OUT != 1 ? out_i + rowbase : nullptr, 0, OUT != 1 ? nrows * w * 4 : 0, 0x00020000);
  const float sx = 0.5f * (float)wt, sy = 0.5f * (float)ht;
  constexpr int NT = MODE == 0 ? 1 : 4;
  unsigned off[J][NT];
  float wgt[J][NT], tv[J][NT];

  // tap offsets (and bilinear weights) of row r0 + rr
  auto coords = [&](int rr) {
    const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c1l), rr));
    const float c4 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c4l), rr));
    const float c7 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c7l), rr));
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float X = __fadd_rn(__fadd_rn(a0[j], c1), t[2]);
      const float Y = __fadd_rn(__fadd_rn(a3[j], c4), t[5]);
      const float Z = __fadd_rn(__fadd_rn(a6[j], c7), t[8]);
      const float r = recip_rn<LEVEL>(__fadd_rn(Z, 1e-8f));
      const float s = (LEVEL == 2 || fabsf(Z) > 1e-8f) ? r : 1.0f;
      // unnorm(): fma(fl(u + 1), size/2, -0.5)
      const float px = __builtin_fmaf(__fadd_rn(__fmul_rn(s, X), 1.0f), sx, -0.5f);
      const float py = __builtin_fmaf(__fadd_rn(__fmul_rn(s, Y), 1.0f), sy, -0.5f);
      if (MODE == 0) {
        off[j][0] = tap_off<LEVEL>(rintf(px), rintf(py), wt, ht);
      } else {
        const float x0 = floorf(px), y0 = floorf(py);
        const float wx1 = __fsub_rn(px, x0), wx0 = __fsub_rn(1.0f, wx1);
        const float wy1 = __fsub_rn(py, y0), wy0 = __fsub_rn(1.0f, wy1);
        wgt[j][0] = __fmul_rn(wy0, wx0);
        wgt[j][1] = __fmul_rn(wy0, wx1);
        wgt[j][2] = __fmul_rn(wy1, wx0);
        wgt[j][3] = __fmul_rn(wy1, wx1);
        if (LEVEL == 0) {
          off[j][0] = tap_off<0>(x0, y0, wt, ht);
          off[j][1] = tap_off<0>(x0 + 1.f, y0, wt, ht);
          off[j][2] = tap_off<0>(x0, y0 + 1.f, wt, ht);
          off[j][3] = tap_off<0>(x0 + 1.f, y0 + 1.f, wt, ht);
        } else {
          // the four taps share one address computation; per-axis validity of x0, x0+1, y0, y0+1
          const int ix = (int)x0, iy = (int)y0;   // saturating conversions of finite values
          const unsigned ux = (unsigned)ix, uy = (unsigned)iy, uw = (unsigned)wt, uh = (unsigned)ht;
          // ix, iy may be saturated: validity masks every use, the wrapped products are never used
          const unsigned o00 = (unsigned)(__mul24(iy, wt) + ix) << 2;   // 24-bit multiply: exact whenever a tap is valid
          off[j][0] = (ux < uw && uy < uh) ? o00 : kTapOOB;
          off[j][1] = (ux + 1u < uw && uy < uh) ? o00 + 4u : kTapOOB;
          off[j][2] = (ux < uw && uy + 1u < uh) ? o00 + uw * 4u : kTapOOB;
          off[j][3] = (ux + 1u < uw && uy + 1u < uh) ? o00 + uw * 4u + 4u : kTapOOB;
        }
      }
    }
  };

  // Software pipeline over the rows: the taps of row rr+1 are issued BEFORE the stores of row rr and are
  // consumed one row of coordinate arithmetic later - the wait for them never covers a younger store
  // (loads and stores share the in-order vmcnt counter) and their latency overlaps the thread's own work.
  auto issue_taps = [&]() {
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int k = 0; k < NT; ++k) tv[j][k] = NOLOAD ? __builtin_bit_cast(float, off[j][k] & 0x3fffffffu) : tap_ld(rt, off[j][k]);
  };
  coords(0);
  issue_taps();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int rr = 0; rr < RPT; ++rr) {
    // no early exit for rows beyond the frame (their coordinates are computed and discarded): a `break` here
    // lets the optimiser sink the prefetched taps back into the next row's block
    float val[J];
    float w0[J][NT];
    if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int k = 0; k < NT; ++k) w0[j][k] = wgt[j][k];
    }
    if (rr + 1 < RPT) coords(rr + 1);     // a row beyond the frame: harmless, its taps are never stored
#pragma unroll
    for (int j = 0; j < J; ++j) {
      val[j] = tv[j][0];
      if (MODE == 1) {
        val[j] = __fmul_rn(tv[j][0], w0[j][0]);
        val[j] = __fadd_rn(val[j], __fmul_rn(tv[j][1], w0[j][1]));
        val[j] = __fadd_rn(val[j], __fmul_rn(tv[j][2], w0[j][2]));
        val[j] = __fadd_rn(val[j], __fmul_rn(tv[j][3], w0[j][3]));
      }
    }
    if (rr + 1 < RPT) issue_taps();
    __builtin_amdgcn_sched_barrier(0);    // the scheduler would sink these loads to their uses one row later
    // A row beyond the frame is skipped explicitly (its offset rides in the scalar operand; on gfx950 the
    // descriptor's range check was observed to cover vector + scalar offset, but nothing here depends on it).
    // Wave-uniform branch around the stores only, so the prefetched taps above stay where they are.
    if (rr < nrows) {
      const int soff = rr * w * 4;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (OUT != 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val[j]), rof, (int)coff[j], soff, 0);
        if (OUT != 1)
          __builtin_amdgcn_raw_buffer_store_b32((unsigned)(int32_t)__fmul_rn(val[j], out_scale), roi, (int)coff[j], soff, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int MODE, int J, int RPT, int OUT, bool NOLOAD = false>
__global__ __launch_bounds__(256) void warp2_kernel(const float* __restrict__ theta,
                                                    const float* __restrict__ tmpl, long tmpl_bstride,
                                                    int ht, int wt, int h, int w, float rdw, float rdh, float out_scale,
                                                    float* __restrict__ out_f, int32_t* __restrict__ out_i) {
  // rdw = 1/(w-1), rdh = 1/(h-1) rounded to nearest (the launcher computes them: wave-uniform IEEE divisions
  // would cost every thread two dozen instructions)
  static_assert((RPT & (RPT - 1)) == 0 && RPT <= 64, "RPT: power of two");
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * (64 * J) + lane;
  const int r0 = (blockIdx.y * 4 + wv) * RPT;
  if (r0 >= h) return;
  float t[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) t[k] = theta[b * 9 + k];
  // wave-uniform classification of theta (|xn|, |yn| <= 1 on the whole frame)
  bool fin = true;
#pragma unroll
  for (int k = 0; k < 9; ++k) fin &= fabsf(t[k]) <= 0x1p59f;      // false for NaN
  const float zs = fabsf(t[6]) + fabsf(t[7]) + fabsf(t[8]);
  const bool live = (fabsf(t[8]) - fabsf(t[6]) - fabsf(t[7])) > 1e-6f * zs + 1e-7f;   // => |Z| > 1e-8 everywhere
#define SFH_WARP2_GO(LEVEL, SMALL) \
  warp2_body<MODE, J, RPT, OUT, LEVEL, SMALL, NOLOAD>(t, b, lane, c0, r0, tmpl, tmpl_bstride, ht, wt, h, w, rdw, rdh, out_scale, out_f, out_i)
  if (fin && w <= 16384 && h <= 16384) {
    if (live) SFH_WARP2_GO(2, true); else SFH_WARP2_GO(1, true);
  } else {
    SFH_WARP2_GO(0, false);
  }
#undef SFH_WARP2_GO
}

// ---------------------------------------------------------------------------------------------
// Nearest warp + consistency score in ONE launch (round 5; models/reconstructor.py:223-240 with warp_size == the logits'
// size): each wave warps its 64*J x RPT pixels exactly as warp2_body<0, ...> does (same instruction sequence for the
// coordinates, so the mask is bit-identical), keeps the class ids in registers, streams the NC logit planes of those pixels
// ONCE (coalesced 256-byte wave loads, prefetched one row ahead like the taps) and accumulates
//     logsumexp(logits[:, p]) - logits[mask[p], p]
// per lane; every wave leaves its sum in `partial`, and a second, tiny launch (warpce_final_kernel: one wave per frame) adds a
// frame's partials up in a fixed order in fp64 and writes the mean.  (The single-launch form - the last block of a frame,
// found through a device counter, does that sum - was built first and measured 2x SLOWER than the separate kernels, 68 us
// against 38 at 640x360 x 16: on this chip a device-scope release fence writes the XCD's L2 back, because the eight L2s are
// not coherent with each other, and every block paid for one; profiles/r05_warpce_sweep.txt.)
// Traffic = logits (4 * NC B / pixel) + mask (4 B / pixel) + one template: 74.6 MB at 640x360 x 16, 298 MB at 1280x720 x 16,
// against 15.7 / 62.7 MB for the warp alone and a second pass over mask + logits for the separate CE kernels.
// LS = 1: the logits have the warp's size.  LS = 2: the warp is twice the logits' size in both directions (predict.py's default
// geometry: UNet 640x360, warp 1280x720); the reference scores through F.interpolate(mask, mode='nearest'), i.e. logit pixel
// (y, x) against mask pixel (2y, 2x): the lanes at even x of the even rows score, the others only warp.
template <int J, int RPT, int NC, int LS, int LEVEL, bool SMALL>
__device__ __forceinline__ float warpce_body(const float (&t)[9], int b, int lane, int c0, int r0,
                                             const float* __restrict__ tmpl, long tmpl_bstride, int ht, int wt, int h, int w,
                                             float rdw, float rdh, float out_scale, const float* __restrict__ logits,
                                             int32_t* __restrict__ out_i) {
  float a0[J], a3[J], a6[J];
  unsigned coff[J], loff[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int c = c0 + 64 * j;
    const float xn = norm_axis2<SMALL>(c < w ? c : w - 1, w, rdw);
    a0[j] = __fmul_rn(t[0], xn);
    a3[j] = __fmul_rn(t[3], xn);
    a6[j] = __fmul_rn(t[6], xn);
    coff[j] = c < w ? (unsigned)c * 4u : kTapOOB;
    loff[j] = (c < w && (c % LS) == 0) ? (unsigned)(c / LS) * 4u : kTapOOB;     // the logit pixel this lane scores, if any
  }
  const float ynl = norm_axis2<SMALL>(r0 + (lane & (RPT - 1)), h, rdh);
  const float c1l = __fmul_rn(t[1], ynl), c4l = __fmul_rn(t[4], ynl), c7l = __fmul_rn(t[7], ynl);
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(tmpl + (long)b * tmpl_bstride), 0, ht * wt * 4, 0x00020000);
  const int nrows = (h - r0 < RPT) ? h - r0 : RPT;
  const long rowbase = ((long)b * h + r0) * w;
  const __amdgpu_buffer_rsrc_t roi = __builtin_amdgcn_make_buffer_rsrc(out_i + rowbase, 0, nrows * w * 4, 0x00020000);
  // the NC logit planes of this frame (hl x wl = the warp's size / LS): plane k, row r at byte (k * hl + r) * wl * 4 (below
  // 4 GiB: checked by the launcher)
  const int hl = h / LS, wl = w / LS;
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(logits + (long)b * NC * hl * wl), 0, NC * hl * wl * 4, 0x00020000);
  const float sx = 0.5f * (float)wt, sy = 0.5f * (float)ht;
  unsigned off[J];
  float tv[J], lg[2][J][NC];

  auto coords = [&](int rr) {
    const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c1l), rr));
    const float c4 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c4l), rr));
    const float c7 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c7l), rr));
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float X = __fadd_rn(__fadd_rn(a0[j], c1), t[2]);
      const float Y = __fadd_rn(__fadd_rn(a3[j], c4), t[5]);
      const float Z = __fadd_rn(__fadd_rn(a6[j], c7), t[8]);
      const float r = recip_rn<LEVEL>(__fadd_rn(Z, 1e-8f));
      const float s = (LEVEL == 2 || fabsf(Z) > 1e-8f) ? r : 1.0f;
      const float px = __builtin_fmaf(__fadd_rn(__fmul_rn(s, X), 1.0f), sx, -0.5f);
      const float py = __builtin_fmaf(__fadd_rn(__fmul_rn(s, Y), 1.0f), sy, -0.5f);
      off[j] = tap_off<LEVEL>(rintf(px), rintf(py), wt, ht);
    }
  };
  auto issue_taps = [&]() {
#pragma unroll
    for (int j = 0; j < J; ++j) tv[j] = tap_ld(rt, off[j]);
  };
  // logits for warp row r0 + rr (a scoring row: rr % LS == 0; r0 is a multiple of RPT, which is even) into register set
  // (rr / LS) & 1 (rows beyond the frame: the scalar offset stays inside the frame's planes or beyond the descriptor - either
  // way the values are never used)
  auto issue_logits = [&](int rr) {
    if (rr % LS != 0) return;
    const int set = (rr / LS) & 1;
    const int row = (r0 + rr < h ? r0 + rr : h - 1) / LS;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int soff = (k * hl + row) * wl * 4;
#pragma unroll
      for (int j = 0; j < J; ++j)
        lg[set][j][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rl, (int)loff[j], soff, 0));
    }
  };
  float sum = 0.f;
  coords(0);
  issue_taps();
  issue_logits(0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int rr = 0; rr < RPT; ++rr) {
    float val[J];
    if (rr + 1 < RPT) coords(rr + 1);
#pragma unroll
    for (int j = 0; j < J; ++j) val[j] = tv[j];
    if (rr + 1 < RPT) {
      issue_taps();
      issue_logits(rr + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (rr < nrows) {
      const int soff = rr * w * 4;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int32_t cls = (int32_t)__fmul_rn(val[j], out_scale);
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)cls, roi, (int)coff[j], soff, 0);
        if (rr % LS != 0) continue;      // (compile time: rr is an unrolled loop index)
        // cross entropy of this pixel: logsumexp - logit of the warped class (a class outside 0 .. NC-1 contributes the
        // logsumexp alone, as xt = 0 in ce_partial_kernel)
        const float (&v)[NC] = lg[(rr / LS) & 1][j];
        float m = v[0];
#pragma unroll
        for (int k = 1; k < NC; ++k) m = fmaxf(m, v[k]);
        float se = 0.f, xt = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          se += __expf(v[k] - m);
          xt = (cls == k) ? v[k] : xt;
        }
        const float ce = (m + __logf(se)) - xt;
        sum += (loff[j] != kTapOOB) ? ce : 0.f;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  return sum;
}

template <int J, int RPT, int NC, int LS>
__global__ __launch_bounds__(256) void warpce_kernel(const float* __restrict__ theta, const float* __restrict__ tmpl,
                                                     long tmpl_bstride, int ht, int wt, int h, int w, float rdw, float rdh,
                                                     float out_scale, const float* __restrict__ logits,
                                                     int32_t* __restrict__ out_i, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.z;
  const int c0 = blockIdx.x * (64 * J) + lane;
  const int r0 = (blockIdx.y * 4 + wv) * RPT;
  float sum = 0.f;
  if (r0 < h) {
    float t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] = theta[b * 9 + k];
    bool fin = true;
#pragma unroll
    for (int k = 0; k < 9; ++k) fin &= fabsf(t[k]) <= 0x1p59f;
    const float zs = fabsf(t[6]) + fabsf(t[7]) + fabsf(t[8]);
    const bool live = (fabsf(t[8]) - fabsf(t[6]) - fabsf(t[7])) > 1e-6f * zs + 1e-7f;
#define SFH_WARPCE_GO(LEVEL, SMALL) \
  sum = warpce_body<J, RPT, NC, LS, LEVEL, SMALL>(t, b, lane, c0, r0, tmpl, tmpl_bstride, ht, wt, h, w, rdw, rdh, out_scale, logits, out_i)
    if (fin && w <= 16384 && h <= 16384) {
      if (live) SFH_WARPCE_GO(2, true); else SFH_WARPCE_GO(1, true);
    } else {
      SFH_WARPCE_GO(0, false);
    }
#undef SFH_WARPCE_GO
  }
  // wave partial (fp32 lane sums of at most J * RPT pixels, fp64 across the lanes); waves beyond the frame write zero
  double ds = (double)sum;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ds += __shfl_down(ds, o);
  if (lane == 0) {
    const unsigned nwv = gridDim.x * gridDim.y * 4u;
    partial[(size_t)b * nwv + (blockIdx.y * gridDim.x + blockIdx.x) * 4u + (unsigned)wv] = (float)ds;
  }
}

// one wave per frame: sum of the frame's wave partials in a fixed order, fp64 -> mean cross entropy
__global__ __launch_bounds__(64) void warpce_final_kernel(const float* __restrict__ partial, int nwv, double inv_hw,
                                                          float* __restrict__ score) {
  const int b = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < nwv; i += 64) acc += (double)partial[(size_t)b * nwv + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (threadIdx.x == 0) score[b] = (float)(acc * inv_hw);
}

// Exhaustive arithmetic self-tests (called by tests/, never by the product path): count the inputs on which
// the fast forms above differ from the IEEE divisions they replace.
__global__ void selftest_recip_kernel(unsigned lo, unsigned long long count, unsigned long long* bad) {
  unsigned long long n = 0;
  for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < count;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const float z = __builtin_bit_cast(float, (unsigned)(lo + i));
    const float a = recip_rn<1>(z), c = __fdiv_rn(1.0f, z);
    n += (__builtin_bit_cast(unsigned, a) != __builtin_bit_cast(unsigned, c)) && !(a != a && c != c);
    const float a2 = recip_rn<1>(-z), c2 = __fdiv_rn(1.0f, -z);
    n += (__builtin_bit_cast(unsigned, a2) != __builtin_bit_cast(unsigned, c2)) && !(a2 != a2 && c2 != c2);
  }
  if (n) atomicAdd(bad, n);
}

__global__ void selftest_axis_kernel(int nmax, unsigned long long* bad) {
  // every (i, n) with 2 <= n <= nmax, 0 <= i < n: blockIdx.x + 2 = n
  const int n = blockIdx.x + 2;
  if (n > nmax) return;
  const float rd = __fdiv_rn(1.0f, (float)(n - 1));
  unsigned long long c = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float a = norm_axis2<true>(i, n, rd), r = norm_axis(i, n);   // product rule: SMALL only for n <= 16384
    c += __builtin_bit_cast(unsigned, a) != __builtin_bit_cast(unsigned, r);
  }
  if (c) atomicAdd(bad, c);
}

// inverse(theta) in fp64 (adjugate / determinant), rounded to fp32, then the same pinned
// fp32 point transform as the warp; one thread per point.
__global__ void poi_kernel(const float* __restrict__ theta, const float* __restrict__ poi, int npts,
                           int normalize, float* __restrict__ out, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int b = idx / npts;
  double m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = (double)theta[b * 9 + k];
  const double c00 = m[4] * m[8] - m[5] * m[7];
  const double c01 = m[5] * m[6] - m[3] * m[8];
  const double c02 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  const double id = 1.0 / det;
  Homog Hi;
  Hi.t[0] = (float)(c00 * id);
  Hi.t[1] = (float)((m[2] * m[7] - m[1] * m[8]) * id);
  Hi.t[2] = (float)((m[1] * m[5] - m[2] * m[4]) * id);
  Hi.t[3] = (float)(c01 * id);
  Hi.t[4] = (float)((m[0] * m[8] - m[2] * m[6]) * id);
  Hi.t[5] = (float)((m[2] * m[3] - m[0] * m[5]) * id);
  Hi.t[6] = (float)(c02 * id);
  Hi.t[7] = (float)((m[1] * m[6] - m[0] * m[7]) * id);
  Hi.t[8] = (float)((m[0] * m[4] - m[1] * m[3]) * id);
  float u, v;
  apply_h(Hi, poi[2 * idx], poi[2 * idx + 1], u, v);
  if (normalize) {
    u = __fadd_rn(__fdiv_rn(u, 2.0f), 0.5f);
    v = __fadd_rn(__fdiv_rn(v, 2.0f), 0.5f);
  }
  out[2 * idx] = u;
  out[2 * idx + 1] = v;
}

// ------------------------------------------------------------------ backward wrt theta
// Bilinear warp (training: models/reconstructor.py:185-190): d loss / d theta from d loss / d out.
// Same chain as autograd through Kornia + F.grid_sample: bilinear-tap differences (out-of-range taps
// are 0), d px/d u = wt/2, u = X*s with s = 1/(Z+1e-8) (constant 1 when |Z| <= 1e-8).  One thread per
// 4 output pixels; the 9 sums are reduced per block and added to fp64 accumulators (B,9) with atomics.
__global__ __launch_bounds__(256) void warp_bwd_theta_kernel(const float* __restrict__ theta,
                                                             const float* __restrict__ tmpl, long tmpl_bstride,
                                                             int ht, int wt, int h, int w,
                                                             const float* __restrict__ dout,
                                                             double* __restrict__ acc) {
  const int b = blockIdx.z;
  const int xq = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  Homog H;
#pragma unroll
  for (int k = 0; k < 9; ++k) H.t[k] = theta[b * 9 + k];
  const float* tm = tmpl + (long)b * tmpl_bstride;
  double s9[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) s9[k] = 0.0;
  if (y < h) {
    const float yn = norm_axis(y, h);
    for (int j = 0; j < 4; ++j) {
      const int x = xq + j;
      if (x >= w) break;
      const float xn = norm_axis(x, w);
      const float X = __fadd_rn(__fadd_rn(__fmul_rn(H.t[0], xn), __fmul_rn(H.t[1], yn)), H.t[2]);
      const float Y = __fadd_rn(__fadd_rn(__fmul_rn(H.t[3], xn), __fmul_rn(H.t[4], yn)), H.t[5]);
      const float Z = __fadd_rn(__fadd_rn(__fmul_rn(H.t[6], xn), __fmul_rn(H.t[7], yn)), H.t[8]);
      const bool live = fabsf(Z) > 1e-8f;
      const float s = live ? __fdiv_rn(1.0f, __fadd_rn(Z, 1e-8f)) : 1.0f;
      const float px = unnorm(__fmul_rn(s, X), wt), py = unnorm(__fmul_rn(s, Y), ht);
      const float x0 = floorf(px), y0 = floorf(py);
      const float wx1 = px - x0, wx0 = 1.0f - wx1, wy1 = py - y0, wy0 = 1.0f - wy1;
      const float v00 = fetch(tm, x0, y0, wt, ht), v01 = fetch(tm, x0 + 1.f, y0, wt, ht);
      const float v10 = fetch(tm, x0, y0 + 1.f, wt, ht), v11 = fetch(tm, x0 + 1.f, y0 + 1.f, wt, ht);
      const float g = dout[((long)b * h + y) * w + x];
      const double gu = (double)g * (double)(wy0 * (v01 - v00) + wy1 * (v11 - v10)) * (0.5 * wt);
      const double gv = (double)g * (double)(wx0 * (v10 - v00) + wx1 * (v11 - v01)) * (0.5 * ht);
      const double gX = gu * s, gY = gv * s;
      const double gZ = live ? -(gu * X + gv * Y) * (double)s * (double)s : 0.0;
      s9[0] += gX * xn; s9[1] += gX * yn; s9[2] += gX;
      s9[3] += gY * xn; s9[4] += gY * yn; s9[5] += gY;
      s9[6] += gZ * xn; s9[7] += gZ * yn; s9[8] += gZ;
    }
  }
  __shared__ double sh[256];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    __syncthreads();
    sh[threadIdx.x] = s9[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) unsafeAtomicAdd(&acc[b * 9 + k], sh[0]);
  }
}

// Backward of transform_poi (models/reconstructor.py:120-130) wrt theta: M = inverse(theta),
// p' = hom(M p) / 2 + 0.5 (if normalize); d theta = -M^T (d M) M^T.  One thread per frame, fp64.
__global__ void poi_bwd_theta_kernel(const float* __restrict__ theta, const float* __restrict__ poi, int npts,
                                     int normalize, const float* __restrict__ dout, int batch,
                                     float* __restrict__ dtheta) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  double m[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) m[k] = (double)theta[b * 9 + k];
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
  const double id = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
  double M[9];
  M[0] = (double)(float)(c00 * id);
  M[1] = (double)(float)((m[2] * m[7] - m[1] * m[8]) * id);
  M[2] = (double)(float)((m[1] * m[5] - m[2] * m[4]) * id);
  M[3] = (double)(float)(c01 * id);
  M[4] = (double)(float)((m[0] * m[8] - m[2] * m[6]) * id);
  M[5] = (double)(float)((m[2] * m[3] - m[0] * m[5]) * id);
  M[6] = (double)(float)(c02 * id);
  M[7] = (double)(float)((m[1] * m[6] - m[0] * m[7]) * id);
  M[8] = (double)(float)((m[0] * m[4] - m[1] * m[3]) * id);
  double dM[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const double half = normalize ? 0.5 : 1.0;
  for (int n = 0; n < npts; ++n) {
    const double px = poi[((long)b * npts + n) * 2], py = poi[((long)b * npts + n) * 2 + 1];
    const double X = M[0] * px + M[1] * py + M[2], Y = M[3] * px + M[4] * py + M[5], Z = M[6] * px + M[7] * py + M[8];
    const bool live = fabs(Z) > 1e-8;
    const double s = live ? 1.0 / (Z + 1e-8) : 1.0;
    const double gu = (double)dout[((long)b * npts + n) * 2] * half, gv = (double)dout[((long)b * npts + n) * 2 + 1] * half;
    const double gX = gu * s, gY = gv * s, gZ = live ? -(gu * X + gv * Y) * s * s : 0.0;
    dM[0] += gX * px; dM[1] += gX * py; dM[2] += gX;
    dM[3] += gY * px; dM[4] += gY * py; dM[5] += gY;
    dM[6] += gZ * px; dM[7] += gZ * py; dM[8] += gZ;
  }
  // d theta = -M^T dM M^T
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
      for (int k = 0; k < 3; ++k) a += M[k * 3 + i] * dM[k * 3 + j];  // (M^T dM)[i][j]
      t[i * 3 + j] = a;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
      for (int k = 0; k < 3; ++k) a += t[i * 3 + k] * M[j * 3 + k];   // (t M^T)[i][j]
      dtheta[b * 9 + i * 3 + j] = (float)(-a);
    }
}

}  // namespace

extern "C" int sfh_homography_warp_fwd(const float* theta, const float* tmpl, int64_t tmpl_bstride,
                                       int ht, int wt, int batch, int h, int w, int mode,
                                       float out_scale, float* out_f32, int32_t* out_i32, void* stream) {
  SFH_REQUIRE(theta && tmpl && (out_f32 || out_i32), "homography_warp: null pointer");
  SFH_REQUIRE(batch > 0 && batch <= 65535 && h > 1 && w > 1 && ht > 0 && wt > 0,
              "homography_warp: bad geometry b=%d h=%d w=%d ht=%d wt=%d", batch, h, w, ht, wt);
  SFH_REQUIRE(mode == 0 || mode == 1, "homography_warp: mode %d (0 nearest, 1 bilinear)", mode);
  SFH_REQUIRE(tmpl_bstride == 0 || tmpl_bstride >= (int64_t)ht * wt, "homography_warp: bad template stride");
  SFH_REQUIRE((int64_t)ht * wt <= (1 << 22) && w <= (1 << 20) && h <= (1 << 20),
              "homography_warp: template %dx%d (at most 4 Mi pixels: exact fp32 tap index) or frame %dx%d too large", wt, ht, w, h);
  const float rdw = 1.0f / (float)(w - 1), rdh = 1.0f / (float)(h - 1);   // IEEE single divisions
  // Wave shape: 64*J consecutive pixels x RPT rows.  J = 5 when the row splits into 320-pixel segments
  // (640, 1280, 1920 ...), else 4; bilinear (4 taps and 4 weights per pixel in flight): J = 2.
  // RPT: the largest of 8 / 4 / 2 that still leaves about eight waves per SIMD on 256 CUs.
  const int J = mode == 1 ? 2 : ((w % 320 == 0) ? 5 : 4);
  const long segs = (long)sfh_cdiv(w, 64 * J) * batch;
  int rpt = mode == 1 ? 4 : 8;
  while (rpt > 2 && segs * sfh_cdiv(h, rpt) < SFH_WARP_MIN_WAVES) rpt >>= 1;
  const int out = (out_f32 && out_i32) ? 2 : (out_f32 ? 1 : 0);
  const dim3 grid((unsigned)sfh_cdiv(w, 64 * J), (unsigned)sfh_cdiv(h, 4 * rpt), (unsigned)batch);
#define SFH_WARP_LAUNCH(MODE, JJ, RR, OO)                                                                    \
  hipLaunchKernelGGL((warp2_kernel<MODE, JJ, RR, OO>), grid, dim3(256), 0, (hipStream_t)stream, theta, tmpl, \
                     (long)tmpl_bstride, ht, wt, h, w, rdw, rdh, out_scale, out_f32, out_i32)
#define SFH_WARP_OUT(MODE, JJ, RR)                    \
  do {                                                \
    if (out == 0) SFH_WARP_LAUNCH(MODE, JJ, RR, 0);   \
    else if (out == 1) SFH_WARP_LAUNCH(MODE, JJ, RR, 1); \
    else SFH_WARP_LAUNCH(MODE, JJ, RR, 2);            \
  } while (0)
#define SFH_WARP_RPT(MODE, JJ)                \
  do {                                        \
    if (rpt == 8) SFH_WARP_OUT(MODE, JJ, 8);  \
    else if (rpt == 4) SFH_WARP_OUT(MODE, JJ, 4); \
    else SFH_WARP_OUT(MODE, JJ, 2);           \
  } while (0)
  if (mode == 1) {
    if (rpt == 4) SFH_WARP_OUT(1, 2, 4); else SFH_WARP_OUT(1, 2, 2);
  } else if (J == 5) {
    SFH_WARP_RPT(0, 5);
  } else {
    SFH_WARP_RPT(0, 4);
  }
#undef SFH_WARP_RPT
#undef SFH_WARP_OUT
#undef SFH_WARP_LAUNCH
  return sfh_check_launch("warp_kernel");
}

extern "C" int64_t sfh_warp_consistency_workspace_floats(int batch, int h, int w) {
  if (batch <= 0 || h <= 1 || w <= 1) return -1;
  const int J = (w % 320 == 0) ? 5 : 4;
  // one float per wave; the smallest rows-per-thread the launcher may choose gives the most waves
  return (int64_t)batch * sfh_cdiv(w, 64 * J) * sfh_cdiv(h, 4 * 2) * 4;
}

extern "C" int sfh_warp_consistency_fwd(const float* theta, const float* tmpl, int64_t tmpl_bstride, int ht, int wt,
                                        int batch, int h, int w, float out_scale, const float* logits, int nc, int hl, int wl,
                                        int32_t* out_i32, float* partial, float* score, void* stream) {
  SFH_REQUIRE(theta && tmpl && logits && out_i32 && partial && score, "warp_consistency: null pointer");
  SFH_REQUIRE(batch > 0 && batch <= 65535 && h > 1 && w > 1 && ht > 0 && wt > 0,
              "warp_consistency: bad geometry b=%d h=%d w=%d ht=%d wt=%d", batch, h, w, ht, wt);
  SFH_REQUIRE(nc == 4, "warp_consistency: nc=%d (the fused kernel is built for 4 classes; use sfh_homography_warp_fwd + "
              "sfh_consistency_ce_fwd otherwise)", nc);
  SFH_REQUIRE((hl == h && wl == w) || (2 * hl == h && 2 * wl == w),
              "warp_consistency: logits %dx%d against a warp of %dx%d (the same size, or exactly half of it in both directions)", wl, hl, w, h);
  const int ls = h / hl;
  SFH_REQUIRE(tmpl_bstride == 0 || tmpl_bstride >= (int64_t)ht * wt, "warp_consistency: bad template stride");
  SFH_REQUIRE((int64_t)ht * wt <= (1 << 22) && w <= (1 << 20) && h <= (1 << 20) && (int64_t)nc * h * w * 4 < 0x7FFFFFF0LL,
              "warp_consistency: template %dx%d or frame %dx%d too large", wt, ht, w, h);
  const float rdw = 1.0f / (float)(w - 1), rdh = 1.0f / (float)(h - 1);
  const int J = (w % 320 == 0) ? 5 : 4;
  const long segs = (long)sfh_cdiv(w, 64 * J) * batch;
  int rpt = 8;
  while (rpt > 2 && segs * sfh_cdiv(h, rpt) < SFH_WARP_MIN_WAVES) rpt >>= 1;
  const dim3 grid((unsigned)sfh_cdiv(w, 64 * J), (unsigned)sfh_cdiv(h, 4 * rpt), (unsigned)batch);
#define SFH_WCE(JJ, RR)                                                                                             \
  do {                                                                                                              \
    if (ls == 1)                                                                                                    \
      hipLaunchKernelGGL((warpce_kernel<JJ, RR, 4, 1>), grid, dim3(256), 0, (hipStream_t)stream, theta, tmpl,         \
                         (long)tmpl_bstride, ht, wt, h, w, rdw, rdh, out_scale, logits, out_i32, partial);           \
    else                                                                                                            \
      hipLaunchKernelGGL((warpce_kernel<JJ, RR, 4, 2>), grid, dim3(256), 0, (hipStream_t)stream, theta, tmpl,         \
                         (long)tmpl_bstride, ht, wt, h, w, rdw, rdh, out_scale, logits, out_i32, partial);           \
  } while (0)
  if (J == 5) {
    if (rpt == 8) SFH_WCE(5, 8); else if (rpt == 4) SFH_WCE(5, 4); else SFH_WCE(5, 2);
  } else {
    if (rpt == 8) SFH_WCE(4, 8); else if (rpt == 4) SFH_WCE(4, 4); else SFH_WCE(4, 2);
  }
#undef SFH_WCE
  int rc = sfh_check_launch("warpce_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(warpce_final_kernel, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)stream, partial,
                     (int)(grid.x * grid.y * 4u), 1.0 / ((double)hl * (double)wl), score);
  return sfh_check_launch("warpce_final_kernel");
}

extern "C" int sfh_selftest_warp_arith(int64_t* mismatches, void* stream) {
  // exhaustive: every float z with 2^-64 <= |z| <= 2^64 for the reciprocal; every (i, n), n <= 16385 for the axis.
  // mismatches: DEVICE pointer to two int64 counters (zeroed here).
  SFH_REQUIRE(mismatches, "selftest_warp_arith: null pointer");
  if (hipMemsetAsync(mismatches, 0, 16, (hipStream_t)stream) != hipSuccess) return sfh_check_launch("memset");
  const unsigned lo = __builtin_bit_cast(unsigned, 0x1p-64f), hi = __builtin_bit_cast(unsigned, 0x1p64f);
  hipLaunchKernelGGL(selftest_recip_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, lo,
                     (unsigned long long)(hi - lo) + 1, (unsigned long long*)mismatches);
  hipLaunchKernelGGL(selftest_axis_kernel, dim3(16384), dim3(256), 0, (hipStream_t)stream, 16385,
                     (unsigned long long*)mismatches + 1);
  return sfh_check_launch("selftest_warp_arith");
}

extern "C" int sfh_poi_project_fwd(const float* theta, const float* poi, int batch, int npts,
                                   int normalize, float* out, void* stream) {
  SFH_REQUIRE(theta && poi && out, "poi_project: null pointer");
  SFH_REQUIRE(batch > 0 && npts > 0, "poi_project: bad geometry");
  const int total = batch * npts;
  hipLaunchKernelGGL(poi_kernel, dim3((unsigned)sfh_cdiv(total, 128)), dim3(128), 0, (hipStream_t)stream,
                     theta, poi, npts, normalize, out, total);
  return sfh_check_launch("poi_kernel");
}

extern "C" int sfh_homography_warp_bwd_theta(const float* theta, const float* tmpl, int64_t tmpl_bstride,
                                             int ht, int wt, int batch, int h, int w, const float* dout,
                                             double* acc, void* stream) {
  SFH_REQUIRE(theta && tmpl && dout && acc, "homography_warp_bwd: null pointer");
  SFH_REQUIRE(batch > 0 && batch <= 65535 && h > 1 && w > 1 && ht > 0 && wt > 0, "homography_warp_bwd: bad geometry");
  SFH_REQUIRE(tmpl_bstride == 0 || tmpl_bstride >= (int64_t)ht * wt, "homography_warp_bwd: bad template stride");
  const dim3 grid((unsigned)sfh_cdiv(w, 256), (unsigned)sfh_cdiv(h, 4), (unsigned)batch);
  hipLaunchKernelGGL(warp_bwd_theta_kernel, grid, dim3(256), 0, (hipStream_t)stream, theta, tmpl,
                     (long)tmpl_bstride, ht, wt, h, w, dout, acc);
  return sfh_check_launch("warp_bwd_theta_kernel");
}

extern "C" int sfh_poi_project_bwd_theta(const float* theta, const float* poi, int batch, int npts,
                                         int normalize, const float* dout, float* dtheta, void* stream) {
  SFH_REQUIRE(theta && poi && dout && dtheta && batch > 0 && npts > 0, "poi_project_bwd: bad argument");
  hipLaunchKernelGGL(poi_bwd_theta_kernel, dim3((unsigned)sfh_cdiv(batch, 64)), dim3(64), 0, (hipStream_t)stream,
                     theta, poi, npts, normalize, dout, batch, dtheta);
  return sfh_check_launch("poi_bwd_theta_kernel");
}
