// warp.hip - STN homography warp of the court template and POI projection.
//
// Replaces, for Reconstructor.warp / transform_poi (models/reconstructor.py:109-130), the
// Kornia chain create_meshgrid -> transform_points -> convert_points_from_homogeneous ->
// F.grid_sample(padding 'zeros', align_corners=False), fused with predict()'s
// `* mask_classes` and `.type(torch.int32)` (models/reconstructor.py:223,240).
//
// HBM-bound: per batch the compulsory traffic is B*h*w*4 B of output + one template image
// (the template is one image replicated over the batch: utils/dataset.py:59) + 36 B of
// theta per frame.  The coordinate arithmetic uses individually rounded fp32 operations
// (__fmul_rn/__fadd_rn/__fdiv_rn: no FMA contraction) in exactly the order of
// oracle/warp_ref.py, so nearest-mode results are integer-identical to the oracle.
//
// Work decomposition: one thread = 4 consecutive output pixels of one row (one 16-byte
// store per output tensor); a 256-thread block covers a 4-row x 256-col strip, so all taps
// of a block fall into a few template rows (L1/L2-resident; the template is <= 3.7 MB).
#include "common.h"

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

template <int MODE>
__device__ __forceinline__ float sample_one(const float* __restrict__ tm, float u, float v, int wt, int ht) {
  const float px = unnorm(u, wt), py = unnorm(v, ht);
  if (MODE == 0) return fetch(tm, rintf(px), rintf(py), wt, ht);  // round-half-to-even
  const float x0 = floorf(px), y0 = floorf(py);
  const float wx1 = __fsub_rn(px, x0), wx0 = __fsub_rn(1.0f, wx1);
  const float wy1 = __fsub_rn(py, y0), wy0 = __fsub_rn(1.0f, wy1);
  float r = __fmul_rn(fetch(tm, x0, y0, wt, ht), __fmul_rn(wy0, wx0));
  r = __fadd_rn(r, __fmul_rn(fetch(tm, x0 + 1.f, y0, wt, ht), __fmul_rn(wy0, wx1)));
  r = __fadd_rn(r, __fmul_rn(fetch(tm, x0, y0 + 1.f, wt, ht), __fmul_rn(wy1, wx0)));
  r = __fadd_rn(r, __fmul_rn(fetch(tm, x0 + 1.f, y0 + 1.f, wt, ht), __fmul_rn(wy1, wx1)));
  return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ theta,
                                                   const float* __restrict__ tmpl, long tmpl_bstride,
                                                   int ht, int wt, int h, int w, float out_scale,
                                                   float* __restrict__ out_f, int32_t* __restrict__ out_i) {
  const int b = blockIdx.z;
  const int xq = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;  // first of 4 pixels
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (y >= h || xq >= w) return;
  Homog H;
#pragma unroll
  for (int k = 0; k < 9; ++k) H.t[k] = theta[b * 9 + k];  // wave-uniform -> scalar loads
  const float* tm = tmpl + (long)b * tmpl_bstride;
  const float yn = norm_axis(y, h);
  float val[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float u, v;
    apply_h(H, norm_axis(xq + j, w), yn, u, v);
    val[j] = (xq + j < w) ? sample_one<MODE>(tm, u, v, wt, ht) : 0.f;
  }
  const long o = ((long)b * h + y) * w + xq;
  const bool vec = (xq + 3 < w) && ((w & 3) == 0);
  if (out_f) {
    if (vec) {
      *reinterpret_cast<f32x4*>(out_f + o) = (f32x4){val[0], val[1], val[2], val[3]};
    } else {
      for (int j = 0; j < 4 && xq + j < w; ++j) out_f[o + j] = val[j];
    }
  }
  if (out_i) {
    int32_t iv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) iv[j] = (int32_t)__fmul_rn(val[j], out_scale);  // trunc, like .type(int32)
    if (vec) {
      *reinterpret_cast<int4*>(out_i + o) = make_int4(iv[0], iv[1], iv[2], iv[3]);
    } else {
      for (int j = 0; j < 4 && xq + j < w; ++j) out_i[o + j] = iv[j];
    }
  }
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

}  // namespace

extern "C" int sfh_homography_warp_fwd(const float* theta, const float* tmpl, int64_t tmpl_bstride,
                                       int ht, int wt, int batch, int h, int w, int mode,
                                       float out_scale, float* out_f32, int32_t* out_i32, void* stream) {
  SFH_REQUIRE(theta && tmpl && (out_f32 || out_i32), "homography_warp: null pointer");
  SFH_REQUIRE(batch > 0 && batch <= 65535 && h > 1 && w > 1 && ht > 0 && wt > 0,
              "homography_warp: bad geometry b=%d h=%d w=%d ht=%d wt=%d", batch, h, w, ht, wt);
  SFH_REQUIRE(mode == 0 || mode == 1, "homography_warp: mode %d (0 nearest, 1 bilinear)", mode);
  SFH_REQUIRE(tmpl_bstride == 0 || tmpl_bstride >= (int64_t)ht * wt, "homography_warp: bad template stride");
  const dim3 grid((unsigned)sfh_cdiv(w, 256), (unsigned)sfh_cdiv(h, 4), (unsigned)batch);
  if (mode == 0)
    hipLaunchKernelGGL(warp_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, theta, tmpl,
                       (long)tmpl_bstride, ht, wt, h, w, out_scale, out_f32, out_i32);
  else
    hipLaunchKernelGGL(warp_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, theta, tmpl,
                       (long)tmpl_bstride, ht, wt, h, w, out_scale, out_f32, out_i32);
  return sfh_check_launch("warp_kernel");
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
