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
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

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
  // normalised x of the block's 256 columns: one IEEE division per thread instead of four (the four rows
  // of the block share them); same arithmetic per column, so the result is unchanged
  __shared__ float xn_s[256];
  {
    const int xc = blockIdx.x * 256 + threadIdx.x;
    xn_s[threadIdx.x] = norm_axis(xc < w ? xc : w - 1, w);
  }
  __syncthreads();
  const int b = blockIdx.z;
  const int lx = (threadIdx.x & 63) * 4;
  const int xq = blockIdx.x * 256 + lx;  // first of 4 pixels
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
    apply_h(H, xn_s[lx + j], yn, u, v);
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
