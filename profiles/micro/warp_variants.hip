// Micro-benchmark of the homography warp kernel variants (csrc/warp.hip) outside torch:
// exactness of every variant against the first-generation kernel, time per launch with HIP events,
// plus a store-only kernel of the same shape (the write roofline this path can reach at best).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize profiles/micro/warp_variants.hip \
//        sports-field-homography_amd/csrc/capi.hip -o profiles/micro/warp_variants
#include "../../sports-field-homography_amd/csrc/warp.hip"

namespace {
// ---- first-generation kernel (round 1), kept here as the A/B baseline and exactness reference
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

}  // namespace

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void store_only(int32_t* out, long n) {
  long i = (long)blockIdx.x * 256 * 8 + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 8; ++k, i += 256)
    if (i < n) out[i] = (int32_t)(i & 3);
}

static const float REAL[2][9] = {
    {8.030766487121582f, -0.22687992453575134f, 9.891857147216797f, 3.553352117538452f, 25.72734260559082f,
     -0.09768841415643692f, 0.1463453769683838f, 5.179210662841797f, 16.56546974182129f},
    {5.78266048f, -0.43701401f, 8.0031395f, 3.63819695f, 15.77359295f, -0.46604609f, 0.14406031f, 3.68673325f, 13.25017166f}};

static bool g_pmc = false;
template <typename F>
static float timeit(F f, int iters) {
  if (g_pmc) { for (int i = 0; i < 3; ++i) f(); CK(hipDeviceSynchronize()); return 1.0f; }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  {  // hold the device busy ~40 ms first so that the clocks have ramped before the timed loop
    CK(hipEventRecord(e0, 0)); f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float one; CK(hipEventElapsedTime(&one, e0, e1));
    const int nw = (int)(40.0f / (one > 1e-3f ? one : 1e-3f)) + 1;
    for (int i = 0; i < (nw < 4000 ? nw : 4000); ++i) f();
    if (iters * one < 20.0f) iters = (int)(20.0f / one) + 1;
    if (iters > 4000) iters = 4000;
  }
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

template <int MODE, int J, int RPT>
static void launch2(const float* th, const float* tm, int ht, int wt, int B, int h, int w, float sc, float* of, int32_t* oi) {
  dim3 grid((w + 64 * J - 1) / (64 * J), (h + 4 * RPT - 1) / (4 * RPT), B);
  if (of && oi) hipLaunchKernelGGL((warp2_kernel<MODE, J, RPT, 2>), grid, dim3(256), 0, 0, th, tm, 0L, ht, wt, h, w, 1.0f / (float)(w - 1), 1.0f / (float)(h - 1), sc, of, oi);
  else if (of) hipLaunchKernelGGL((warp2_kernel<MODE, J, RPT, 1>), grid, dim3(256), 0, 0, th, tm, 0L, ht, wt, h, w, 1.0f / (float)(w - 1), 1.0f / (float)(h - 1), sc, of, oi);
  else hipLaunchKernelGGL((warp2_kernel<MODE, J, RPT, 0>), grid, dim3(256), 0, 0, th, tm, 0L, ht, wt, h, w, 1.0f / (float)(w - 1), 1.0f / (float)(h - 1), sc, of, oi);
}

template <int J, int RPT>
static void launch_noload(const float* th, const float* tm, int ht, int wt, int B, int h, int w, float sc, int32_t* oi) {
  dim3 grid((w + 64 * J - 1) / (64 * J), (h + 4 * RPT - 1) / (4 * RPT), B);
  hipLaunchKernelGGL((warp2_kernel<0, J, RPT, 0, true>), grid, dim3(256), 0, 0, th, tm, 0L, ht, wt, h, w, 1.0f / (float)(w - 1), 1.0f / (float)(h - 1), sc, (float*)nullptr, oi);
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 30;
  const bool pmc = argc > 2 && !strcmp(argv[2], "pmc");   // one configuration, few launches: for rocprofv3 --pmc
  g_pmc = pmc;
  // exhaustive arithmetic checks first
  unsigned long long* bad; CK(hipMalloc(&bad, 16)); CK(hipMemset(bad, 0, 16));
  {
    const unsigned lo = __builtin_bit_cast(unsigned, 0x1p-64f), hi = __builtin_bit_cast(unsigned, 0x1p64f);
    hipLaunchKernelGGL(selftest_recip_kernel, dim3(4096), dim3(256), 0, 0, lo, (unsigned long long)(hi - lo) + 1, bad);
    hipLaunchKernelGGL(selftest_axis_kernel, dim3(16384), dim3(256), 0, 0, 16385, bad + 1);
    unsigned long long hb[2]; CK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost));
    printf("{\"selftest_recip_mismatches\": %llu, \"values\": %llu, \"selftest_axis_mismatches\": %llu}\n", hb[0],
           2ull * ((unsigned long long)(hi - lo) + 1), hb[1]);
  }
  const int sizes[2][2] = {{640, 360}, {1280, 720}};
  const int batches[3] = {16, 128, 1024};
  for (int si = pmc ? 1 : 0; si < 2; ++si) {
    const int w = sizes[si][0], h = sizes[si][1], wt = w, ht = h;
    std::vector<float> tmpl((size_t)ht * wt);
    for (int y = 0; y < ht; ++y) for (int x = 0; x < wt; ++x) tmpl[(size_t)y * wt + x] = (float)(((x / 37) + (y / 23)) & 3) * 0.25f;
    float* dtm; CK(hipMalloc(&dtm, tmpl.size() * 4)); CK(hipMemcpy(dtm, tmpl.data(), tmpl.size() * 4, hipMemcpyHostToDevice));
    for (int bi = pmc ? 1 : 0; bi < (pmc ? 2 : 3); ++bi) {
      const int B = batches[bi];
      const size_t npx = (size_t)B * h * w;
      std::vector<float> th((size_t)B * 9);
      for (int b = 0; b < B; ++b) {
        srand(b);
        const int k = b & 1;
        for (int q = 0; q < 9; ++q) th[b * 9 + q] = REAL[k][q] + 1e-3f * ((rand() % 2001) - 1000) / 500.f;
        if (b % 7 == 3) { const float id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; for (int q = 0; q < 9; ++q) th[b * 9 + q] = id[q] + 0.02f * ((rand() % 2001) - 1000) / 1000.f; }
      }
      float* dth; CK(hipMalloc(&dth, th.size() * 4)); CK(hipMemcpy(dth, th.data(), th.size() * 4, hipMemcpyHostToDevice));
      int32_t *o1, *o2; float *f1, *f2;
      CK(hipMalloc(&o1, npx * 4)); CK(hipMalloc(&o2, npx * 4)); CK(hipMalloc(&f1, npx * 4)); CK(hipMalloc(&f2, npx * 4));
      const double bytes = (double)npx * 4 + (double)ht * wt * 4 + B * 36.0;
      auto report = [&](const char* name, float us, long mism) {
        printf("{\"size\": \"%dx%d\", \"batch\": %d, \"kernel\": \"%s\", \"us\": %.2f, \"GB/s\": %.1f, \"frac_8TB/s\": %.4f, \"mismatches_vs_v1\": %ld}\n",
               w, h, B, name, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0, mism);
        fflush(stdout);
      };
      auto cmp_i = [&]() { std::vector<int32_t> a(npx), c(npx); CK(hipMemcpy(a.data(), o1, npx * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), o2, npx * 4, hipMemcpyDeviceToHost)); long m = 0; for (size_t i = 0; i < npx; ++i) m += a[i] != c[i]; return m; };
      auto cmp_f = [&]() { std::vector<float> a(npx), c(npx); CK(hipMemcpy(a.data(), f1, npx * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), f2, npx * 4, hipMemcpyDeviceToHost)); long m = 0; for (size_t i = 0; i < npx; ++i) m += memcmp(&a[i], &c[i], 4) != 0; return m; };
      const bool check = B <= 128;
      // ---- nearest -> int32
      {
        dim3 g1((w + 255) / 256, (h + 3) / 4, B);
        auto v1 = [&]() { hipLaunchKernelGGL(warp_kernel<0>, g1, dim3(256), 0, 0, dth, dtm, 0L, ht, wt, h, w, 4.0f, (float*)nullptr, o1); };
        report("v1 nearest", timeit(v1, iters), 0);
        CK(hipMemset(o2, 0xff, npx * 4));
#define RUN2(J, R) { auto f = [&]() { launch2<0, J, R>(dth, dtm, ht, wt, B, h, w, 4.0f, nullptr, o2); }; float us = timeit(f, iters); report("v2 nearest J" #J " RPT" #R, us, check ? cmp_i() : -1); }
        RUN2(5, 2) RUN2(5, 4) RUN2(5, 8) RUN2(4, 4) RUN2(2, 8) RUN2(10, 4)
        { auto f = [&]() { launch_noload<5, 4>(dth, dtm, ht, wt, B, h, w, 4.0f, o2); }; report("v2 nearest J5 RPT4 NOLOAD", timeit(f, iters), -1); }
#undef RUN2
        auto so = [&]() { hipLaunchKernelGGL(store_only, dim3((unsigned)((npx + 2047) / 2048)), dim3(256), 0, 0, o2, (long)npx); };
        report("store-only", timeit(so, iters), -1);
      }
      // ---- bilinear -> f32
      {
        dim3 g1((w + 255) / 256, (h + 3) / 4, B);
        auto v1 = [&]() { hipLaunchKernelGGL(warp_kernel<1>, g1, dim3(256), 0, 0, dth, dtm, 0L, ht, wt, h, w, 1.0f, f1, (int32_t*)nullptr); };
        report("v1 bilinear", timeit(v1, iters), 0);
#define RUN2(J, R) { auto f = [&]() { launch2<1, J, R>(dth, dtm, ht, wt, B, h, w, 1.0f, f2, nullptr); }; float us = timeit(f, iters); report("v2 bilinear J" #J " RPT" #R, us, check ? cmp_f() : -1); }
        RUN2(5, 2) RUN2(5, 4) RUN2(2, 4) RUN2(2, 8)
#undef RUN2
      }
      CK(hipFree(dth)); CK(hipFree(o1)); CK(hipFree(o2)); CK(hipFree(f1)); CK(hipFree(f2));
    }
    CK(hipFree(dtm));
  }
  return 0;
}
