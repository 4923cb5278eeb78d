// Probe for the two-plane fp16 operand format ("f16x3"): (1) does v_mfma_f32_16x16x32_f16 honour fp16
// subnormal inputs, (2) its issue rate next to the bf16 instruction in the conv_s3 step shape
// (2 operand reads + 6 MFMAs per step instead of 3 + 12).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f16_probe mfma_f16_probe.hip && ./mfma_f16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 hf(const u32x4& v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ bf16x8 bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// A[i][k] = a for k == 0 else 0; B[k][j] = b for k == 0: D[i][j] = a*b
__global__ void denorm_probe(const unsigned short* ab, float* out) {
  const int lane = threadIdx.x;
  f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  if (lane < 16) {
    a[0] = __builtin_bit_cast(_Float16, ab[0]);
    b[0] = __builtin_bit_cast(_Float16, ab[1]);
  }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  if (lane == 0) out[0] = acc[0];
  // conversion: does v_cvt_f16_f32 produce subnormals?
  if (lane == 0) {
    const float tiny = 3.0e-6f;
    const _Float16 h = (_Float16)tiny;
    out[1] = (float)h;
    out[2] = (float)__builtin_bit_cast(unsigned short, h);
  }
}

template <int F16, int NP>
__global__ __launch_bounds__(256, 2) void rate(float* out, int iters, const unsigned* seed) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  u32x4* lds = (u32x4*)sm;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) lds[i] = (u32x4){seed[i & 255], seed[(i + 1) & 255], seed[(i + 2) & 255], seed[(i + 3) & 255]};
  __syncthreads();
  f32x4 acc[2][8];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
  u32x4 w[NP][2], x[NP];
  for (int p = 0; p < NP; ++p) { x[p] = lds[lane + 64 * p]; for (int n = 0; n < 2; ++n) w[p][n] = lds[1024 + lane + 64 * (2 * p + n)]; }
  constexpr int NPROD = NP == 3 ? 6 : 3;
  constexpr int PW3[6] = {0, 1, 2, 0, 1, 0}, PX3[6] = {2, 1, 0, 1, 0, 0};
  constexpr int PW2[3] = {0, 1, 0}, PX2[3] = {1, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
        for (int p = 0; p < NP; ++p) x[p] = lds[lane + 64 * p + mi * 16 + t * 128 + (it & 1)];
#pragma unroll
        for (int k = 0; k < NPROD; ++k)
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const int pw = NP == 3 ? PW3[k] : PW2[k], px = NP == 3 ? PX3[k] : PX2[k];
            if (F16)
              acc[n][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(w[pw][n]), hf(x[px]), acc[n][mi], 0, 0, 0);
            else
              acc[n][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(w[pw][n]), bf(x[px]), acc[n][mi], 0, 0, 0);
          }
      }
    }
  }
  f32x4 s = {0, 0, 0, 0};
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 8; ++b) s += acc[a][b];
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

static unsigned short f2h(float f) {  // host float -> fp16 bits, round to nearest even, with subnormals
  _Float16 h = (_Float16)f;
  unsigned short u;
  __builtin_memcpy(&u, &h, 2);
  return u;
}

int main() {
  float* out;
  unsigned short* ab;
  hipMalloc(&out, 1 << 22);
  hipMalloc(&ab, 16);
  const float as[4] = {ldexpf(1.f, -20), ldexpf(1.5f, -16), ldexpf(1.f, -24), 1.0f};
  for (int i = 0; i < 4; ++i) {
    unsigned short h[2] = {f2h(as[i]), f2h(1024.f)};
    hipMemcpy(ab, h, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(denorm_probe, dim3(1), dim3(64), 0, 0, ab, out);
    float r[3];
    hipMemcpy(r, out, 12, hipMemcpyDeviceToHost);
    printf("a=%g (bits %04x) x 1024 -> mfma %g (expected %g)%s | cvt(3e-6)=%g bits %g\n", as[i], h[0], r[0], as[i] * 1024.f,
           r[0] == as[i] * 1024.f ? "" : "  ** FLUSHED/ROUNDED **", r[1], r[2]);
  }
  // both operands subnormal-free check done; now the rate
  unsigned hs[256];
  unsigned* seed;
  hipMalloc(&seed, 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  // occupancy: how much of the matrix pipe does ONE wave per SIMD fill in this step shape (2 reads + 6 MFMAs)?
  {
    for (int i = 0; i < 256; ++i) { unsigned r = (unsigned)rand(); hs[i] = (0x3800u | (r & 0x3FF)) | ((0x3800u | ((r >> 10) & 0x3FF)) << 16); }
    hipMemcpy(seed, hs, 1024, hipMemcpyHostToDevice);
    for (int wg = 1; wg <= 3; ++wg) {
      const int iters = 2000, grid = 256 * wg;   // one round: wg workgroups of 4 waves per CU
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rate<1, 2>), dim3(grid), dim3(256), wg == 3 ? 50000 : 65536, 0, out, iters, seed);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const double fl = (double)grid * 4 * iters * 72 * 3 * 2 * 2.0 * 16 * 16 * 32;
      printf("f16 x3, %d workgroup(s) per CU (%d wave(s) per SIMD): %.3f ms  %.1f TFLOP/s of MFMA work\n", wg, wg, best, fl / best * 1e-9);
    }
  }
  for (int dat = 0; dat < 2; ++dat) {
    for (int i = 0; i < 256; ++i) {
      // two 16-bit values per word: dat 0 = small-exponent random mantissas, dat 1 = zeros
      unsigned r = (unsigned)rand();
      unsigned lo = 0x3800u | (r & 0x3FF), hi = 0x3800u | ((r >> 10) & 0x3FF);
      hs[i] = dat ? 0u : (lo | (hi << 16));
    }
    hipMemcpy(seed, hs, 1024, hipMemcpyHostToDevice);
    const int iters = 400, grid = 512 * 8;
    for (int var = 0; var < 3; ++var) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (var == 0) hipLaunchKernelGGL((rate<0, 3>), dim3(grid), dim3(256), 65536, 0, out, iters, seed);
        if (var == 1) hipLaunchKernelGGL((rate<1, 2>), dim3(grid), dim3(256), 65536, 0, out, iters, seed);
        if (var == 2) hipLaunchKernelGGL((rate<0, 2>), dim3(grid), dim3(256), 65536, 0, out, iters, seed);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
      }
      const int nprod = var == 0 ? 6 : 3;
      const double mf = (double)grid * 4 * iters * 72 * nprod * 2;  // MFMAs
      const double fl = mf * 2.0 * 16 * 16 * 32;
      printf("data %s  %s: %.3f ms  %.1f TFLOP/s of MFMA work = %.1f TFLOP/s fp32-equivalent\n", dat ? "zeros " : "random",
             var == 0 ? "bf16 x6 (3 reads + 12 mfma / step)" : var == 1 ? "f16  x3 (2 reads +  6 mfma / step)" : "bf16 x3 (2 reads +  6 mfma / step)",
             best, fl / best * 1e-9, fl / nprod / best * 1e-9);
    }
  }
  return 0;
}
