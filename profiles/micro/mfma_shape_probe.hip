// Which fp16 MFMA shape sustains more under the chip's power limit: v_mfma_f32_16x16x32_f16 (what conv_s3_kernel
// issues) or v_mfma_f32_32x32x16_f16 (half the operand-register reads per FLOP)?  Register-resident operands only
// (the upper bound of either shape), random mantissas, every CU busy for ~10 ms; reports TFLOP/s of MFMA work and
// the clock the chip held inside the kernel (s_memtime / s_memrealtime, as profiles/r03_clock_h2.txt).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip && ./mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 hf(const u32x4& v) { return __builtin_bit_cast(f16x8, v); }

// SHAPE 0: 16x16x32, 16 independent accumulators (64 VGPRs); SHAPE 1: 32x32x16, 4 independent accumulators (64 VGPRs)
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void rate(float* out, int iters, const unsigned* seed, unsigned long long* clk) {
  const int tid = threadIdx.x, lane = tid & 63;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    const unsigned s0 = seed[(lane * 7 + i * 13) & 255], s1 = seed[(lane * 11 + i * 5 + 1) & 255];
    a[i] = (u32x4){s0, s1, s0 ^ 0x00550055u, s1 ^ 0x00330033u};
    b[i] = (u32x4){s1, s0 ^ 0x000F000Fu, s1 ^ 0x00770077u, s0};
  }
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  float res = 0.f;
  if (SHAPE == 0) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(a[(i + k) & 3]), hf(b[(i >> 2) ^ k]), acc[i], 0, 0, 0);
    }
    f32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) s += acc[i];
    res = s[0] + s[1] + s[2] + s[3];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k)   // 8 x 4 MFMAs of 32K FLOP = the 64 x 16K FLOP of the other shape
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hf(a[(i + k) & 3]), hf(b[(i ^ k) & 3]), acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j) res += acc[i][j];
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  out[blockIdx.x * 256 + tid] = res;
  if (lane == 0 && (blockIdx.x & 31) == 0) {
    atomicAdd(&clk[0], c1 - c0);
    atomicAdd(&clk[1], r1 - r0);
  }
}

// The conv_s3 step shape: per step two 16-byte LDS reads per lane (the two planes of 16 pixels x 32 k) feed 3 * NI MFMAs
// (NI = cout groups of 16 per wave: 2 today, 4 = a wave covering 64 couts); weights stay in registers here.
template <int NI>
__global__ __launch_bounds__(256, 2) void conv_shape(float* out, int iters, const unsigned* seed, unsigned long long* clk) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  u32x4* lds = (u32x4*)sm;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2816; i += 256) lds[i] = (u32x4){seed[i & 255], seed[(i + 1) & 255], seed[(i + 2) & 255], seed[(i + 3) & 255]};
  __syncthreads();
  constexpr int MT = NI == 2 ? 8 : 4;   // the same 64 accumulator registers either way
  f32x4 acc[NI][MT];
  for (int a = 0; a < NI; ++a) for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
  u32x4 w[2][NI], x[2];
  for (int p = 0; p < 2; ++p) for (int n = 0; n < NI; ++n) w[p][n] = lds[1024 + lane + 64 * (NI * p + n)];
  constexpr int PW2[3] = {0, 1, 0}, PX2[3] = {1, 0, 0};
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
        for (int p = 0; p < 2; ++p) x[p] = lds[lane + 1408 * p + mi * 16 + t * 34 + (it & 1)];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int n = 0; n < NI; ++n)
            acc[n][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(w[PW2[k]][n]), hf(x[PX2[k]]), acc[n][mi], 0, 0, 0);
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  f32x4 s = {0, 0, 0, 0};
  for (int a = 0; a < NI; ++a) for (int b = 0; b < MT; ++b) s += acc[a][b];
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0 && (blockIdx.x & 31) == 0) {
    atomicAdd(&clk[0], c1 - c0);
    atomicAdd(&clk[1], r1 - r0);
  }
}

int main() {
  float* out;
  unsigned* seed;
  unsigned long long* clk;
  hipMalloc(&out, 1 << 24);
  hipMalloc(&seed, 1024);
  hipMalloc(&clk, 16);
  unsigned hs[256];
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int dat = 0; dat < 2; ++dat) {
    for (int i = 0; i < 256; ++i) {
      unsigned r = (unsigned)rand();
      hs[i] = dat ? 0u : ((0x3800u | (r & 0x3FF)) | ((0x3800u | ((r >> 10) & 0x3FF)) << 16));
    }
    hipMemcpy(seed, hs, 1024, hipMemcpyHostToDevice);
    for (int wg = 1; wg <= 2; ++wg)
      for (int shape = 0; shape < 2; ++shape) {
        const int grid = 256 * wg * 4, iters = 6000 / wg;   // four rounds of wg workgroups per CU, ~10 ms
        float best = 1e9f;
        unsigned long long hc[2] = {0, 0};
        for (int rep = 0; rep < 4; ++rep) {
          hipMemset(clk, 0, 16);
          hipEventRecord(e0);
          if (shape == 0) hipLaunchKernelGGL((rate<0>), dim3(grid), dim3(256), 0, 0, out, iters, seed, clk);
          else hipLaunchKernelGGL((rate<1>), dim3(grid), dim3(256), 0, 0, out, iters, seed, clk);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (rep && ms < best) { best = ms; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost); }
        }
        const double fl = (double)grid * 4 * iters * 64 * 2.0 * 16 * 16 * 32;
        printf("data %s  %s  %d workgroup(s)/CU: %7.3f ms  %7.1f TFLOP/s of fp16 MFMA work  in-kernel clock %.2f GHz\n",
               dat ? "zeros " : "random", shape ? "32x32x16" : "16x16x32", wg, best, fl / best * 1e-9,
               hc[1] ? 0.1 * (double)hc[0] / (double)hc[1] : 0.0);
      }
  }
  // the conv step shape with its LDS operand reads: 2 reads per 6 MFMAs (today) against 2 reads per 12
  for (int i = 0; i < 256; ++i) {
    unsigned r = (unsigned)rand();
    hs[i] = (0x3800u | (r & 0x3FF)) | ((0x3800u | ((r >> 10) & 0x3FF)) << 16);
  }
  hipMemcpy(seed, hs, 1024, hipMemcpyHostToDevice);
  for (int wg = 2; wg <= 3; ++wg)
    for (int ni = 2; ni <= 4; ni += 2) {
      const int grid = 256 * wg * 4;
      const int iters = (ni == 2 ? 600 : 1200) / wg * 2;   // per iteration 9 * MT * 3 * NI MFMAs = 432
      float best = 1e9f;
      unsigned long long hc[2] = {0, 0};
      for (int rep = 0; rep < 4; ++rep) {
        hipMemset(clk, 0, 16);
        hipEventRecord(e0);
        if (ni == 2) hipLaunchKernelGGL((conv_shape<2>), dim3(grid), dim3(256), 46080, 0, out, iters, seed, clk);
        else hipLaunchKernelGGL((conv_shape<4>), dim3(grid), dim3(256), 46080, 0, out, iters, seed, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) { best = ms; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost); }
      }
      const double fl = (double)grid * 4 * iters * 432 * 2.0 * 16 * 16 * 32;
      printf("conv step shape, 2 LDS reads per %2d MFMAs, %d workgroups/CU: %7.3f ms  %7.1f TFLOP/s of fp16 MFMA work (%.0f fp32-grade)  clock %.2f GHz\n",
             3 * ni, wg, best, fl / best * 1e-9, fl / best * 1e-9 / 3, hc[1] ? 0.1 * (double)hc[0] / (double)hc[1] : 0.0);
    }
  return 0;
}
