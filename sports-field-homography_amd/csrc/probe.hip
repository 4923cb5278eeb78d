// probe.hip - device calibration probe (include/sfh_amd.h: sfh_probe_mfma_f16).  Not on the hot path: bench.py runs it
// once per rank so that every bench line carries what THIS device sustains (the pool's MI355X differ by up to 4 % in the
// clock they hold under an MFMA-dense load, which is more than most kernel changes are worth).
#include "common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Register-resident v_mfma_f32_16x16x32_f16 loop: 16 independent accumulators per wave, operands = fp16 values in
// [0.5, 1) with random mantissas (the matrix cores' power draw, and with it the clock the chip holds, follows the data).
// Per iteration 64 MFMAs per wave.  Shader-clock cycles (s_memtime) and 100 MHz ticks (s_memrealtime) over the loop are
// summed by one wave of every 32nd workgroup: clock = 100 MHz * clk[0] / clk[1].
__global__ __launch_bounds__(256, 2) void probe_mfma_f16_kernel(float* out, int iters, unsigned long long* clk) {
  const int tid = threadIdx.x, lane = tid & 63;
  u32x4 a[4], b[4];
  unsigned s = 0x9E3779B9u * (unsigned)(lane + 1) + 0x7F4A7C15u * (unsigned)(tid >> 6);
  for (int i = 0; i < 4; ++i) {
    unsigned w[8];
    for (int j = 0; j < 8; ++j) {
      s = s * 1664525u + 1013904223u;
      const unsigned r = s >> 8;
      w[j] = (0x3800u | (r & 0x3FFu)) | ((0x3800u | ((r >> 10) & 0x3FFu)) << 16);
    }
    a[i] = (u32x4){w[0], w[1], w[2], w[3]};
    b[i] = (u32x4){w[4], w[5], w[6], w[7]};
  }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  unsigned long long c0, r0, c1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(i + k) & 3]),
                                                        __builtin_bit_cast(f16x8, b[(i >> 2) ^ k]), acc[i], 0, 0, 0);
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
  f32x4 t = {0, 0, 0, 0};
  for (int i = 0; i < 16; ++i) t += acc[i];
  out[(size_t)blockIdx.x * 256 + tid] = t[0] + t[1] + t[2] + t[3];
  if (lane == 0 && (blockIdx.x & 31) == 0) {
    atomicAdd(&clk[0], c1 - c0);
    atomicAdd(&clk[1], r1 - r0);
  }
}

extern "C" int sfh_probe_mfma_f16(int iters, int workgroups, float* out, uint64_t* clk, void* stream) {
  SFH_REQUIRE(out && clk, "probe_mfma_f16: null pointer");
  SFH_REQUIRE(iters > 0 && iters <= (1 << 20) && workgroups > 0 && workgroups <= (1 << 16),
              "probe_mfma_f16: iters=%d workgroups=%d out of range", iters, workgroups);
  hipLaunchKernelGGL(probe_mfma_f16_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, out, iters,
                     (unsigned long long*)clk);
  return sfh_check_launch("probe_mfma_f16_kernel");
}
