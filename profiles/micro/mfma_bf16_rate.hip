// Microbenchmark: achievable v_mfma_f32_16x16x32_bf16 rate in the conv_s3 step shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// MODE 0: operands fixed in registers; MODE 1: x operands re-read from LDS every step (3 b128 / 12 MFMA)
// MODE 2: as 1 plus weight fragments (6 x b128 per 8 steps) streamed from global (L2-resident)
template <int MODE, int CHAIN>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, const u32x4* gw, const unsigned* seed) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  u32x4* lds = (u32x4*)sm;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) lds[i] = (u32x4){seed[i & 255], seed[(i + 1) & 255], seed[(i + 2) & 255], seed[(i + 3) & 255]};
  __syncthreads();
  f32x4 acc[2][8];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
  u32x4 w[3][2], x[3];
  for (int p = 0; p < 3; ++p) { x[p] = lds[lane + 64 * p]; for (int n = 0; n < 2; ++n) w[p][n] = lds[1024 + lane + 64 * (2 * p + n)]; }
  constexpr int PW[6] = {0, 1, 2, 0, 1, 0}, PX[6] = {2, 1, 0, 1, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (MODE == 2) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int n = 0; n < 2; ++n) w[p][n] = gw[((it * 9 + t) & 1023) * 768 + (2 * p + n) * 64 + lane];
      }
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
        if (MODE >= 1) {
#pragma unroll
          for (int p = 0; p < 3; ++p) x[p] = lds[lane + 64 * p + mi * 16 + t * 128 + (it & 1)];
        }
        if (CHAIN) {
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int k6 = 0; k6 < 6; ++k6)
              acc[n][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(w[PW[k6]][n]), bf(x[PX[k6]]), acc[n][mi], 0, 0, 0);
        } else {
#pragma unroll
          for (int k6 = 0; k6 < 6; ++k6)
#pragma unroll
            for (int n = 0; n < 2; ++n)
              acc[n][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(w[PW[k6]][n]), bf(x[PX[k6]]), acc[n][mi], 0, 0, 0);
        }
      }
    }
  }
  f32x4 s = {0, 0, 0, 0};
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 8; ++b) s += acc[a][b];
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

// stage-structured variant mirroring conv_s3_kernel<DB>: 72 steps per stage, LDS double buffer,
// 17 LDS-DMA pieces (64 lanes x 16 B, pixel-strided like the halo) spread over the first steps,
// weight fragments from global one tap ahead, [vmcnt(0); barrier] per stage.
template <int DMA>
__global__ __launch_bounds__(256, 2) void kstage(float* out, int nstages, const u32x4* gw, const unsigned* seed,
                                                 const float* gx, unsigned gxbytes) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  u32x4* lds = (u32x4*)sm;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 4352; i += 256) lds[i] = (u32x4){seed[i & 255], seed[(i + 1) & 255], seed[(i + 2) & 255], seed[(i + 3) & 255]};
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gx), 0, (int)gxbytes, 0x00020000);
  unsigned hoff[17];
  for (int i = 0; i < 17; ++i) {
    if (DMA == 2) {  // 4 lanes cover the 64 contiguous bytes of one (pixel, plane): 16 pixels per piece
      const unsigned sl = tid + 256u * i;
      hoff[i] = ((blockIdx.x * 340u + (sl >> 2) % 352u) * 384u + ((sl >> 2) / 352u) * 64u + (sl & 3u) * 16u) % (gxbytes - 4096u);
    } else {
      hoff[i] = ((blockIdx.x * 340u + (tid + 256u * i) % 352u) * 384u + ((tid + 256u * i) / 352u) * 16u) % (gxbytes - 4096u);
    }
  }
  f32x4 acc[2][8];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
  u32x4 wa[3][2], wb[3][2], x[3][3];
  constexpr int PW[6] = {0, 1, 2, 0, 1, 0}, PX[6] = {2, 1, 0, 1, 0, 0};
  unsigned wi = 0;
  auto load_w = [&](u32x4 (&w)[3][2]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int n = 0; n < 2; ++n) w[p][n] = gw[(wi & 1023) * 768 + (2 * p + n) * 64 + lane];
    ++wi;
  };
  load_w(wa);
  for (int st = 0; st < nstages; ++st) {
    const int cur = st & 1;
    const u32x4* halo = lds + cur * 4352;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto ld_x = [&](int s, int buf) {
#pragma unroll
      for (int p = 0; p < 3; ++p) x[buf][p] = halo[lane + 64 * p + (s % 8) * 16 + (s / 8) * 40 + p * 1408];
    };
    ld_x(0, 0); ld_x(1, 1);
#pragma unroll
    for (int s = 0; s < 72; ++s) {
      const int t = s / 8, mi = s % 8;
      u32x4 (&wc)[3][2] = (t & 1) ? wb : wa;
      u32x4 (&wn)[3][2] = (t & 1) ? wa : wb;
      if (s + 2 < 72) ld_x(s + 2, (s + 2) % 3);
      if (mi == 0) load_w(wn);
      if (DMA && s < 17)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds + (cur ^ 1) * 4352 + wv * 64 + 256 * s), 16, (int)hoff[s], (int)((st & 7) * 192u), 0, 0);
#pragma unroll
      for (int k6 = 0; k6 < 6; ++k6)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[n][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf(wc[PW[k6]][n]), bf(x[s % 3][PX[k6]]), acc[n][mi], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // NTAP odd: swap weight sets by copying (microbench only)
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int n = 0; n < 2; ++n) { u32x4 tmp = wa[p][n]; wa[p][n] = wb[p][n]; wb[p][n] = tmp; }
  }
  f32x4 s = {0, 0, 0, 0};
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 8; ++b) s += acc[a][b];
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

template <class F>
void run(const char* name, F launch, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double mf = (double)blocks * 4 * iters * 9 * 8 * 12;           // MFMA instructions
  double flops = mf * 16384.0;
  printf("%-58s %8.3f ms %7.1f TFLOP/s bf16 (%5.1f TF fp32-equiv) %5.1f cyc/MFMA @2.4GHz\n", name, ms, flops / (ms * 1e-3) / 1e12,
         flops / 6 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (mf / (blocks < 256 ? blocks : 256) / 4 / ((blocks + 255) / 256 > 0 ? 1 : 1)) * ((blocks <= 256) ? 1.0 : 256.0 / blocks));
}

int main() {
  float* out; u32x4* gw; unsigned* seed;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&gw, 1024 * 768 * 16); hipMalloc(&seed, 1024);
  unsigned h[256]; for (int i = 0; i < 256; ++i) h[i] = 0x3f803f80u ^ (i * 2654435761u & 0x007f007f);
  hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(gw, 0x3c, 1024 * 768 * 16);
  const int iters = 32;
  for (int blocks : {256, 512}) {
    printf("blocks=%d (<=2 WG/CU)\n", blocks);
    run("regs only, interleaved chains", [&] { hipLaunchKernelGGL((k<0, 0>), dim3(blocks), dim3(256), 70000, 0, out, iters, gw, seed); }, blocks, iters);
    run("regs only, 6-long dependent chains", [&] { hipLaunchKernelGGL((k<0, 1>), dim3(blocks), dim3(256), 70000, 0, out, iters, gw, seed); }, blocks, iters);
    run("+ LDS operand reads (3 b128 / 12 MFMA)", [&] { hipLaunchKernelGGL((k<1, 0>), dim3(blocks), dim3(256), 70000, 0, out, iters, gw, seed); }, blocks, iters);
    run("+ LDS reads + weight fragments from L2", [&] { hipLaunchKernelGGL((k<2, 0>), dim3(blocks), dim3(256), 70000, 0, out, iters, gw, seed); }, blocks, iters);
  }
  float* gx; const unsigned gxbytes = 512u << 20; hipMalloc(&gx, gxbytes); hipMemset(gx, 0x3c, gxbytes);
  hipFuncSetAttribute((const void*)kstage<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)kstage<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)kstage<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int blocks : {256, 1024}) {
    printf("stage-structured (1 WG/CU, 139 KB LDS), blocks=%d, 32 stages\n", blocks);
    run("barrier per stage, no DMA", [&] { hipLaunchKernelGGL(kstage<0>, dim3(blocks), dim3(256), 139264, 0, out, 32, gw, seed, gx, gxbytes); }, blocks, 32);
    run("barrier per stage + 17 LDS-DMA pieces", [&] { hipLaunchKernelGGL(kstage<1>, dim3(blocks), dim3(256), 139264, 0, out, 32, gw, seed, gx, gxbytes); }, blocks, 32);
    run("  same, pieces = 16 pixels x 64 contiguous B", [&] { hipLaunchKernelGGL(kstage<2>, dim3(blocks), dim3(256), 139264, 0, out, 32, gw, seed, gx, gxbytes); }, blocks, 32);
  }
  return 0;
}
