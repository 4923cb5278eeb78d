// Microbenchmark: achievable fp32 MFMA rate of the conv inner-loop shapes (GPU box only).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k16(float* out, int iters, const float* in) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  f32x4* lds = (f32x4*)sm;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 3584; i += 256) lds[i] = (f32x4){in[i & 255], 1.f, 2.f, 3.f};
  __syncthreads();
  f32x4 acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
  f32x4 xv[4], wv[4];
  for (int m = 0; m < 4; ++m) { xv[m] = lds[lane + 64 * m]; wv[m] = lds[1024 + lane + 64 * m]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (MODE == 1) {  // operand reads from LDS each tap (like the conv loop)
#pragma unroll
        for (int m = 0; m < 4; ++m) { xv[m] = lds[lane + 64 * m + t * 16 + (it & 1)]; wv[m] = lds[1024 + lane + 64 * m + t * 256]; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[n][j], xv[m][j], acc[n][m], 0, 0, 0);
    }
  }
  f32x4 s = {0, 0, 0, 0};
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b];
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256, 2) void k32(float* out, int iters, const float* in) {
  const int tid = threadIdx.x;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0;
  float a0 = in[tid & 255], a1 = in[(tid + 7) & 255], b0 = in[(tid + 3) & 255], b1 = in[(tid + 11) & 255];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 72; ++t) {  // 72*4 MFMA 32x32x2 = same FLOPs as 576 16x16x4
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
  }
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + tid] = s;
}

// MODE 2: + one barrier per stage; MODE 3: + 15 LDS-DMA (58 KB per stage per WG) from a
// global buffer into the other half of LDS; MODE 4: 15 plain buffer loads -> regs -> ds_write
template <int MODE>
__global__ __launch_bounds__(256, 2) void kst(float* out, int iters, const float* in, const float* gbuf, unsigned gbytes) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  f32x4* lds = (f32x4*)sm;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * 3840; i += 256) lds[i] = (f32x4){in[i & 255], 1.f, 2.f, 3.f};
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gbuf), 0, (int)gbytes, 0x00020000);
  f32x4 acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
  f32x4 xv[4], wv4[4];
  f32x4 stg[15];
  const unsigned base = (blockIdx.x * 61440u) % (gbytes - 61440u * 2);
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    const f32x4* rd = lds + cur * 3840;
    f32x4* wr = lds + (cur ^ 1) * 3840;
    if (MODE >= 2) __syncthreads();
    if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 15; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(wr + i * 256 + wv * 64), 16, (int)(tid * 16u), (int)(base + i * 4096u + (it & 7) * 61440u % 4096u), 0, 0);
    }
    if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 15; ++i) wr[i * 256 + tid] = stg[i];
#pragma unroll
      for (int i = 0; i < 15; ++i)
        stg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(tid * 16u), (int)(base + i * 4096u), 0));
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int m = 0; m < 4; ++m) { xv[m] = rd[lane + 64 * m + t * 16]; wv4[m] = rd[1536 + lane + 64 * m + t * 256]; }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv4[n][j], xv[m][j], acc[n][m], 0, 0, 0);
    }
  }
  f32x4 s = {0, 0, 0, 0};
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b];
  out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

template <class F>
void run(const char* name, F launch, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flops = (double)blocks * 4 * iters * 576 * 2048.0;
  printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 4096);
  hipMemset(in, 0, 4096);
  float h[256]; for (int i = 0; i < 256; ++i) h[i] = 0.001f * (i % 17) - 0.005f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 64;
  float* gbuf; const unsigned gbytes = 64u << 20; hipMalloc(&gbuf, gbytes); hipMemset(gbuf, 0, gbytes);
  hipFuncSetAttribute((const void*)kst<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)kst<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)kst<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int blocks : {256, 1024}) {
    printf("stage-structured, blocks=%d (1 WG/CU, 120 KB LDS)\n", blocks);
    run("  + barrier per 576 MFMA", [&] { hipLaunchKernelGGL(kst<2>, dim3(blocks), dim3(256), 122880, 0, out, iters, in, gbuf, gbytes); }, blocks, iters);
    run("  + barrier + 15 LDS-DMA per stage", [&] { hipLaunchKernelGGL(kst<3>, dim3(blocks), dim3(256), 122880, 0, out, iters, in, gbuf, gbytes); }, blocks, iters);
    run("  + barrier + 15 buffer loads + ds_write", [&] { hipLaunchKernelGGL(kst<4>, dim3(blocks), dim3(256), 122880, 0, out, iters, in, gbuf, gbytes); }, blocks, iters);
  }
  for (int blocks : {256, 512, 1024}) {
    printf("blocks=%d\n", blocks);
    run("16x16x4 regs only, 58KB LDS (2 WG/CU)", [&] { hipLaunchKernelGGL(k16<0>, dim3(blocks), dim3(256), 59392, 0, out, iters, in); }, blocks, iters);
    run("16x16x4 + LDS operand reads (2 WG/CU)", [&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 59392, 0, out, iters, in); }, blocks, iters);
    hipFuncSetAttribute((const void*)k16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    run("16x16x4 regs only, 120KB LDS (1 WG/CU)", [&] { hipLaunchKernelGGL(k16<0>, dim3(blocks), dim3(256), 122880, 0, out, iters, in); }, blocks, iters);
    run("16x16x4 + LDS operand reads (1 WG/CU)", [&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 122880, 0, out, iters, in); }, blocks, iters);
    run("32x32x2 regs only (2 WG/CU)", [&] { hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 59392, 0, out, iters, in); }, blocks, iters);
    run("32x32x2 regs only (1 WG/CU)", [&] { hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 122880, 0, out, iters, in); }, blocks, iters);
  }
  return 0;
}
