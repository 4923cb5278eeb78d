// Prototype for the review's "ResNet-STN tile family" item (round 5): does a FINER work unit shorten the small-map 3x3
// launches of ResNet layer3 / layer4?  Stand-alone (no product code), timing only - random operands, fp32 NHWC output.
//
// The product runs these launches as ONE round of 256-pixel x 64-cout workgroups with one wave per SIMD: layer4 (12x20 frames,
// 512 -> 512, batch 16) = 192 workgroups x 6912 MFMAs per wave = 58 us at 100 % issue, measured 83; layer3 (24x40, 256 -> 256)
// = 256 workgroups x 3456 = 29 us, measured 51.  The shape tried here:
//   * tile = 12 x 20 pixels = 15 pixel groups of 4 x 4 (a whole layer4 frame, a quarter of a layer3 frame: no padded column,
//     no shared zero rows) x 32 couts; 4 waves, wave w owns groups 4w .. 4w+3 (group 15 is a dummy) and both cout groups:
//     layer4 = 256 workgroups of 3240 MFMAs per wave, layer3 = 512 workgroups (two per CU) of 1620;
//   * a wave's step is the product's: two operand reads (two fp16 planes of 16 pixels x 32 k) + six MFMAs (three products x two
//     cout groups); all four waves need the SAME weights, so the 36 KB of a stage's weight fragments go through LDS with the
//     40 KB halo (LDS-DMA, one buffer, two workgroups per CU) instead of 4 x from L2.
//   hipcc --offload-arch=gfx950 -O3 -o small_map_conv_probe small_map_conv_probe.hip && ./small_map_conv_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 hf(const u32x4& v) { return __builtin_bit_cast(f16x8, v); }

constexpr int TH = 12, TW = 20, HH = TH + 2, HWD = TW + 2, HPIX = HH * HWD, HPIXP = 320;
constexpr int HSLOTS = 8 * HPIXP;      // [plane 2][group 4][pixel 320] x 16 B = 40 KB
constexpr int WSLOTS = 9 * 2 * 2 * 64;  // [tap 9][plane 2][cout group 2][lane 64] x 16 B = 36 KB
constexpr unsigned kOOB = 0xFFFFFFF0u;

struct Geo {
  int B, H, W, C, Cout, tiles_y, tiles_x, nblk;
};

// DB = false: one 76 KB buffer, two workgroups per CU (each other's DMA wait is the other's compute);
// DB = true : two buffers (152 KB), one workgroup per CU, the next stage's DMA issued before this stage's MFMAs: for grids of at
//             most one workgroup per CU (layer4), where nothing else would cover the DMA latency of every stage
template <bool DB, bool WLDS>
__global__ __launch_bounds__(256, DB ? 1 : 2) void small_conv(const unsigned short* __restrict__ x, const unsigned short* __restrict__ wp,
                                                              float* __restrict__ out, Geo g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  u32x4* const lds = reinterpret_cast<u32x4*>(smem);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  // workgroup -> (cout block, tile): an XCD (blockIdx & 7) keeps to nblk / 8 cout blocks, so that their weights stay in its L2
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  int nb, tile;
  if (g.nblk >= 8) {
    const int per = g.nblk >> 3;
    nb = xcd * per + idx % per;
    tile = idx / per;
  } else {
    nb = idx % g.nblk;
    tile = (idx / g.nblk) * 8 + xcd;
  }
  const int ntiles = g.B * g.tiles_y * g.tiles_x;
  if (tile >= ntiles) return;
  const int b = tile / (g.tiles_y * g.tiles_x), tr = tile - b * (g.tiles_y * g.tiles_x);
  const int y0 = (tr / g.tiles_x) * TH, x0 = (tr % g.tiles_x) * TW;
  const int nst = g.C / 32;
  const unsigned xbytes = (unsigned)((size_t)g.B * g.H * nst * 8 * g.W * 16);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(x), 0, (int)xbytes, 0x00020000);
  const unsigned wbytes = (unsigned)nst * WSLOTS * 16u;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(wp) + (size_t)nb * nst * WSLOTS * 8, 0, (int)wbytes, 0x00020000);
  // halo slots of this thread: slot = tid + 256 * i -> (plane/group, halo pixel)
  constexpr int NSL = HSLOTS / 256;   // 10
  unsigned hoff[NSL];
#pragma unroll
  for (int i = 0; i < NSL; ++i) {
    const int slot = tid + 256 * i;
    const int plg = slot / HPIXP, p = slot - plg * HPIXP;
    const int hy = p / HWD, hx = p - hy * HWD;
    const int y = y0 - 1 + hy, xx = x0 - 1 + hx;
    const bool ok = p < HPIX && y >= 0 && y < g.H && xx >= 0 && xx < g.W;
    // (B, H, C/32, 2, 4, W, 8) halfs: byte offset of (b, y, block 0, plg, xx)
    hoff[i] = ok ? ((((unsigned)(b * g.H + y) * (unsigned)nst) * 8u + (unsigned)plg) * (unsigned)g.W + (unsigned)xx) * 16u : kOOB;
  }
  constexpr int NWL = WSLOTS / 256;   // 9
  constexpr int BUF = HSLOTS + (WLDS ? WSLOTS : 0);
  auto dma_stage = [&](int st, int buf) {
    const unsigned cb = (unsigned)st * 8u * (unsigned)g.W * 16u;   // next 32-channel block of the same row
    u32x4* const base = lds + buf * BUF;
#pragma unroll
    for (int i = 0; i < NSL; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + wv * 64 + 256 * i), 16, (int)hoff[i], (int)cb, 0, 0);
    if (WLDS) {
#pragma unroll
      for (int i = 0; i < NWL; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + HSLOTS + wv * 64 + 256 * i), 16, (int)((tid + 256 * i) * 16),
                                                 (int)((unsigned)st * WSLOTS * 16u), 0, 0);
    }
  };
  // this wave's four pixel groups (group 15 does not exist: it re-reads group 14's pixels, its results are dropped)
  int pixbase[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    int gi = 4 * wv + mi;
    if (gi > 14) gi = 14;
    const int gy = gi / 5, gx = gi - gy * 5;
    pixbase[mi] = lg * HPIXP + (gy * 4 + (lq >> 2)) * HWD + gx * 4 + (lq & 3);
  }
  f32x4 acc[2][4];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[n][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // explicit software pipeline, as in the product kernel: operand reads two steps (12 MFMAs) ahead of their use, the next
  // tap's weight fragments one tap ahead, every step's LDS reads pinned right behind its first MFMA
  auto compute = [&](int buf, int st) {
    const u32x4* const hl = lds + buf * BUF;
    const u32x4* const wl = hl + HSLOTS;
    u32x4 wr[2][2][2], xq[3][2];
    auto ld_w = [&](int t, int set) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          if (WLDS) wr[set][p][n] = wl[((t * 2 + p) * 2 + n) * 64 + lane];
          else wr[set][p][n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                   rw, (int)((((t * 2 + p) * 2 + n) * 64 + lane) * 16), (int)((unsigned)st * WSLOTS * 16u), 0));
        }
    };
    auto ld_x = [&](int s_, int set) {
      const int t = s_ >> 2, mi = s_ & 3;
      const int toff = (t / 3) * HWD + (t % 3);
#pragma unroll
      for (int p = 0; p < 2; ++p) xq[set][p] = hl[pixbase[mi] + p * 4 * HPIXP + toff];
    };
    ld_w(0, 0);
    ld_x(0, 0);
    ld_x(1, 1);
#pragma unroll
    for (int s_ = 0; s_ < 36; ++s_) {
      const int t = s_ >> 2, mi = s_ & 3;
      if (s_ + 2 < 36) ld_x(s_ + 2, (s_ + 2) % 3);
      if (mi == 0 && t + 1 < 9) ld_w(t + 1, (t + 1) & 1);
      constexpr int PW[3] = {0, 1, 0}, PX[3] = {1, 0, 0};
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[n][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hf(wr[t & 1][PW[k]][n]), hf(xq[s_ % 3][PX[k]]), acc[n][mi], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (WLDS) {
        if (mi == 0 && t + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
        else if (s_ + 2 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      } else {
        if (mi == 0 && t + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);   // the next tap's weight fragments (VMEM)
        if (s_ + 2 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (DB) {
    dma_stage(0, 0);
    for (int st = 0; st < nst; ++st) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                       // stage st has landed for every wave; everyone is done with the other buffer
      if (st + 1 < nst) dma_stage(st + 1, (st + 1) & 1);
      compute(st & 1, st);
    }
  } else {
    for (int st = 0; st < nst; ++st) {
      dma_stage(st, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0, st);
      __syncthreads();   // the buffer is free again
    }
  }
  // plain fp32 NHWC store (timing prototype: no BatchNorm / ReLU / split)
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int gi = 4 * wv + mi;
    if (gi > 14) continue;
    const int gy = gi / 5, gx = gi - gy * 5;
    const int y = y0 + gy * 4 + (lq >> 2), xx = x0 + gx * 4 + (lq & 3);
    if (y < g.H && xx < g.W) {
#pragma unroll
      for (int n = 0; n < 2; ++n)
        *reinterpret_cast<f32x4*>(out + ((size_t)(b * g.H + y) * g.W + xx) * g.Cout + nb * 32 + n * 16 + 4 * lg) = acc[n][mi];
    }
  }
}

template <bool DB, bool WLDS>
static void run(const char* name, int B, int H, int W, int C, int Cout) {
  Geo g{B, H, W, C, Cout, (H + TH - 1) / TH, (W + TW - 1) / TW, Cout / 32};
  const size_t xh = (size_t)B * H * (C / 32) * 8 * W * 8, wh = (size_t)(Cout / 32) * (C / 32) * WSLOTS * 8;
  std::vector<unsigned short> hx(xh), hw(wh);
  for (auto& v : hx) v = (unsigned short)(0x3000u | (rand() & 0x0FFF));   // fp16 in [0.125, 0.5)
  for (auto& v : hw) v = (unsigned short)((rand() & 0x8000) | 0x2C00u | (rand() & 0x03FF));
  unsigned short *dx, *dw;
  float* dout;
  hipMalloc(&dx, xh * 2 + 4096);
  hipMalloc(&dw, wh * 2 + 4096);
  hipMalloc(&dout, (size_t)B * H * W * Cout * 4);
  hipMemcpy(dx, hx.data(), xh * 2, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), wh * 2, hipMemcpyHostToDevice);
  const int ntiles = B * g.tiles_y * g.tiles_x;
  const int grid = ((ntiles + 7) / 8 * 8) * g.nblk;
  const int ldsb = (HSLOTS + (WLDS ? WSLOTS : 0)) * 16 * (DB ? 2 : 1);
  hipFuncSetAttribute((const void*)small_conv<DB, WLDS>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((small_conv<DB, WLDS>), dim3(grid), dim3(256), ldsb, 0, dx, dw, dout, g);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms / 10 < best) best = ms / 10;
  }
  const double fl = 2.0 * B * H * W * (double)Cout * 9 * C;
  printf("%-28s %4d -> %-4d %3dx%-3d batch %d: %d workgroups (tile 12x20 x 32 couts, %s, weights %s)  %7.1f us per launch  %6.1f TFLOP/s fp32-grade  [%s]\n",
         name, C, Cout, H, W, B, grid, DB ? "two LDS buffers, 1 per CU" : "one LDS buffer, 2 per CU", WLDS ? "through LDS" : "L2 -> registers", best * 1e3, fl / best * 1e-9, hipGetErrorString(hipGetLastError()));
  hipFree(dx);
  hipFree(dw);
  hipFree(dout);
}

int main() {
  run<false, true>("ResNet layer4 conv", 16, 12, 20, 512, 512);
  run<true, true>("ResNet layer4 conv", 16, 12, 20, 512, 512);
  run<false, false>("ResNet layer4 conv", 16, 12, 20, 512, 512);
  run<true, false>("ResNet layer4 conv", 16, 12, 20, 512, 512);
  run<false, true>("ResNet layer3 conv", 16, 23, 40, 256, 256);
  run<true, true>("ResNet layer3 conv", 16, 23, 40, 256, 256);
  run<false, false>("ResNet layer3 conv", 16, 23, 40, 256, 256);
  run<true, false>("ResNet layer3 conv", 16, 23, 40, 256, 256);
  run<false, true>("ResNet layer2 conv", 16, 45, 80, 128, 128);
  run<false, false>("ResNet layer2 conv", 16, 45, 80, 128, 128);
  run<false, true>("ResNet layer1 conv", 16, 90, 160, 64, 64);
  run<false, false>("ResNet layer1 conv", 16, 90, 160, 64, 64);
  run<false, true>("UNet d4.3", 16, 22, 40, 1024, 1024);
  run<false, false>("UNet d4.3", 16, 22, 40, 1024, 1024);
  run<false, false>("UNet d3.3", 16, 45, 80, 512, 512);
  return 0;
}
