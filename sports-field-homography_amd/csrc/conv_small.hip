// conv_small.hip - the 3x3 stride-1 two-plane fp16 ("f16x3") convolution for launches whose grid of 256-pixel x 64-cout
// workgroups (conv_s3.hip) leaves the chip under-filled: ResNet layer4 at batch 16 (12x20 frames: 192 workgroups, ONE wave per
// SIMD, 6912 MFMAs per wave - the launch takes what its busiest SIMD takes), small batches everywhere.
//
// Finer work unit, same arithmetic: tile = 12 x 20 pixels = 15 pixel groups of 4 x 4 (a whole layer4 frame: no padded column,
// no shared zero rows) x 32 couts.  Four waves; wave w owns groups 4w .. 4w+3 (group 15 does not exist: it re-reads group
// 14's pixels and its results are dropped) and both 16-cout groups.  A wave's step is conv_s3_kernel's: two operand reads (the
// two fp16 planes of 16 pixels x 32 k) + six MFMAs (three kept products x two cout groups), operand reads two steps ahead.  All
// four waves need the SAME weight fragments, so a stage's 36 KB of them go through LDS next to its 40 KB halo (LDS-DMA both;
// conv_s3_kernel's waves pull their own from L2 - here that would be 4 x the stream).  Weights are read from the SAME packed
// buffer as conv_s3_kernel's (sfh_pack_h2_weights: per 64 couts; a workgroup takes the two cout groups of its half), and every
// output accumulates its products in the same order (stage, tap, product): results are BIT-IDENTICAL to sfh_conv_s3_fwd's.
// Two variants: one 76 KB buffer and two workgroups per CU, or - grids of at most one workgroup per CU - two buffers, the next
// stage's DMA issued before this stage's MFMAs.  Measured (profiles/r05_small_map_probe.txt: the prototype of this kernel
// against conv_s3_kernel on one device): layer4 79 -> 50 us per launch; no gain on grids that already fill the chip.
#include "common.h"
#include "conv_epilogue.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr unsigned kOOB = 0xFFFFFFF0u;

struct SmallCfg {   // what sfh_conv_epilogue needs to know about the tile
  static constexpr int KS = 3, SUBX = 5, SH = 4, SW = 4, NGRP = 15;
  static constexpr bool FLATROWS = false;
  static constexpr int TH = 12, TW = 20, HH = TH + 2, HWD = TW + 2, HPIX = HH * HWD, HPIXP = 320;
  static constexpr int HSLOTS = 8 * HPIXP;       // [plane 2][channel group 4][halo pixel 320] x 16 B = 40 KB
  static constexpr int WSLOTS = 9 * 2 * 2 * 64;  // [tap 9][plane 2][cout group 2][lane 64] x 16 B = 36 KB
  static constexpr int BUF = HSLOTS + WSLOTS;
};

struct SmallGeom {
  int Ho, Wo, rows_total, rows_per_img;   // (rows_* only exist for the epilogue's flattened-row mode, unused here)
  unsigned rows_magic;
  int tiles_y, tiles_x, ntiles, nblk;
  int tile_stationary;   // 1: an XCD owns a contiguous tile range and walks all cout blocks per tile (weights of the layer fit L2)
  unsigned bytes0;
};

__device__ __forceinline__ f16x8 as_hf(const u32x4& v) { return __builtin_bit_cast(f16x8, v); }

template <bool DB>
__global__ __launch_bounds__(256, DB ? 1 : 2) void conv_small_kernel(const sfh_conv_desc d, const SmallGeom g) {
  using C = SmallCfg;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  u32x4* const lds = reinterpret_cast<u32x4*>(smem_f);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  // workgroup -> (32-cout block, tile): an XCD (blockIdx & 7) keeps to nblk / 8 cout blocks, so that their weights stay in its L2
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  int nb, tile;
  if (g.tile_stationary) {
    // round 6 (profiles/r06_tcc_per_launch.txt): with the weight-stationary order below every XCD reads the WHOLE input tensor
    // (ResNet layer3: 147 MB of L2 misses per launch for 33 MB of tensors).  Where all weights of the layer fit an XCD's L2
    // (layer3: 2.4 MB) the other order is the cheap one: an XCD owns a contiguous range of tiles, fetches their halos once and
    // finds every cout block's weights in its L2 after the first tile.
    const int tpx = (g.ntiles + 7) >> 3;
    nb = idx % g.nblk;
    tile = xcd * tpx + idx / g.nblk;
    if (idx / g.nblk >= tpx) return;
  } else if (g.nblk >= 8 && (g.nblk & 7) == 0) {
    const int per = g.nblk >> 3;
    nb = xcd * per + idx % per;
    tile = idx / per;
  } else {
    nb = idx % g.nblk;
    tile = (idx / g.nblk) * 8 + xcd;
  }
  if (tile >= g.ntiles) return;
  const int tpi = g.tiles_y * g.tiles_x;
  const int b = tile / tpi, tr = tile - b * tpi;
  const int ty = tr / g.tiles_x;
  const int y0 = ty * C::TH, x0 = (tr - ty * g.tiles_x) * C::TW;
  const int nst = d.c0 >> 5;
  const unsigned nblk_src = (unsigned)d.cs0 >> 5;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.src0), 0, (int)g.bytes0, 0x00020000);
  // packed weights [cout block of 64][stage][tap][plane 2][cout group 4][lane 64][8 x fp16]: this workgroup's half of a block
  const unsigned wtotal = (unsigned)nst * 9u * 8192u;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(d.wpacked)) + (size_t)(nb >> 1) * wtotal, 0, (int)wtotal, 0x00020000);
  // halo slots of this thread: slot = tid + 256 * i -> (plane / channel group, halo pixel); out-of-frame pixels read zeros
  constexpr int NSL = C::HSLOTS / 256;   // 10
  unsigned hoff[NSL];
#pragma unroll
  for (int i = 0; i < NSL; ++i) {
    const int slot = tid + 256 * i;
    const int plg = slot / C::HPIXP, p = slot - plg * C::HPIXP;
    const int hy = p / C::HWD, hx = p - hy * C::HWD;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = p < C::HPIX && y >= 0 && y < d.H && x >= 0 && x < d.W;
    hoff[i] = ok ? ((((unsigned)(b * d.H + y) * nblk_src) * 8u + (unsigned)plg) * (unsigned)d.W + (unsigned)x) * 16u : kOOB;
  }
  // weight slots: tap i of the stage = 256 slots = [plane 2][cout group 2][lane 64]; thread tid fetches plane tid >> 7, cout
  // group (tid >> 6) & 1 of the workgroup's half
  const unsigned woff = (unsigned)(tid >> 7) * 4096u + (unsigned)(2 * (nb & 1) + ((tid >> 6) & 1)) * 1024u + (unsigned)(tid & 63) * 16u;
  auto dma_stage = [&](int st, int buf) {
    const unsigned cb = (unsigned)st * 128u * (unsigned)d.W;   // the row's next 32-channel block: 8 runs of W x 16 bytes
    u32x4* const base = lds + buf * C::BUF;
#pragma unroll
    for (int i = 0; i < NSL; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + wv * 64 + 256 * i), 16, (int)hoff[i], (int)cb, 0, 0);
#pragma unroll
    for (int i = 0; i < 9; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + C::HSLOTS + wv * 64 + 256 * i), 16, (int)woff,
                                               (int)(((unsigned)st * 9u + (unsigned)i) * 8192u), 0, 0);
  };
  int pixbase[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    int gi = 4 * wv + mi;
    if (gi > C::NGRP - 1) gi = C::NGRP - 1;
    const int gy = gi / C::SUBX, gx = gi - gy * C::SUBX;
    pixbase[mi] = lg * C::HPIXP + (gy * 4 + (lq >> 2)) * C::HWD + gx * 4 + (lq & 3);
  }
  f32x4 acc[2][4];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[n][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // one stage = 9 taps x 4 pixel groups; operand reads two steps (12 MFMAs) ahead of their use, the next tap's weight fragments
  // one tap ahead, every step's LDS reads pinned right behind its first MFMA
  auto compute = [&](int buf) {
    const u32x4* const hl = lds + buf * C::BUF;
    const u32x4* const wl = hl + C::HSLOTS;
    u32x4 wr[2][2][2], xq[3][2];
    auto ld_w = [&](int t, int set) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int n = 0; n < 2; ++n) wr[set][p][n] = wl[((t * 2 + p) * 2 + n) * 64 + lane];
    };
    auto ld_x = [&](int s_, int set) {
      const int t = s_ >> 2, mi = s_ & 3;
      const int toff = (t / 3) * C::HWD + (t % 3);
#pragma unroll
      for (int p = 0; p < 2; ++p) xq[set][p] = hl[pixbase[mi] + p * 4 * C::HPIXP + toff];
    };
    ld_w(0, 0);
    ld_x(0, 0);
    ld_x(1, 1);
#pragma unroll
    for (int s_ = 0; s_ < 36; ++s_) {
      const int t = s_ >> 2, mi = s_ & 3;
      if (s_ + 2 < 36) ld_x(s_ + 2, (s_ + 2) % 3);
      if (mi == 0 && t + 1 < 9) ld_w(t + 1, (t + 1) & 1);
      // the kept products in conv_s3_kernel's order (smallest first): w0*x1, w1*x0, w0*x0
      constexpr int PW[3] = {0, 1, 0}, PX[3] = {1, 0, 0};
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[n][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_hf(wr[t & 1][PW[k]][n]), as_hf(xq[s_ % 3][PX[k]]), acc[n][mi], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (mi == 0 && t + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
      else if (s_ + 2 < 36) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (DB) {
    dma_stage(0, 0);
    for (int st = 0; st < nst; ++st) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                       // stage st has landed for every wave; everyone has left the other buffer
      if (st + 1 < nst) dma_stage(st + 1, (st + 1) & 1);
      compute(st & 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (no DMA is in flight here; kept beside the rule of conv_s3_kernel)
  } else {
    for (int st = 0; st < nst; ++st) {
      dma_stage(st, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(0);
      __syncthreads();   // the buffer is free again
    }
  }
  sfh_conv_epilogue<C, 2, 4, 2>(d, g, acc, nb * 32, 4 * wv, (b << 16) | y0, x0, lq, lg);
}

}  // namespace

extern "C" int sfh_conv_small_fwd(const sfh_conv_desc* dp, void* stream_) {
  SFH_REQUIRE(dp, "conv_small_fwd: null descriptor");
  sfh_conv_desc d = *dp;
  SFH_REQUIRE(d.src0 && d.wpacked && d.scale && d.shift && d.dst, "conv_small_fwd: null pointer");
  SFH_REQUIRE(d.src_fmt == SFH_FMT_H2 && (d.dst_fmt == SFH_FMT_H2 || d.dst_fmt == SFH_FMT_F32),
              "conv_small_fwd: H2 sources, an H2 or fp32 destination (src_fmt=%d, dst_fmt=%d)", d.src_fmt, d.dst_fmt);
  SFH_REQUIRE(d.ksize == 3 && d.stride == 1 && !d.src1 && !d.pool0 && !d.dst_pool && !d.head_w && !d.acc_init && !(d.ksplit > 1) &&
                  !d.stats_partial && !d.bwd_z && d.out_mode == SFH_OUT_NHWC && !d.shift_border && !d.residual_f32,
              "conv_small_fwd: a plain 3x3 stride-1 conv (one source, optional residual of the destination's format, ReLU)");
  SFH_REQUIRE(d.batch > 0 && d.batch < 32768 && d.H > 0 && d.W > 0 && d.H < 65536 && d.h0 == d.H && d.w0 == d.W,
              "conv_small_fwd: bad geometry b=%d %dx%d (source %dx%d)", d.batch, d.H, d.W, d.h0, d.w0);
  SFH_REQUIRE(d.cout > 0 && d.cout % 64 == 0 && d.c0 > 0 && d.c0 % 32 == 0 && d.cs0 >= d.c0 && d.cs0 % 32 == 0,
              "conv_small_fwd: cout=%d must be a multiple of 64, c0=%d of 32 (cs0=%d)", d.cout, d.c0, d.cs0);
  SFH_REQUIRE(d.h2_exp_dst >= -64 && d.h2_exp_dst <= 64 && d.h2_exp_res >= -64 && d.h2_exp_res <= 64,
              "conv_small_fwd: h2_exp_dst=%d / h2_exp_res=%d out of range (-64 .. 64)", d.h2_exp_dst, d.h2_exp_res);
  SmallGeom g;
  g.Ho = d.H;
  g.Wo = d.W;
  g.rows_per_img = d.H;
  g.rows_total = d.batch * d.H;
  g.rows_magic = 0u;
  g.tiles_y = sfh_cdiv(d.H, SmallCfg::TH);
  g.tiles_x = sfh_cdiv(d.W, SmallCfg::TW);
  g.ntiles = d.batch * g.tiles_y * g.tiles_x;
  g.nblk = d.cout / 32;
  const unsigned long long b0 = 4ULL * d.batch * d.H * d.W * d.cs0;
  SFH_REQUIRE(b0 < kOOB && 4ULL * d.batch * d.H * d.W * (unsigned long long)d.dst_cs < kOOB,
              "conv_small_fwd: a tensor exceeds the 4 GiB descriptor range");
  g.bytes0 = (unsigned)b0;
  const long nblocks = (long)sfh_cdiv(g.ntiles, 8) * 8 * g.nblk;
  SFH_REQUIRE(nblocks < (1L << 31), "conv_small_fwd: grid too large");
  hipStream_t stream = (hipStream_t)stream_;
  // at most one workgroup per CU (256 CUs): two LDS buffers, the DMA of stage s + 1 under the MFMAs of stage s
  // (sfh_conv_desc.wg_couts, which has no other meaning here, overrides: 1 = one buffer, 2 = two buffers)
  // (+ 16: experiment knob - keep the weight-stationary block order whatever the weights' size)
  const int force_ws = d.wg_couts >= 16;
  d.wg_couts &= 15;
  SFH_REQUIRE(d.wg_couts >= 0 && d.wg_couts <= 2, "conv_small_fwd: wg_couts=%d (0 = the launcher decides, 1 = one LDS buffer, 2 = two)", d.wg_couts);
  // all weights of the layer (two fp16 planes) within 3 MB of an XCD's 4 MB L2 -> tiles stay put, weights are found in L2
  g.tile_stationary = (!force_ws && (long)d.cout * d.c0 * 9 * 4 <= (3L << 20)) ? 1 : 0;
  if (d.wg_couts == 2 || (d.wg_couts == 0 && (long)g.ntiles * g.nblk <= 256)) {
    sfh_allow_big_lds((const void*)conv_small_kernel<true>);
    hipLaunchKernelGGL(conv_small_kernel<true>, dim3((unsigned)nblocks), dim3(256), 2 * SmallCfg::BUF * 16, stream, d, g);
  } else {
    sfh_allow_big_lds((const void*)conv_small_kernel<false>);
    hipLaunchKernelGGL(conv_small_kernel<false>, dim3((unsigned)nblocks), dim3(256), SmallCfg::BUF * 16, stream, d, g);
  }
  return sfh_check_launch("conv_small_kernel");
}
