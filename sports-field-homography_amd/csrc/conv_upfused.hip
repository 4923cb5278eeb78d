// conv_upfused.hip - the first conv of a fused Up block as ONE kernel (round 5; the engine's default for up3 / up4, SFH_UP_SINGLE):
//   conv3x3(cat([skip, ConvTranspose2d(x)])) + BatchNorm + ReLU  (unet/unet_parts.py:52-68)
// Today (engine.UNetEngine, "swap + seed" path): the composed 2x2 conv over the low-resolution x (conv_s3_kernel<KS = 2>,
// up-scatter) writes an fp32 partial in the skip-half conv's accumulator units, and the skip-half 3x3 conv STARTS from it
// (sfh_conv_desc.acc_init): two launches, 2 x 4 bytes per output element of partial traffic (1.9 GB at u4).  Here both halves
// accumulate into the same registers:
//   phase A  acc  = sum over the 2x2 window of x and its channels with the weights of the output pixel's PARITY (py, px)
//            v    = acc * up_scale[co'] + shift_border[class][co']          (exactly the composed conv's epilogue)
//   phase B  v   += sum over the 3x3 window of the skip tensor                (exactly acc_init + the skip-half conv's stage loop)
//            out  = relu(v * scale[co] + shift[co]) -> H2
// with the same products in the same order per output, so the result is BIT-IDENTICAL to the two-launch path.
// The composed weights depend on the output parity, so the 16 pixels of an MFMA column group must share one parity: a wave IS a
// parity class - wave w = (py, px) owns, in a tile of 16 x 32 output pixels, the 8 x 16 pixels (y0 + 2i + py, x0 + 2j + px) as 8
// groups (one per i) of 16 (j = lane & 15) - and covers all 64 couts of the workgroup (four 16-cout groups: 128 accumulator
// registers).  Its operand reads of the low-resolution halo (10 x 18 pixels) are unit-stride, those of the skip halo (18 x 34)
// have a stride of two pixels.  Weights come from the two packed buffers the two-launch path already has (the four quadrant blocks
// of the composed conv, the skip-half's block), per wave from L2, one tap ahead.  One LDS buffer of 78 KB (the skip halo of a
// 32-channel stage), two workgroups per CU.  (Measured and not kept: the skip halo stored with every row's even columns first,
// so that the stride-2 reads become unit-stride - SQ_LDS_BANK_CONFLICT says a third of this kernel's LDS cycles are conflicts -
// bit-identical, same 2.37 ms per step for the two launches: the LDS pipe is not what they wait for.)
#include "common.h"
#include "conv_epilogue.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr unsigned kOOB = 0xFFFFFFF0u;
constexpr int TH = 16, TW = 32;
constexpr int LH = TH / 2 + 2, LW = TW / 2 + 2, LPIX = LH * LW, LPIXP = 192;   // low-resolution halo 10 x 18
constexpr int SHH = TH + 2, SWW = TW + 2, SPIX = SHH * SWW, SPIXP = 624;      // skip halo 18 x 34
constexpr int LSLOTS = 8 * LPIXP, SSLOTS = 8 * SPIXP;                        // [plane 2][channel group 4][pixel] x 16 B

struct UpGeom {
  int tiles_y, tiles_x, ntiles, nblk;
  int xcd_order;   // 1: every XCD owns a contiguous range of pixel tiles and runs a tile's cout blocks back to back (round 6)
  unsigned bytes_skip, bytes_low;
};

__device__ __forceinline__ f16x8 as_hf(const u32x4& v) { return __builtin_bit_cast(f16x8, v); }

__global__ __launch_bounds__(256, 2) void conv_upfused_kernel(const sfh_conv_desc d, const UpGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  u32x4* const lds = reinterpret_cast<u32x4*>(smem_f);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15, lg = lane >> 4;
  const int py = wv >> 1, px = wv & 1, qd = wv;          // the wave's output parity = quadrant of the composed conv
  // Workgroups go round-robin to the 8 XCDs, each with its own 4 MB L2.  Round 6 (profiles/r06_tcc_per_launch.txt): with
  // tile = blockIdx / nblk the cout blocks of ONE pixel tile landed on different XCDs - its skip / low-resolution halos crossed
  // the fabric once per cout block (u3: 3.28 GB of L2 misses for 1.18 GB of tensors) - and neighbouring tiles never shared an L2.
  // Now, as in conv_s3_kernel: XCD x owns the contiguous tile range [x * tpx, (x + 1) * tpx) (whole tile rows next to each
  // other, so vertically adjacent tiles meet their halo rows in that L2) and runs a tile's cout blocks back to back.
  int nb, tile;
  if (g.xcd_order) {
    const int xcd = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
    const int tpx = (g.ntiles + 7) >> 3;
    nb = k % g.nblk;
    tile = xcd * tpx + k / g.nblk;
    if (tile >= g.ntiles || k / g.nblk >= tpx) return;      // (the whole workgroup leaves: uniform)
  } else {
    nb = (int)blockIdx.x % g.nblk;
    tile = (int)blockIdx.x / g.nblk;
  }
  const int tpi = g.tiles_y * g.tiles_x;
  const int b = tile / tpi, tr = tile - b * tpi;
  const int ty = tr / g.tiles_x;
  const int y0 = ty * TH, x0 = (tr - ty * g.tiles_x) * TW;
  const int Y0 = y0 >> 1, X0 = x0 >> 1;
  const int nst_low = d.c1 >> 5, nst_skip = d.c0 >> 5;
  const unsigned nblk_low = (unsigned)d.cs1 >> 5, nblk_skip = (unsigned)d.cs0 >> 5;
  const __amdgpu_buffer_rsrc_t r_skip = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.src0), 0, (int)g.bytes_skip, 0x00020000);
  const __amdgpu_buffer_rsrc_t r_low = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.src1), 0, (int)g.bytes_low, 0x00020000);
  // packed weights, per 64 couts: [stage][tap][plane 2][cout group 4][lane 64][8 x fp16]
  const unsigned wtot_up = (unsigned)nst_low * 4u * 8192u, wtot_sk = (unsigned)nst_skip * 9u * 8192u;
  const __amdgpu_buffer_rsrc_t rw_up = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(d.up_wpacked)) + (size_t)(qd * g.nblk + nb) * wtot_up, 0, (int)wtot_up, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw_sk = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(d.wpacked)) + (size_t)nb * wtot_sk, 0, (int)wtot_sk, 0x00020000);

  f32x4 acc[4][8];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[n][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // one phase = nst stages of 32 channels; a stage = NTAP taps x 8 pixel groups; per (tap, group): two operand reads (the two
  // fp16 planes) + twelve MFMAs (three kept products x four cout groups); the next tap's eight weight fragments one tap ahead,
  // operand reads two steps ahead
  auto run_phase = [&](auto ntap_tag, int nst, const __amdgpu_buffer_rsrc_t& rsrc, const __amdgpu_buffer_rsrc_t& rw,
                       auto&& halo_off, int nslots, int rowstride_bytes, auto&& xaddr) {
    constexpr int NTAP = decltype(ntap_tag)::value;
    constexpr int NSTEP = NTAP * 8;
    for (int st = 0; st < nst; ++st) {
      __syncthreads();                                     // everyone has left the buffer (previous stage / phase)
      const unsigned cb = (unsigned)st * (unsigned)rowstride_bytes;
      for (int s0 = tid; s0 < nslots; s0 += 256)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(lds + (s0 - lane)), 16, (int)halo_off(s0), (int)cb, 0, 0);
      u32x4 wr[2][2][4], xq[3][2];
      auto ld_w = [&](int t, int set) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            wr[set][p][n] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rw, (int)((p * 4 + n) * 1024 + lane * 16), (int)(((unsigned)st * NTAP + (unsigned)t) * 8192u), 0));
      };
      ld_w(0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // the DMA pieces (older than the eight weight loads) have landed
      __syncthreads();
      auto ld_x = [&](int s_, int set) {
        const int t = s_ >> 3, i = s_ & 7;
#pragma unroll
        for (int p = 0; p < 2; ++p) xq[set][p] = lds[xaddr(t, i, p)];
      };
      ld_x(0, 0);
      ld_x(1, 1);
#pragma unroll
      for (int s_ = 0; s_ < NSTEP; ++s_) {
        const int t = s_ >> 3, i = s_ & 7;
        if (s_ + 2 < NSTEP) ld_x(s_ + 2, (s_ + 2) % 3);
        if (i == 0 && t + 1 < NTAP) ld_w(t + 1, (t + 1) & 1);
        constexpr int PW[3] = {0, 1, 0}, PX[3] = {1, 0, 0};
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[n][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_hf(wr[t & 1][PW[k]][n]), as_hf(xq[s_ % 3][PX[k]]), acc[n][i], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (s_ + 2 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        if (i == 0 && t + 1 < NTAP) __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- phase A: the composed 2x2 conv over the low-resolution tensor; window of output (y, x): rows (y >> 1) + py - 1 + a
  {
    auto halo_off = [&](int slot) -> unsigned {
      const int plg = slot / LPIXP, p = slot - plg * LPIXP;
      const int hy = p / LW, hx = p - hy * LW;
      const int y = Y0 - 1 + hy, x = X0 - 1 + hx;
      const bool ok = p < LPIX && y >= 0 && y < d.h1 && x >= 0 && x < d.w1;
      return ok ? ((((unsigned)(b * d.h1 + y) * nblk_low) * 8u + (unsigned)plg) * (unsigned)d.w1 + (unsigned)x) * 16u : kOOB;
    };
    const int base = lg * LPIXP + py * LW + lq + px;
    auto xaddr = [&](int t, int i, int p) { return base + p * 4 * LPIXP + (i + (t >> 1)) * LW + (t & 1); };
    run_phase(std::integral_constant<int, 4>{}, nst_low, r_low, rw_up, halo_off, LSLOTS, 128 * d.w1, xaddr);
  }
  // ---- the composed conv's epilogue, in registers: v = acc * up_scale + shift_border[class of the output pixel]
  const int xo = x0 + 2 * lq + px;
  {
    const int uh = 2 * d.h1, uw = 2 * d.w1;
    const unsigned clsx = (unsigned)(xo == 0 ? 0 : (xo == uw - 1 ? 2 : (xo >= uw ? 3 : 1)));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int yo = y0 + 2 * i + py;
      const unsigned cls = (unsigned)(yo == 0 ? 0 : (yo == uh - 1 ? 2 : (yo >= uh ? 3 : 1))) * 4u + clsx;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int cov = qd * d.cout + nb * 64 + n * 16 + 4 * lg;      // virtual cout of the composed conv
        const f32x4 sc = *reinterpret_cast<const f32x4*>(d.up_scale + cov);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(d.shift_border + (size_t)cls * (size_t)(4 * d.cout) + cov);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[n][i][j] = acc[n][i][j] * sc[j] + sh[j];
      }
    }
  }
  // ---- phase B: the skip half's 3x3 conv starts from v (sfh_conv_desc.acc_init of the two-launch path)
  {
    auto halo_off = [&](int slot) -> unsigned {
      const int plg = slot / SPIXP, p = slot - plg * SPIXP;
      const int hy = p / SWW, hx = p - hy * SWW;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = p < SPIX && y >= 0 && y < d.H && x >= 0 && x < d.W;
      return ok ? ((((unsigned)(b * d.H + y) * nblk_skip) * 8u + (unsigned)plg) * (unsigned)d.W + (unsigned)x) * 16u : kOOB;
    };
    const int base = lg * SPIXP + py * SWW + 2 * lq + px;
    auto xaddr = [&](int t, int i, int p) { return base + p * 4 * SPIXP + (2 * i + t / 3) * SWW + (t % 3); };
    run_phase(std::integral_constant<int, 9>{}, nst_skip, r_skip, rw_sk, halo_off, SSLOTS, 128 * d.W, xaddr);
  }
  // ---- BatchNorm scale / shift, ReLU, two-plane split, store (H2: (B, H, cs/32, 2, 4, W, 8))
  const float dscale = sfh_h2_pow2(d.h2_exp_dst);
  const unsigned cs = (unsigned)d.dst_cs, run = (unsigned)d.W * 16u, planeb = 4u * run;
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(d.dst, 0, (int)kOOB, 0x00020000);
  unsigned over = 0u;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const unsigned co = (unsigned)(nb * 64 + n * 16 + 4 * lg);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(d.scale + co), sh = *reinterpret_cast<const f32x4*>(d.shift + co);
    const unsigned lane_co = ((co >> 5) * 8u + ((co & 31u) >> 3)) * run + ((co >> 2) & 1u) * 8u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4 v = acc[n][i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = v[j] * sc[j] + sh[j];
        if (d.relu) v[j] = sfh_relu(v[j]);
      }
      const int yo = y0 + 2 * i + py;
      const bool ok = yo < d.H && xo < d.W;
      const unsigned voff = ok ? (unsigned)(b * d.H + yo) * ((cs >> 5) * 8u * run) + (unsigned)xo * 16u + lane_co : kOOB;
      sfh_u32x2 pl[2];
      sfh_split4_h2(v, dscale, pl, over);
#pragma unroll
      for (int p = 0; p < 2; ++p) __builtin_amdgcn_raw_buffer_store_b64(pl[p], rd, (int)voff, (int)(p * planeb), 0);
    }
  }
  sfh_h2_report(over, d.h2_overflow, d.h2_range);
}

}  // namespace

extern "C" int sfh_conv_upfused_fwd(const sfh_conv_desc* dp, void* stream_) {
  SFH_REQUIRE(dp, "conv_upfused_fwd: null descriptor");
  const sfh_conv_desc& d = *dp;
  SFH_REQUIRE(d.src0 && d.src1 && d.wpacked && d.up_wpacked && d.up_scale && d.shift_border && d.scale && d.shift && d.dst,
              "conv_upfused_fwd: null pointer");
  SFH_REQUIRE(d.src_fmt == SFH_FMT_H2 && d.dst_fmt == SFH_FMT_H2, "conv_upfused_fwd: H2 sources and destination");
  SFH_REQUIRE(d.batch > 0 && d.H > 0 && d.W > 0 && d.h0 == d.H && d.w0 == d.W && d.h1 > 0 && d.w1 > 0 &&
                  (d.H == 2 * d.h1 || d.H == 2 * d.h1 + 1) && (d.W == 2 * d.w1 || d.W == 2 * d.w1 + 1),
              "conv_upfused_fwd: skip %dx%d against a low-resolution source %dx%d (twice its size, plus at most one padded row / column)",
              d.W, d.H, d.w1, d.h1);
  SFH_REQUIRE(d.cout > 0 && d.cout % 64 == 0 && d.c0 > 0 && d.c0 % 32 == 0 && d.cs0 >= d.c0 && d.cs0 % 32 == 0 && d.c1 > 0 &&
                  d.c1 % 32 == 0 && d.cs1 >= d.c1 && d.cs1 % 32 == 0 && d.dst_cs >= d.cout && d.dst_cs % 32 == 0,
              "conv_upfused_fwd: channel counts (cout %d, skip %d / %d, low %d / %d, dst %d)", d.cout, d.c0, d.cs0, d.c1, d.cs1, d.dst_cs);
  SFH_REQUIRE(!d.residual && !d.dst_pool && !d.head_w && !d.acc_init && !(d.ksplit > 1) && !d.stats_partial && !d.pool0,
              "conv_upfused_fwd: a plain launch (no residual / pooled output / head / acc_init / split-K / statistics)");
  SFH_REQUIRE(d.h2_exp_dst >= -64 && d.h2_exp_dst <= 64, "conv_upfused_fwd: h2_exp_dst=%d out of range (-64 .. 64)", d.h2_exp_dst);
  UpGeom g;
  g.tiles_y = sfh_cdiv(d.H, TH);
  g.tiles_x = sfh_cdiv(d.W, TW);
  g.ntiles = d.batch * g.tiles_y * g.tiles_x;
  g.nblk = d.cout / 64;
  const unsigned long long bs = 4ULL * d.batch * d.H * d.W * d.cs0, bl = 4ULL * d.batch * d.h1 * d.w1 * d.cs1;
  SFH_REQUIRE(bs < kOOB && bl < kOOB && 4ULL * d.batch * d.H * d.W * (unsigned long long)d.dst_cs < kOOB,
              "conv_upfused_fwd: a tensor exceeds the 4 GiB descriptor range");
  g.bytes_skip = (unsigned)bs;
  g.bytes_low = (unsigned)bl;
  g.xcd_order = d.wg_couts == 1 ? 0 : 1;     // (wg_couts = 1: the round-5 order, for same-device A/Bs)
  const long nblocks = g.xcd_order ? (long)sfh_cdiv(g.ntiles, 8) * 8 * g.nblk : (long)g.ntiles * g.nblk;
  SFH_REQUIRE(nblocks < (1L << 31), "conv_upfused_fwd: grid too large");
  sfh_allow_big_lds((const void*)conv_upfused_kernel);
  hipLaunchKernelGGL(conv_upfused_kernel, dim3((unsigned)nblocks), dim3(256), SSLOTS * 16, (hipStream_t)stream_, d, g);
  return sfh_check_launch("conv_upfused_kernel");
}
