// conv_s3.hip - fp32-grade convolution on the 16-bit matrix cores with split operands.
//
// Two operand formats, one kernel source (S3Cfg::NP = planes per operand):
//   * H2 ("f16x3", the default of the engines): every value is two fp16 planes of u = v * 2^e,
//     p0 = f16(u), p1 = f16(u - p0) - 22 significand bits; e per tensor for activations (default 2), per layer for weights
//     (max |w| 2^e in [2^13, 2^14), folded into the epilogue scale).  A product is accumulated from
//            w*x  ~=  w0x1 + w1x0 + w0x0                                  (dropped term <= 2^-22)
//     by v_mfma_f32_16x16x32_f16 in fp32: 3 MFMAs per 32 k, ceiling 2.5 PFLOP/s / 3 = 833 TFLOP/s of fp32-grade
//     work.  The operand representation perturbs a whole forward pass less than the accumulation order of an
//     fp32 run does (tests/test_oracle.py::test_f16x3_operand_representation_...); values beyond fp16's range
//     raise sfh_conv_desc.h2_overflow / h2_range: the host lowers the tensor's exponent and repeats from that layer.
//   * S3 ("bf16x6"): three bf16 planes v = v0 + v1 + v2 (v0 = bf16(v), v1 = bf16(v - v0), v2 = bf16(v - v0 - v1):
//     exact, 3 x 8 significand bits, fp32's exponent range), six partial products
//            w*x  ~=  w0x2 + w1x1 + w2x0 + w0x1 + w1x0 + w0x0            (dropped terms <= 2^-24)
//     by v_mfma_f32_16x16x32_bf16: ceiling 2.5 PFLOP/s / 6 = 417 TFLOP/s, 2.65x the fp32 matrix peak.
//
// Data movement (2 * NP bytes per element):
//   * activations live in HBM as (B, H, C/32, NP planes, 4 groups of 8 ch, W, 8) 16-bit tensors:
//     every (plane, group) of an image row is a contiguous run along x, so an LDS-DMA piece (64
//     consecutive halo pixels of one plane/group) and a lane group's epilogue stores are long
//     contiguous runs (measured: pixel-strided 16-byte pieces cost 17 % of the MFMA rate, contiguous
//     ones nothing); the producer's epilogue splits;
//   * the input halo of a 32-channel stage goes global -> LDS by buffer_load ... lds (LDS-DMA,
//     no VGPRs, out-of-frame slots read zeros through the descriptor's range check);
//   * weights never touch LDS: each wave loads its own pre-packed, pre-split 1 KB fragments
//     (16 couts x 32 k x 16 bit) straight from L2 into registers one tap ahead.
// Workgroup shapes (S3Cfg::NWM x NWN waves): 2 x 2 = 256 pixels x 64 couts (default); 1 x 4 = 256 pixels x 128
// couts, every wave covering all pixels for its own 32 couts (H2, long K, >= 128 couts: no two waves request the
// same weights and one halo feeds twice the MFMAs).  ONE LDS buffer and TWO workgroups per CU, so that staging /
// prologue / epilogue of one workgroup run beside the other's MFMAs; grids of at most 320 workgroups (one per CU
// at best) take the double-buffered variant (DB): two LDS buffers, one barrier per stage, the DMA of stage s+1
// issued inside the MFMA stream of stage s.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_epilogue.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr unsigned kOOB = 0xFFFFFFF0u;

// experiment knob (profiles/ab_bench.sh): wave priority inside the MFMA stages (s_setprio), 0 = leave it alone
#ifndef SFH_STAGE_PRIO
#define SFH_STAGE_PRIO 0
#endif

// NP_ = planes per operand: 3 = S3 tensors, bf16, six products per fp32 product ("bf16x6");
//                           2 = H2 tensors, fp16, three products ("f16x3", include/sfh_amd.h)
// NWN_ = waves along the cout axis: 2 = the 4-wave workgroup (256 pixels x 64 couts), 4 = the 8-wave workgroup
//        (256 pixels x 128 couts: the same halo feeds twice the MFMAs; one workgroup per CU, double-buffered)
// NWM_ = waves along the pixel axis: 2 = each wave owns half of the tile's pixel groups; 1 = every wave covers all 256
//        pixels for its own 32 couts (with NWN_ = 4: 4 waves, 256 pixels x 128 couts - no two waves of a workgroup
//        request the same weight fragments, and the halo feeds twice the MFMAs)
// STATS_ = the epilogue also leaves BatchNorm batch statistics (sfh_conv_desc.stats_partial): training-mode layers; a separate
//        instantiation, so that the inference instances keep their register counts
template <int KS_, int STRIDE_, int SH_, int SW_, int TH_, int TW_, int NP_ = 3, int NWN_ = 2, int NWM_ = 2, bool STATS_ = false>
struct S3Cfg {
  static constexpr int KS = KS_, STRIDE = STRIDE_, NP = NP_, NWN = NWN_, NWM = NWM_;
  static constexpr bool STATS = STATS_;
  static constexpr int NT = 64 * NWM * NWN;   // threads
  static constexpr int NCO = 32 * NWN;        // couts per workgroup
  static constexpr int SH = SH_, SW = SW_, TH = TH_, TW = TW_;
  static constexpr int PAD = KS / 2;          // padding before (3x3: 1, 1x1: 0, 4x4 stem: 2)
  static constexpr int PADA = (KS - 1) / 2;   // padding after  (3x3: 1, 1x1: 0, 4x4 stem: 1)
  static constexpr int NTAP = KS * KS;
  static constexpr int CKS = 32;  // channels per stage = k of one MFMA
  static constexpr int HH = (TH - 1) * STRIDE + KS;
  static constexpr int HW = (TW - 1) * STRIDE + KS;
  static constexpr int HPIX = HH * HW;
  static constexpr int HPIXP = (HPIX + 15) / 16 * 16;
  static constexpr int HSLOTS = 4 * NP * HPIXP;  // [NP planes][4 channel groups of 8][pixels] x 16 B
  static constexpr int NSL = (HSLOTS + NT - 1) / NT;
  static constexpr int BUF = NSL * NT;       // slots per LDS buffer (DMA rounds are whole)
  static constexpr int LDS_BYTES = 2 * BUF * 16;
  static constexpr int SUBX = TW / SW;
  static constexpr int NSUBT = (TH / SH) * SUBX;
  static constexpr int MT_M = NSUBT / NWM;   // pixel groups per wave
  static constexpr bool FLATROWS = (STRIDE == 1);
  static_assert(SH * SW == 16 && (NSUBT == 16 || NSUBT == 8), "tile = 8 or 16 pixel groups of 16");
};

struct S3Geom {
  int tiles_x, tiles_y, ntiles, nblk_n;
  int Ho, Wo, rows_total, rows_per_img;
  unsigned rows_magic;
  unsigned bytes0, bytes1;
  int nb_group;  // cout blocks of one pixel tile that run back to back on one XCD (1 .. nblk_n)
};

__device__ __forceinline__ bf16x8 as_bf(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f16x8 as_hf(const u32x4& v) { return __builtin_bit_cast(f16x8, v); }

// pad_y / pad_x: rows / columns before the output pixel covered by the window (C::PAD except for the
// 2x2 up-scatter conv, whose window position depends on the output quadrant)
template <class C>
__device__ __forceinline__ unsigned s3_halo_voffset(const sfh_conv_desc& d, const S3Geom& g, int which,
                                                    int slot, int r0, int x0, int pad_y, int pad_x) {
  if (slot >= C::HSLOTS) return kOOB;
  const int pl = slot / C::HPIXP, p = slot - pl * C::HPIXP;  // pl = plane*4 + channel group
  if (p >= C::HPIX) return kOOB;
  const int hy = p / C::HW, hx = p - hy * C::HW;
  int b, y;
  if (C::FLATROWS) {
    const int r = r0 - pad_y + hy;
    if (r < 0) return kOOB;
    b = (int)__umulhi((unsigned)r, g.rows_magic);
    y = r - b * g.rows_per_img;
    if (b >= d.batch || y >= d.H) return kOOB;
  } else {
    b = r0 >> 16;
    y = (r0 & 0xFFFF) * C::STRIDE - pad_y + hy;
    if (y < 0 || y >= d.H) return kOOB;
  }
  const int x = x0 * C::STRIDE - pad_x + hx;
  if (x < 0 || x >= d.W) return kOOB;
  // S3 / H2 layout (B, H, cs/32, NP, 4, W, 8): byte offset of (row, channel block 0, plane/group pl, x)
  unsigned rowi, xs_, ws_, nblk;
  if (which == 0) {
    if (C::KS == 2 && (y >= d.h0 || x >= d.w0)) return kOOB;  // fused Up with F.pad: the extra row / column reads 0
    rowi = (unsigned)(b * d.h0 + y);
    xs_ = (unsigned)x;
    ws_ = (unsigned)d.w0;
    nblk = (unsigned)d.cs0 >> 5;
  } else {
    const int ys = y - d.pad_top1, xs = x - d.pad_left1;
    if (ys < 0 || ys >= d.h1 || xs < 0 || xs >= d.w1) return kOOB;
    rowi = (unsigned)(b * d.h1 + ys);
    xs_ = (unsigned)xs;
    ws_ = (unsigned)d.w1;
    nblk = (unsigned)d.cs1 >> 5;
  }
  return ((rowi * nblk * (4u * C::NP) + (unsigned)pl) * ws_ + xs_) * 16u;
}

// All NSL halo slots of a thread at once (slot = tid + NT * i), same results as s3_halo_voffset: straight-line code -
// one validity predicate per slot instead of an early return per test (36 divergent branches in the prologue of a 3x3
// instance), and (plane/group, halo row, halo column) of slot i + 1 follow from slot i by constant increments instead
// of two divisions.  The prologue runs beside the co-resident workgroup's MFMAs at half the vector issue rate and was
// 14 % of a wave's life in the 64-channel layers (profiles/r03_diag_h2.txt).
template <class C>
__device__ __forceinline__ void s3_halo_offsets(const sfh_conv_desc& d, const S3Geom& g, int which, int tid, int r0,
                                                int x0, int pad_y, int pad_x, unsigned (&hoff)[C::NSL]) {
  constexpr int QP = C::NT / C::HPIXP, RP = C::NT % C::HPIXP;     // slot + NT: pl += QP, p += RP (carry below)
  constexpr int QH = RP / C::HW, RH = RP % C::HW;                 // p + RP:     hy += QH, hx += RH
  constexpr int QX = C::HPIXP / C::HW, RX = C::HPIXP % C::HW;     // p - HPIXP:  hy -= QX, hx -= RX
  int pl = tid / C::HPIXP, p = tid - pl * C::HPIXP;
  int hy = p / C::HW, hx = p - hy * C::HW;
  const int hsrc = which == 0 ? d.h0 : d.h1, wsrc = which == 0 ? d.w0 : d.w1;
  const int oy = which == 0 ? 0 : d.pad_top1, ox = which == 0 ? 0 : d.pad_left1;
  const unsigned nblk = (unsigned)(which == 0 ? d.cs0 : d.cs1) >> 5;
  const unsigned rowmul = nblk * (4u * C::NP);
  const int xbase = x0 * C::STRIDE - pad_x - ox;
  int ybase, bimg = 0;
  if (C::FLATROWS) {
    ybase = r0 - pad_y;
  } else {
    bimg = r0 >> 16;
    ybase = (r0 & 0xFFFF) * C::STRIDE - pad_y;
  }
#pragma unroll
  for (int i = 0; i < C::NSL; ++i) {
    bool ok = pl < 4 * C::NP && p < C::HPIX;
    int b, y;
    if (C::FLATROWS) {
      const int r = ybase + hy;
      b = (int)__umulhi((unsigned)r, g.rows_magic);
      y = r - b * g.rows_per_img;
      ok = ok && r >= 0 && b < d.batch && y < d.H;
    } else {
      b = bimg;
      y = ybase + hy;
      ok = ok && y >= 0 && y < d.H;
    }
    const int xs = xbase + hx;            // column inside the source (source 1: relative to its placement)
    const int x = xs + ox;                // column inside the conv frame
    ok = ok && x >= 0 && x < d.W;
    const int ys = y - oy;
    // source 0 of the fused Up conv (KS == 2) may be one row / column short of the frame; source 1 sits inside it
    if (which != 0 || C::KS == 2) ok = ok && ys >= 0 && ys < hsrc && xs >= 0 && xs < wsrc;
    const unsigned rowi = (unsigned)(b * hsrc + ys);
    const unsigned off = ((rowi * rowmul + (unsigned)pl) * (unsigned)wsrc + (unsigned)xs) * 16u;
    hoff[i] = ok ? off : kOOB;
    // next slot of this thread
    pl += QP;
    p += RP;
    hy += QH;
    hx += RH;
    if (hx >= C::HW) { hx -= C::HW; hy += 1; }
    if (p >= C::HPIXP) {
      p -= C::HPIXP;
      pl += 1;
      hy -= QX;
      hx -= RX;
      if (hx < 0) { hx += C::HW; hy -= 1; }
    }
  }
}

}  // namespace

namespace {

// DB = true : one workgroup per CU, two LDS buffers, the next stage's DMA rides inside the MFMA
//             stream, one barrier per stage - best for long K (>= 4 stages).
// DB = false: one LDS buffer, two workgroups per CU: stage DMA / prologue / epilogue of one
//             workgroup hide under the other's MFMAs - best for short K (64..128 channels, 1x1).
// Two-plane (H2) instances need 158 registers and 45 KB of LDS per workgroup, so three workgroups per CU would fit;
// measured (B=16, 640x360, two alternating runs on one device): 10.02-10.07 ms per batch for the DoubleConv
// launches with three, 9.96 ms with two - the register cap of 168 costs more than the third workgroup hides.
#ifndef SFH_H2_WD
#define SFH_H2_WD 2
#endif
#ifndef SFH_H2_XD
#define SFH_H2_XD 2
#endif
#ifndef SFH_W128_NWM
#define SFH_W128_NWM 1
#endif
#ifndef SFH_W8_MIN_BLOCKS
#define SFH_W8_MIN_BLOCKS 384
#endif
#ifndef SFH_H2_WAVES_PER_SIMD
#define SFH_H2_WAVES_PER_SIMD 2
#endif
template <class C, bool DB>
__global__ __launch_bounds__(C::NT, (C::NP == 2 && !DB) ? SFH_H2_WAVES_PER_SIMD : 2) void conv_s3_kernel(const sfh_conv_desc d,
                                                                                                     const S3Geom g) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  u32x4* const lds = reinterpret_cast<u32x4*>(smem_f);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  SFH_STAMP_INIT();
  SFH_CLOCK_BEGIN();

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / C::NWN, wn = wv % C::NWN;
  const int lq = lane & 15, lg = lane >> 4;

  // split-K (sfh_conv_desc.ksplit > 1): the grid holds ksplit copies of the (tile, cout block) space, copy ks
  // accumulates the stages [st0, st1) of the K loop and writes its own fp32 slab (sfh_splitk_finish adds them up)
  const int ksn = d.ksplit > 1 ? d.ksplit : 1;
  const int per_split = (int)(gridDim.x / (unsigned)ksn);
  const int ks = ksn > 1 ? (int)blockIdx.x / per_split : 0;
  const int bid = (int)blockIdx.x - ks * per_split;
  const int xcd = bid & 7, kk_ = bid >> 3;
  // within an XCD the pixel tiles of ONE cout block run back to back: the ~30 workgroups resident
  // on the XCD stream the same weight fragments, which then stay in its 4 MB L2 (weights are the
  // dominant L2->CU stream of this kernel: 12 KB per tap per workgroup)
  const int tpx = (g.ntiles + 7) >> 3;  // tiles per XCD
  // nb_group = G cout blocks of one pixel tile run back to back: their input tile is fetched into the
  // XCD's L2 once instead of G times.  G is chosen on the host so that the G weight sets still fit L2
  // next to it (transposed conv: G = all quadrants / couts, so the interleaved output rows are
  // completed in L2 before they are written back).
  const int per = tpx * g.nb_group;
  const int gi = kk_ / per, rr = kk_ - gi * per;
  const int nb = gi * g.nb_group + rr % g.nb_group;
  // each XCD owns a contiguous range of pixel tiles (whole tile rows), so vertically adjacent tiles
  // share their halo rows in that XCD's L2 (the round-robin order tile = tl * 8 + xcd measured 1.4x the L2 fills)
  // reverse_tiles: every XCD walks its range backwards, i.e. starts on what it wrote last in the previous launch
  const int tl = d.reverse_tiles ? tpx - 1 - rr / g.nb_group : rr / g.nb_group;
  const int tile = xcd * tpx + tl;
  if (tile >= g.ntiles) return;
  const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
  const int x0 = tx * C::TW;
  int r0;
  if (C::FLATROWS) {
    r0 = ty * C::TH;
  } else {
    const int img = ty / g.tiles_y;
    r0 = (img << 16) | ((ty - img * g.tiles_y) * C::TH);
  }
  const int n0 = nb * C::NCO;

  const int nst0 = d.c0 / C::CKS;
  const int nst1 = d.src1 ? d.c1 / C::CKS : 0;
  const int nst = nst0 + nst1;
  const int st0 = ks * nst / ksn, st1 = (ks + 1) * nst / ksn;   // this workgroup's stages (all of them without split-K)

  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.src0), 0, (int)g.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.src1 ? d.src1 : d.src0), 0, (int)(d.src1 ? g.bytes1 : 0u), 0x00020000);
  // packed weights of this cout block: [stage][tap][plane NP][cout group 4][lane 64][8 x 16 bit]
  constexpr int NP = C::NP;
  constexpr unsigned WTAP = (unsigned)NP * 4u * 1024u;  // bytes per (stage, tap)
  const unsigned wtotal = (unsigned)nst * C::NTAP * WTAP;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(d.wpacked)) + (size_t)(nb * (C::NWN / 2) + (wn >> 1)) * wtotal, 0,
      (int)wtotal, 0x00020000);   // weights are packed per 64 couts: waves (wn >> 1) share a block
  const unsigned wvoff = lane * 16u + (unsigned)(2 * (wn & 1)) * 1024u;

  // byte offsets of this thread's halo slots in the CURRENT source (recomputed once when the
  // stage loop crosses from source 0 to source 1)
  // 2x2 up-scatter conv: quadrant (dy,dx) of this cout block reads rows y + dy - 1 .. y + dy
  int pad_y = C::PAD, pad_x = C::PAD;
  if (C::KS == 2) {
    const int qd = n0 / (d.cout >> 2);
    pad_y = 1 - (qd >> 1);
    pad_x = 1 - (qd & 1);
  }
  unsigned hoff[C::NSL];
  s3_halo_offsets<C>(d, g, 0, tid, r0, x0, pad_y, pad_x, hoff);

  // LDS-DMA piece i (64 slots of this wave) of stage st into buffer b
  auto dma_piece = [&](int st, int b, int i) {
    u32x4* const hb = lds + b * C::BUF;
    const bool first = st < nst0;
    // stage = one 32-channel block = 4 * NP (plane, group) runs of W x 16 bytes of the row
    constexpr unsigned SB = 64u * (unsigned)NP;
    const unsigned cb = first ? (unsigned)st * (SB * (unsigned)d.w0) : (unsigned)(st - nst0) * (SB * (unsigned)d.w1);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? rs0 : rs1, (lds_ptr_t)(hb + wv * 64 + C::NT * i), 16,
                                             (int)hoff[i], (int)cb, 0, 0);
  };
  auto dma_stage = [&](int st, int b) {
#pragma unroll
    for (int i = 0; i < C::NSL; ++i) dma_piece(st, b, i);
  };

  // weight fragments of one tap for this wave: [plane][cout group of the wave]
  auto load_w = [&](u32x4 (&w)[NP][2], unsigned soff) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        w[p][ni] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (int)(wvoff + ni * 1024u), (int)(soff + p * 4096u), 0);
  };

  // LDS slot of this lane's pixel in pixel group 0 of the wave; group mi adds a compile-time
  // offset (the wave's MT_M groups start on a tile-row boundary), so every operand read is
  // `base + immediate`
  const int pixbase0 = lg * C::HPIXP +
                       ((wm * C::MT_M / C::SUBX) * C::SH + lq / C::SW) * C::STRIDE * C::HW +
                       (lq % C::SW) * C::STRIDE;
  static_assert(C::MT_M % C::SUBX == 0, "a wave's pixel groups must start on a tile-row boundary");

  f32x4 acc[2][C::MT_M];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < C::MT_M; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // sfh_conv_desc.acc_init: the accumulators start from an fp32 NHWC tensor in ACCUMULATOR units (the partial of the
  // other half of a fused Up block, written by its producer divided by this layer's scale) instead of zero.  The sixteen
  // loads are requested here and land under the first stage's DMA wait; added as a residual in the epilogue they were
  // sixteen load -> wait -> use round trips at the END of the workgroup (49 % of a wave's life at u4.skip).
  if constexpr (C::KS == 3 && C::STRIDE == 1) {
    if (d.acc_init) {
      const __amdgpu_buffer_rsrc_t ri =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.acc_init), 0, (int)kOOB, 0x00020000);
      const unsigned cs_i = (unsigned)d.cout;                        // channel stride of acc_init = all couts of the layer
      const unsigned c_lane_i = (unsigned)(n0 + 32 * wn + 4 * lg);
#pragma unroll
      for (int mi = 0; mi < C::MT_M; ++mi) {
        const int sgi = wm * C::MT_M + mi;
        const int sy = sgi / C::SUBX, sx = sgi - sy * C::SUBX;
        const int oy = sy * C::SH + lq / C::SW, ox = sx * C::SW + lq % C::SW;
        const int x = x0 + ox, r = r0 + oy;
        const int b = (int)__umulhi((unsigned)r, g.rows_magic);
        const int y = r - b * g.rows_per_img;
        const bool ok = x < g.Wo && r < g.rows_total && y < g.Ho;
        const unsigned off = ok ? (((unsigned)(b * g.Ho + y) * (unsigned)g.Wo + (unsigned)x) * cs_i + c_lane_i) * 4u : kOOB;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[ni][mi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ri, (int)off, ni * 64, 0));
      }
    }
  }

  // weight fragments: a ring of WD register sets, tap t of a stage lives in set (R0 + t) % WD and the set that
  // tap t - 1 leaves is refilled with tap t + WD - 1.  Two sets (one tap ahead) are the default for both formats.
  // Measured for the two-plane format, whose taps are half as long (B=16, 640x360, DoubleConv launches, two
  // alternating runs on one device): WD 2 / XD 2 10.03-10.15 ms, WD 3 / XD 2 10.64-10.66, WD 3 / XD 3 10.58,
  // WD 2 / XD 3 10.01-10.07: neither the weight stream from L2 nor the operand reads are what the MFMAs wait for.
  constexpr int WD = (NP == 2 && C::NTAP % 3 == 0) ? SFH_H2_WD : 2;
  constexpr int XD = NP == 2 ? SFH_H2_XD : 2;   // operand reads are issued XD steps ahead of their use
  u32x4 wr[WD][NP][2];
  unsigned wsoff = (unsigned)st0 * C::NTAP * WTAP;   // byte offset of the NEXT (stage, tap) fragment set
  const unsigned wlast = wtotal - WTAP;
#pragma unroll
  for (int i = 0; i < WD - 1; ++i) {
    load_w(wr[i], wsoff);
    wsoff = (wsoff + WTAP <= wlast) ? wsoff + WTAP : wlast;
  }

  // One stage = NTAP taps x MT_M pixel groups; per (tap, group): 3 operand reads + 12 MFMAs.
  // The reads of step s+1 are issued behind the first MFMAs of step s (two register sets,
  // static indices), the next tap's weight fragments one tap ahead, the next stage's DMA batch
  // behind the first steps of the stage.  SW selects which weight register set holds tap 0
  // (NTAP is odd, so the sets swap roles every stage).
  auto stage = [&](int st, int cur, auto r0_tag) {
    constexpr int R0 = decltype(r0_tag)::value;   // ring position of tap 0 in this stage
    constexpr int NSTEP = C::NTAP * C::MT_M;
    const u32x4* const halo = lds + cur * C::BUF;
    // operand ring of three register sets: the reads of step s+2 are issued in step s, i.e. two
    // steps (~380 cycles of MFMA) ahead of their use - one step is not enough to cover the LDS
    // latency with four waves reading (measured: 358 cycles per 192-cycle step before)
    u32x4 xq[XD + 1][NP];
    auto ld_x = [&](int s, int buf) {
      const int t = s / C::MT_M, mi = s % C::MT_M;
      const int toff = (t / C::KS) * C::HW + (t % C::KS);
      const int moff = (mi / C::SUBX) * C::SH * C::STRIDE * C::HW + (mi % C::SUBX) * C::SW * C::STRIDE;
#pragma unroll
      for (int p = 0; p < NP; ++p) xq[buf][p] = halo[pixbase0 + (p * 4 * C::HPIXP + moff + toff)];
    };
#pragma unroll
    for (int i = 0; i < XD; ++i)
      if (i < NSTEP) ld_x(i, i);
    const int stn = st + 1 < st1 ? st + 1 : st;  // unconditional DMA: no branch in the MFMA block
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      const int t = s / C::MT_M, mi = s % C::MT_M;
      const int xb = s % (XD + 1);
      u32x4 (&wc)[NP][2] = wr[(R0 + t) % WD];
      u32x4 (&wnx)[NP][2] = wr[(R0 + t + WD - 1) % WD];
      if (s + XD < NSTEP) ld_x(s + XD, (s + XD) % (XD + 1));
      if (mi == 0) {
        load_w(wnx, wsoff);  // next tap (or tap 0 of the next stage): one tap of MFMAs ahead
        wsoff = (wsoff + WTAP <= wlast) ? wsoff + WTAP : wlast;
      }
      // the next stage's LDS-DMA pieces, spread over the first steps
      // (8-wave workgroup: all of them in step 0, right BEHIND the weight request of that step - vector loads
      //  return in order, so a weight fragment requested behind a DMA piece waits for that piece)
      constexpr int PPS = C::NT == 512 ? C::NSL : (C::NSL + NSTEP - 1) / NSTEP;
      if (DB) {
#pragma unroll
        for (int q = 0; q < PPS; ++q)
          if (s * PPS + q < C::NSL) dma_piece(stn, cur ^ 1, s * PPS + q);
      }
      // the kept partial products (six of 3 x 3 bf16 planes, three of 2 x 2 fp16 planes), smallest first; the
      // two cout groups are interleaved so that consecutive MFMAs never depend on each other
      constexpr int NPROD = NP == 3 ? 6 : 3;
      constexpr int PW[6] = {0, 1, NP == 3 ? 2 : 0, 0, 1, 0}, PX[6] = {NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0, 0};
#pragma unroll
      for (int k6 = 0; k6 < NPROD; ++k6)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if constexpr (NP == 3)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(wc[PW[k6]][ni]), as_bf(xq[xb][PX[k6]]),
                                                                 acc[ni][mi], 0, 0, 0);
          else
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_hf(wc[PW[k6]][ni]), as_hf(xq[xb][PX[k6]]),
                                                                acc[ni][mi], 0, 0, 0);
        }
      // keep each step's memory instructions inside the step, operand reads of the next step
      // right behind the first MFMA so that the step's other MFMAs cover their LDS latency
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NPROD - 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // odd tap counts (1, 9) swap the roles of the two weight register sets every stage, even ones (16) do not
  constexpr int SWAPS = C::NTAP % WD;   // ring position of tap 0 in odd stages (2 * NTAP = 0 mod WD)
  static_assert((2 * C::NTAP) % WD == 0, "the stage loop is unrolled by two");

  dma_stage(st0, 0);
  if (SFH_STAGE_PRIO) __builtin_amdgcn_s_setprio(SFH_STAGE_PRIO);
  auto maybe_switch = [&](int st) {  // the DMA issued during stage st targets stage st+1
    if (d.src1 && st + 1 == nst0) {
      s3_halo_offsets<C>(d, g, 1, tid, r0, x0, pad_y, pad_x, hoff);
    }
  };
  // hipcc's own wait before the barrier covers only part of the outstanding LDS-DMA (observed:
  // s_waitcnt vmcnt(6) with DMA pieces younger than that in flight -> stale halo data in stages
  // >= 2), so the drain is explicit: every DMA piece of this wave has landed, then the barrier.
  auto stage_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  if (DB) {
    for (int st = st0; st < st1; st += 2) {
      maybe_switch(st);
      stage_barrier();  // stage st landed for every wave; the other buffer is free
      stage(st, 0, std::integral_constant<int, 0>{});
      if (st + 1 < st1) {
        maybe_switch(st + 1);
        stage_barrier();
        stage(st + 1, 1, std::integral_constant<int, SWAPS>{});
      }
    }
    // the last stage re-issues its own DMA into the other buffer (branch-free MFMA block): it must
    // have landed before the wave ends, or it would write into the LDS of the workgroup that
    // inherits this allocation
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // single buffer: [DMA stage st] [drain + barrier] [compute] [barrier: buffer free again]
    // (diag build: segments 8 prologue, 9 DMA wait + barrier, 10 MFMA stage, 11 buffer-free barrier + DMA issue, 12 epilogue)
    SFH_STAMP(0);
    for (int st = st0; st < st1; st += 2) {
      stage_barrier();
      SFH_STAMP(1);
      stage(st, 0, std::integral_constant<int, 0>{});
      SFH_STAMP(2);
      if (st + 1 < st1) {
        if (d.src1 && st + 1 == nst0) {
          s3_halo_offsets<C>(d, g, 1, tid, r0, x0, pad_y, pad_x, hoff);
        }
        __syncthreads();
        dma_stage(st + 1, 0);
        SFH_STAMP(3);
        stage_barrier();
        SFH_STAMP(1);
        stage(st + 1, 0, std::integral_constant<int, SWAPS>{});
        SFH_STAMP(2);
      }
      if (st + 2 < st1) {
        if (d.src1 && st + 2 == nst0) {
          s3_halo_offsets<C>(d, g, 1, tid, r0, x0, pad_y, pad_x, hoff);
        }
        __syncthreads();
        dma_stage(st + 2, 0);
        SFH_STAMP(3);
      }
    }
  }
  if (SFH_STAGE_PRIO) __builtin_amdgcn_s_setprio(0);
  sfh_conv_epilogue<C, 2, C::MT_M, (NP == 3 ? 1 : 2)>(d, g, acc, n0 + 32 * wn, wm * C::MT_M, r0, x0, lq, lg,
                                                      (size_t)ks * (size_t)d.ksplit_stride,
                                                      (unsigned)(tile * C::NWM + wm));
  if (!DB) {
    SFH_STAMP(4);
    SFH_CLOCK_END();
    SFH_STAMP_FLUSH_AT(8);
  }

  // ---- OutConv fused behind the last conv (unet/unet_parts.py:74-77): acc now holds the activated outputs
  if constexpr (C::KS == 3 && C::STRIDE == 1 && C::NWN == 2) {
    if (d.head_w) {
      float* const hl = smem_f;  // [2 cout halves][NSUBT groups][16 pixels][8 classes]
      const int nc = d.head_nc;
      __syncthreads();           // every wave is past its last operand read: the stage buffer is free
      for (int k = 0; k < nc; ++k) {
        f32x4 wk[2];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) wk[ni] = *reinterpret_cast<const f32x4*>(d.head_w + k * 64 + 32 * wn + ni * 16 + 4 * lg);
#pragma unroll
        for (int mi = 0; mi < C::MT_M; ++mi) {
          float p = 0.f;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int j = 0; j < 4; ++j) p += acc[ni][mi][j] * wk[ni][j];
          p += __shfl_xor(p, 16);
          p += __shfl_xor(p, 32);
          if (lg == 0) hl[((wn * C::NSUBT + wm * C::MT_M + mi) * 16 + lq) * 8 + k] = p;
        }
      }
      __syncthreads();
      // one thread per pixel of the tile (NSUBT * 16 <= 256)
      if (tid < C::NSUBT * 16) {
        const int s = tid >> 4, q = tid & 15;
        const int sy = s / C::SUBX, sx = s - sy * C::SUBX;
        const int oy = sy * C::SH + q / C::SW, ox = sx * C::SW + q % C::SW;
        const int x = x0 + ox, r = r0 + oy;
        const int b = (int)__umulhi((unsigned)r, g.rows_magic);
        const int y = r - b * g.rows_per_img;
        if (x < g.Wo && r < g.rows_total && y < g.Ho) {
          float lgt[8];
#pragma unroll
          for (int k = 0; k < 8; ++k)
            lgt[k] = k < nc ? hl[(s * 16 + q) * 8 + k] + hl[((C::NSUBT + s) * 16 + q) * 8 + k] + d.head_b[k] : 0.f;
          const long hw = (long)g.Ho * g.Wo, pix = (long)y * g.Wo + x;
          for (int k = 0; k < nc; ++k) d.head_logits[((long)b * nc + k) * hw + pix] = lgt[k];
          if (d.head_stn) {
            const f32x4 fr = *reinterpret_cast<const f32x4*>(d.head_frame + ((long)b * hw + pix) * 4);
            float v8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < nc; ++k) v8[k] = lgt[k];
            for (int k = 0; k < 3 && nc + k < 8; ++k) v8[nc + k] = fr[k];
            float* o = d.head_stn + ((long)b * hw + pix) * 8;
            *reinterpret_cast<f32x4*>(o) = (f32x4){v8[0], v8[1], v8[2], v8[3]};
            *reinterpret_cast<f32x4*>(o + 4) = (f32x4){v8[4], v8[5], v8[6], v8[7]};
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------ weight packing (split)
// packed[nb][stage][tap][plane NP][cout group 4][lane 64][j 8] bf16 (NP = 3) or fp16 planes of w * wscale (NP = 2)
//   cout = nb*64 + ng*16 + (lane&15);  channel-in-source = stage_local*32 + 8*(lane>>4) + j
template <int NP>
__global__ void pack_s3_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed,
                                       int ks, int c0, int c1, int coutv, int transposed, int aux, long total,
                                       float wscale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one (lane, j-octet, all planes)
  if (idx >= total) return;
  const int ntap = ks * ks;
  const int nst0 = c0 / 32, nst1 = c1 / 32, nst = nst0 + nst1;
  long r = idx;
  const int lane = r & 63; r >>= 6;
  const int ng = r & 3; r >>= 2;
  const int tap = r % ntap; r /= ntap;
  const int st = r % nst;
  const int nb = r / nst;
  const int ky = tap / ks, kx = tap % ks;
  const int cv = nb * 64 + ng * 16 + (lane & 15);
  const int cin_total = c0 + c1;
  const long base = ((((long)nb * nst + st) * ntap + tap) * NP) * 4096 + (long)ng * 1024 + lane * 16;  // bytes / 1
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int cl = (st < nst0 ? st : st - nst0) * 32 + 8 * (lane >> 4) + j;
    const int cin = st < nst0 ? cl : c0 + cl;
    float v;
    if (transposed == 2) {  // 7x7 s2 stem as a 4x4 conv over the 2x2 space-to-depth input (aux = real cin)
      const int csd = c0 >> 2;
      const int par = cl / csd, c = cl - par * csd;
      const int ky7 = 2 * ky + (par >> 1) - 1, kx7 = 2 * kx + (par & 1) - 1;
      v = (c < aux && ky7 >= 0 && ky7 <= 6 && kx7 >= 0 && kx7 <= 6)
              ? w[(((size_t)cv * aux + c) * 7 + ky7) * 7 + kx7] : 0.f;
    } else if (transposed == 3) {  // backward-data of Conv2d (see sfh_pack_conv_weights mode 3)
      v = cv < aux ? w[(((size_t)cin * aux + cv) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)] : 0.f;
    } else if (transposed == 4) {  // backward-data of ConvTranspose2d (mode 4): aux = orig cout
      const int qd = cin / aux, co = cin - qd * aux;
      v = w[(((size_t)cv * aux + co) * 2 + (qd >> 1)) * 2 + (qd & 1)];
    } else if (transposed) {
      const int cout = coutv >> 2;
      const int qd = cv / cout, co = cv - qd * cout;
      v = w[(((size_t)cin * cout + co) * 2 + (qd >> 1)) * 2 + (qd & 1)];
    } else {
      v = w[(((size_t)cv * cin_total + cin) * ks + ky) * ks + kx];
    }
    if constexpr (NP == 3) {
      const __bf16 v0 = (__bf16)v;
      const float r1 = v - (float)v0;
      const __bf16 v1 = (__bf16)r1;
      const __bf16 v2 = (__bf16)(r1 - (float)v1);
      packed[(base + 0 * 4096) / 2 + j] = __builtin_bit_cast(unsigned short, v0);
      packed[(base + 1 * 4096) / 2 + j] = __builtin_bit_cast(unsigned short, v1);
      packed[(base + 2 * 4096) / 2 + j] = __builtin_bit_cast(unsigned short, v2);
    } else {
      const float u = fminf(fmaxf(v * wscale, -65504.f), 65504.f);
      const _Float16 h0 = (_Float16)u;
      const _Float16 h1 = (_Float16)(u - (float)h0);
      packed[(base + 0 * 4096) / 2 + j] = __builtin_bit_cast(unsigned short, h0);
      packed[(base + 1 * 4096) / 2 + j] = __builtin_bit_cast(unsigned short, h1);
    }
  }
}

// Same packing for the two layouts a training step re-packs after every optimizer update (mode 0: forward
// OIHW, mode 3: backward-data = channels swapped + taps flipped), 3x3 and 1x1: one thread = (cout, 8 cins) for
// ALL taps, so that it reads whole contiguous runs of the checkpoint tensor (72 floats in mode 0, nine floats
// per cin in mode 3) instead of one float every 36 bytes (the per-tap kernel above moved 0.5 TB/s).
template <int KS, int MODE, int NP>
__global__ __launch_bounds__(256) void pack_s3_weights_alltaps_kernel(const float* __restrict__ w,
                                                                      unsigned short* __restrict__ packed, int c0, int c1,
                                                                      int coutv, int aux, long total, float wscale) {
  constexpr int NT = KS * KS;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (nb, st, ng, lane)
  if (idx >= total) return;
  const int nst0 = c0 / 32, nst = nst0 + c1 / 32;
  long r = idx;
  const int lane = r & 63; r >>= 6;
  const int ng = r & 3; r >>= 2;
  const int st = r % nst;
  const int nb = r / nst;
  const int cv = nb * 64 + ng * 16 + (lane & 15);
  const int cin_total = c0 + c1;
  const int cl = (st < nst0 ? st : st - nst0) * 32 + 8 * (lane >> 4);
  const int cin0 = st < nst0 ? cl : c0 + cl;
  float v[8][NT];
  if (MODE == 0) {
    const float* src = w + ((size_t)cv * cin_total + cin0) * NT;     // 8 * NT contiguous floats, 16-byte aligned
#pragma unroll
    for (int q = 0; q < 2 * NT; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[(4 * q + e) / NT][(4 * q + e) % NT] = t[e];
    }
  } else {  // backward-data: w[cin][cv][flipped tap], zero rows for cv >= aux (padded output channels)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* src = w + ((size_t)(cin0 + j) * aux + cv) * NT;
#pragma unroll
      for (int t = 0; t < NT; ++t) v[j][t] = cv < aux ? src[NT - 1 - t] : 0.f;
    }
  }
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    u16x8 p0, p1, p2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = v[j][t];
      if constexpr (NP == 3) {
        const __bf16 v0 = (__bf16)x;
        const float r1 = x - (float)v0;
        const __bf16 v1 = (__bf16)r1;
        const __bf16 v2 = (__bf16)(r1 - (float)v1);
        p0[j] = __builtin_bit_cast(unsigned short, v0);
        p1[j] = __builtin_bit_cast(unsigned short, v1);
        p2[j] = __builtin_bit_cast(unsigned short, v2);
      } else {   // two fp16 planes of w * wscale (as pack_s3_weights_kernel<2>)
        const float u = fminf(fmaxf(x * wscale, -65504.f), 65504.f);
        const _Float16 h0 = (_Float16)u;
        const _Float16 h1 = (_Float16)(u - (float)h0);
        p0[j] = __builtin_bit_cast(unsigned short, h0);
        p1[j] = __builtin_bit_cast(unsigned short, h1);
      }
    }
    const long base = ((((long)nb * nst + st) * NT + t) * NP) * 4096 + (long)ng * 1024 + lane * 16;  // bytes
    *reinterpret_cast<u16x8*>(packed + (base + 0 * 4096) / 2) = p0;
    *reinterpret_cast<u16x8*>(packed + (base + 1 * 4096) / 2) = p1;
    if constexpr (NP == 3) *reinterpret_cast<u16x8*>(packed + (base + 2 * 4096) / 2) = p2;
  }
}

// fp32 NHWC (B,H,W,cs) -> S3 (B,H,W,3,cs) and back (tests, network input, debugging)
// element (row, x, c, plane) of an S3 tensor (rows, cs/32, 3, 4, W, 8)
__device__ __forceinline__ long s3_elem(long row, int x, int c, int plane, int W, int cs) {
  return ((((row * (cs >> 5) + (c >> 5)) * 3 + plane) * 4 + ((c & 31) >> 3)) * W + x) * 8 + (c & 7);
}

// one thread = 8 channels (one group) of one pixel; a wave covers 16 pixels x the 4 groups of one
// 32-channel block: reads are whole 128-byte lines, writes 256 contiguous bytes per (plane, group)
__global__ __launch_bounds__(256) void f32_to_s3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                        int W, int cs, int xchunks, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int g = (int)(i & 3), px = (int)((i >> 2) & 15);
  long r = i >> 6;
  const int xc = (int)(r % xchunks); r /= xchunks;
  const int cb = (int)(r % (cs >> 5));
  const long row = r / (cs >> 5);
  const int x = xc * 16 + px;
  if (x >= W) return;
  const float* sp = src + (row * W + x) * cs + cb * 32 + g * 8;
  const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
  const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  u16x8 p0, p1, p2;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 v0 = (__bf16)v[j];
    const float r1 = v[j] - (float)v0;
    const __bf16 v1 = (__bf16)r1;
    const __bf16 v2 = (__bf16)(r1 - (float)v1);
    p0[j] = __builtin_bit_cast(unsigned short, v0);
    p1[j] = __builtin_bit_cast(unsigned short, v1);
    p2[j] = __builtin_bit_cast(unsigned short, v2);
  }
  const long e = s3_elem(row, x, cb * 32 + g * 8, 0, W, cs);
  const long ps = 4L * W * 8;  // plane stride in elements
  *reinterpret_cast<u16x8*>(dst + e) = p0;
  *reinterpret_cast<u16x8*>(dst + e + ps) = p1;
  *reinterpret_cast<u16x8*>(dst + e + 2 * ps) = p2;
}

__global__ void s3_to_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, int W, int cs,
                                 long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long pix = i / cs;
  const int c = i - pix * cs;
  const long row = pix / W;
  const int x = pix - row * W;
  float v = 0.f;
#pragma unroll
  for (int p = 2; p >= 0; --p) v += __builtin_bit_cast(float, (unsigned)src[s3_elem(row, x, c, p, W, cs)] << 16);
  dst[i] = v;
}

// fp32 NHWC <-> H2 (rows, cs/32, 2, 4, W, 8) fp16 (format: include/sfh_amd.h); thread mapping as f32_to_s3_kernel
// (no early exit: every lane reaches sfh_h2_report)
__global__ __launch_bounds__(256) void f32_to_h2_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                                        int W, int cs, int xchunks, long total, float scale,
                                                        unsigned* overflow, unsigned* range) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int g = (int)(i & 3), px = (int)((i >> 2) & 15);
  long r = i >> 6;
  const int xc = (int)(r % xchunks); r /= xchunks;
  const int cb = (int)(r % (cs >> 5));
  const long row = r / (cs >> 5);
  const int x = xc * 16 + px;
  unsigned over = 0u;
  if (i < total && x < W) {
    const float* sp = src + (row * W + x) * cs + cb * 32 + g * 8;
    const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
    sfh_u32x2 pa[2], pb[2];
    sfh_split4_h2(a, scale, pa, over);
    sfh_split4_h2(b, scale, pb, over);
    const long e = ((((row * (cs >> 5) + cb) * 2) * 4 + g) * W + x) * 8;
    const long ps = 4L * W * 8;  // plane stride in elements
    *reinterpret_cast<u32x4*>(dst + e) = (u32x4){pa[0][0], pa[0][1], pb[0][0], pb[0][1]};
    *reinterpret_cast<u32x4*>(dst + e + ps) = (u32x4){pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
  }
  sfh_h2_report(over, overflow, range);
}

// MaxPool2d(3, stride 2, padding 1) of an fp32 NHWC tensor written straight into a split tensor (ResNetSTN's stem pooling,
// models/resnet.py:176,243: the pooled tensor is the first one that enters the split domain).  Same values as
// maxpool3x3s2_kernel followed by f32_to_h2 / f32_to_s3 - the split is monotone, the maximum is taken in fp32 - in one pass
// and one launch less.  Thread mapping as f32_to_s3_kernel (8 channels of one output pixel); no early exit (sfh_h2_report).
template <int DFMT>
__global__ __launch_bounds__(256) void maxpool3x3s2_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ dst,
                                                                 int H, int W, int Ho, int Wo, int cs, int xchunks, long total,
                                                                 float scale, unsigned* overflow, unsigned* range) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int g = (int)(i & 3), px = (int)((i >> 2) & 15);
  long r = i >> 6;
  const int xc = (int)(r % xchunks); r /= xchunks;
  const int cb = (int)(r % (cs >> 5));
  const long row = r / (cs >> 5);          // b * Ho + yo
  const int xo = xc * 16 + px;
  unsigned over = 0u;
  if (i < total && xo < Wo) {
    const long b = row / Ho;
    const int yo = (int)(row - b * Ho);
    f32x4 m0 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, m1 = m0;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = 2 * yo - 1 + dy;
      if (yy < 0 || yy >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = 2 * xo - 1 + dx;
        if (xx < 0 || xx >= W) continue;
        const float* sp = x + ((b * H + yy) * W + xx) * cs + cb * 32 + g * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(sp), c = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          m0[j] = sfh_max_nan(m0[j], a[j]);
          m1[j] = sfh_max_nan(m1[j], c[j]);
        }
      }
    }
    const long ps = 4L * Wo * 8;  // plane stride in elements
    if constexpr (DFMT == SFH_FMT_H2) {
      sfh_u32x2 pa[2], pb[2];
      sfh_split4_h2(m0, scale, pa, over);
      sfh_split4_h2(m1, scale, pb, over);
      const long e = ((((row * (cs >> 5) + cb) * 2) * 4 + g) * Wo + xo) * 8;
      *reinterpret_cast<u32x4*>(dst + e) = (u32x4){pa[0][0], pa[0][1], pb[0][0], pb[0][1]};
      *reinterpret_cast<u32x4*>(dst + e + ps) = (u32x4){pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
    } else {
      sfh_u32x2 pa[3], pb[3];
      sfh_split4(m0, pa);
      sfh_split4(m1, pb);
      const long e = s3_elem(row, xo, cb * 32 + g * 8, 0, Wo, cs);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + e + p * ps) = (u32x4){pa[p][0], pa[p][1], pb[p][0], pb[p][1]};
    }
  }
  if constexpr (DFMT == SFH_FMT_H2) sfh_h2_report(over, overflow, range);
}

__global__ void h2_to_f32_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, int W, int cs,
                                 long total, float inv_scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long pix = i / cs;
  const int c = i - pix * cs;
  const long row = pix / W;
  const int x = pix - row * W;
  const long e = ((((row * (cs >> 5) + (c >> 5)) * 2) * 4 + ((c & 31) >> 3)) * W + x) * 8 + (c & 7);
  const float lo = (float)__builtin_bit_cast(_Float16, src[e + 4L * W * 8]);
  const float hi = (float)__builtin_bit_cast(_Float16, src[e]);
  dst[i] = (lo + hi) * inv_scale;
}

// ------------------------------------------------------------------ split-K: second half
// y = [relu](sum_k slab_k + shift + residual) -> F32 NHWC / S3 / H2; thread mapping as f32_to_s3_kernel (one thread =
// 8 channels of one pixel); no early exit (sfh_h2_report reads every lane)
template <int DFMT>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slabs, int nslabs, long slab_elems,
                                                            const float* __restrict__ shift, const void* residual,
                                                            int res_fmt, float res_inv, int relu, int W, int cs, int xchunks,
                                                            long total, void* dstv, float dst_scale,
                                                            unsigned* overflow, unsigned* range) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int g = (int)(i & 3), px = (int)((i >> 2) & 15);
  long r = i >> 6;
  const int xc = (int)(r % xchunks); r /= xchunks;
  const int nb32 = cs >> 5;
  const int cb = (int)(r % nb32);
  const long row = r / nb32;
  const int x = xc * 16 + px;
  unsigned over = 0u;
  if (i < total && x < W) {
    const int c0 = cb * 32 + g * 8;
    const long e32 = (row * W + x) * cs + c0;
    float v[8];
    {
      const f32x4 a = *reinterpret_cast<const f32x4*>(shift + c0), b = *reinterpret_cast<const f32x4*>(shift + c0 + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    for (int k = 0; k < nslabs; ++k) {
      const float* sp = slabs + (long)k * slab_elems + e32;
      const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
      v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3]; v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
    }
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    if (residual) {
      if (res_fmt == SFH_FMT_F32) {
        const float* rp = reinterpret_cast<const float*>(residual) + e32;
        const f32x4 a = *reinterpret_cast<const f32x4*>(rp), b = *reinterpret_cast<const f32x4*>(rp + 4);
        v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3]; v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
      } else {
        const int np = res_fmt == SFH_FMT_H2 ? 2 : 3;
        const unsigned short* rp = reinterpret_cast<const unsigned short*>(residual) +
                                   ((((row * nb32 + cb) * np) * 4 + g) * W + x) * 8;
        const long ps = 4L * W * 8;
        float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int p = np - 1; p >= 0; --p) {   // smallest plane first, as the conv epilogue adds them
          const u32x4 w = *reinterpret_cast<const u32x4*>(rp + p * ps);   // 8 x 16 bit
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            if (res_fmt == SFH_FMT_H2) {
              const sfh_f32x2 f = sfh_unpack_h(w[h]);
              q[2 * h] += f[0];
              q[2 * h + 1] += f[1];
            } else {
              q[2 * h] += __builtin_bit_cast(float, w[h] << 16);
              q[2 * h + 1] += __builtin_bit_cast(float, w[h] & 0xFFFF0000u);
            }
          }
        }
        const float rs_ = res_fmt == SFH_FMT_H2 ? res_inv : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += q[j] * rs_;
      }
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = sfh_relu(v[j]);
    }
    if constexpr (DFMT == SFH_FMT_F32) {
      float* dp = reinterpret_cast<float*>(dstv) + e32;
      *reinterpret_cast<f32x4*>(dp) = (f32x4){v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(dp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
    } else if constexpr (DFMT == SFH_FMT_H2) {
      unsigned short* dst = reinterpret_cast<unsigned short*>(dstv);
      sfh_u32x2 pa[2], pb[2];
      sfh_split4_h2((f32x4){v[0], v[1], v[2], v[3]}, dst_scale, pa, over);
      sfh_split4_h2((f32x4){v[4], v[5], v[6], v[7]}, dst_scale, pb, over);
      const long e = ((((row * nb32 + cb) * 2) * 4 + g) * W + x) * 8;
      const long ps = 4L * W * 8;
      *reinterpret_cast<u32x4*>(dst + e) = (u32x4){pa[0][0], pa[0][1], pb[0][0], pb[0][1]};
      *reinterpret_cast<u32x4*>(dst + e + ps) = (u32x4){pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
    } else {
      unsigned short* dst = reinterpret_cast<unsigned short*>(dstv);
      u16x8 p0, p1, p2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const __bf16 v0 = (__bf16)v[j];
        const float r1 = v[j] - (float)v0;
        const __bf16 v1 = (__bf16)r1;
        const __bf16 v2 = (__bf16)(r1 - (float)v1);
        p0[j] = __builtin_bit_cast(unsigned short, v0);
        p1[j] = __builtin_bit_cast(unsigned short, v1);
        p2[j] = __builtin_bit_cast(unsigned short, v2);
      }
      const long e = s3_elem(row, x, c0, 0, W, cs);
      const long ps = 4L * W * 8;
      *reinterpret_cast<u16x8*>(dst + e) = p0;
      *reinterpret_cast<u16x8*>(dst + e + ps) = p1;
      *reinterpret_cast<u16x8*>(dst + e + 2 * ps) = p2;
    }
  }
  if constexpr (DFMT == SFH_FMT_H2) sfh_h2_report(over, overflow, range);
}

template <class C, bool DB>
int launch_s3(const sfh_conv_desc& d, hipStream_t stream) {
  S3Geom g;
  g.Ho = (d.H + C::PAD + C::PADA - C::KS) / C::STRIDE + 1;
  g.Wo = (d.W + C::PAD + C::PADA - C::KS) / C::STRIDE + 1;
  g.tiles_x = sfh_cdiv(g.Wo, C::TW);
  if (C::FLATROWS) {
    int zr = C::PAD;
    if ((g.Ho + zr) & 1) ++zr;  // even rows per frame: 2x2 pool windows never straddle a tile edge
    if (zr == 0) zr = (g.Ho & 1) ? 1 : 0;
    g.rows_per_img = g.Ho + zr;
    g.rows_total = d.batch * g.rows_per_img;
    g.rows_magic = (unsigned)((1ULL << 32) / (unsigned)g.rows_per_img) + 1u;
    SFH_REQUIRE((unsigned long long)(g.rows_total + 64) * g.rows_per_img < (1ULL << 32),
                "conv_s3: flattened row space too large");
    g.tiles_y = sfh_cdiv(g.rows_total, C::TH);
    g.ntiles = g.tiles_x * g.tiles_y;
  } else {
    g.rows_total = 0;
    g.rows_per_img = g.Ho;
    g.rows_magic = 0;
    g.tiles_y = sfh_cdiv(g.Ho, C::TH);
    g.ntiles = g.tiles_x * g.tiles_y * d.batch;
    SFH_REQUIRE(g.Ho < 65536 && d.batch < 32768, "conv_s3: geometry too large");
  }
  const unsigned long long b0 = 2ULL * C::NP * d.batch * d.h0 * d.w0 * d.cs0;
  const unsigned long long b1 = d.src1 ? 2ULL * C::NP * d.batch * d.h1 * d.w1 * d.cs1 : 0ULL;
  SFH_REQUIRE(b0 < kOOB && b1 < kOOB, "conv_s3: a source tensor of %llu bytes exceeds the 4 GiB descriptor range; split the batch",
              b0 > b1 ? b0 : b1);
  g.bytes0 = (unsigned)b0;
  g.bytes1 = (unsigned)b1;
  g.nblk_n = d.cout / C::NCO;
  {
    // weights of one cout block: (cin/32) stages x taps x 12 KB.  Measured on the UNet (B=16, 640x360,
    // DoubleConv ms per batch vs budget for the group's weights): 0.5 MB 22.41, 2 MB 22.10, 16 MB 21.92,
    // 64 MB 21.61, unbounded 21.74 - re-reading the input tile once per cout block costs more than
    // streaming weights that no longer fit the 4 MB L2 (they hit the 256 MB Infinity Cache), so the budget
    // is 64 MB: every layer of this model is fully grouped.
    const long wblock = (long)((d.c0 + (d.src1 ? d.c1 : 0)) / 32) * C::NTAP * (C::NP * 4096);
    const long budget = 64L << 20;
    int G = d.out_mode == SFH_OUT_UPSCATTER2 ? g.nblk_n : (int)(budget / (wblock > 0 ? wblock : 1));
    if (G < 1) G = 1;
    if (G > g.nblk_n) G = g.nblk_n;
    while (g.nblk_n % G) --G;  // groups must tile the cout blocks
    g.nb_group = G;
  }
  const int ksn = d.ksplit > 1 ? d.ksplit : 1;
  const long nblocks = (long)sfh_cdiv(g.ntiles, 8) * 8 * g.nblk_n * ksn;
  SFH_REQUIRE(nblocks < (1L << 31), "conv_s3: grid too large");
  // small grids (at most ~one workgroup per CU anyway, e.g. ResNet layer3/4): the double-buffered
  // variant overlaps each stage's DMA latency with the previous stage's MFMAs
  if constexpr (!DB && C::LDS_BYTES <= 160 * 1024) {
    if (nblocks <= 320) return launch_s3<C, true>(d, stream);
#ifdef SFH_KS2_DB   // experiment (profiles/build_variant.py): the 64-cout 2x2 up-scatter launches double-buffered, two per CU
    if (C::KS == 2 && C::NWN == 2 && C::NP == 2) return launch_s3<C, true>(d, stream);
#endif
  }
  // (an LDS-free variant for 1x1 / transposed convs that streams both operands straight into
  // registers was measured slower: 3.69 ms vs 2.96 ms per step for the four ConvTranspose launches)
  sfh_allow_big_lds(reinterpret_cast<const void*>(&conv_s3_kernel<C, DB>));
  hipLaunchKernelGGL((conv_s3_kernel<C, DB>), dim3((unsigned)nblocks), dim3(C::NT),
                     DB ? C::LDS_BYTES : C::LDS_BYTES / 2, stream, d, g);
  return sfh_check_launch("conv_s3_kernel");
}

}  // namespace

extern "C" int64_t sfh_packed_s3_weight_bytes(int ksize, int c0, int c1, int cout_virtual) {
  if ((ksize != 1 && ksize != 2 && ksize != 3 && ksize != 4) || c0 <= 0 || c0 % 32 || c1 < 0 || c1 % 32 || cout_virtual <= 0 ||
      cout_virtual % 64)
    return -1;
  return (int64_t)(cout_virtual / 64) * ((c0 + c1) / 32) * (ksize * ksize) * 3 * 4096;
}

extern "C" int sfh_pack_s3_weights(const float* w, void* packed, int ksize, int c0, int c1, int cout_virtual,
                                   int mode, int aux, void* stream) {
  const int64_t n = sfh_packed_s3_weight_bytes(ksize, c0, c1, cout_virtual);
  SFH_REQUIRE(n > 0, "pack_s3_weights: bad geometry ks=%d c0=%d c1=%d cout=%d", ksize, c0, c1, cout_virtual);
  SFH_REQUIRE(w && packed, "pack_s3_weights: null pointer");
  SFH_REQUIRE((mode == 0 && ksize != 4) || (mode == 1 && ksize == 1 && c1 == 0 && cout_virtual % 256 == 0) ||
                  (mode == 2 && ksize == 4 && c1 == 0 && aux > 0 && aux <= c0 / 4) ||
                  (mode == 3 && c1 == 0 && (ksize == 1 || ksize == 3) && aux > 0 && aux <= cout_virtual) ||
                  (mode == 4 && ksize == 1 && c1 == 0 && aux > 0 && c0 == 4 * aux),
              "pack_s3_weights: bad mode/geometry (mode %d, ksize %d)", mode, ksize);
  if ((mode == 0 || mode == 3) && (ksize == 3 || ksize == 1)) {
    const long tot = n / 48 / (ksize * ksize);   // one thread per (cout, 8 cins), all taps
    const dim3 grid((unsigned)((tot + 255) / 256));
#define SFH_PACK_AT(KS_, M_)                                                                                     \
  hipLaunchKernelGGL((pack_s3_weights_alltaps_kernel<KS_, M_, 3>), grid, dim3(256), 0, (hipStream_t)stream, w,   \
                     (unsigned short*)packed, c0, c1, cout_virtual, aux, tot, 1.f)
    if (ksize == 3 && mode == 0) SFH_PACK_AT(3, 0);
    else if (ksize == 3) SFH_PACK_AT(3, 3);
    else if (mode == 0) SFH_PACK_AT(1, 0);
    else SFH_PACK_AT(1, 3);
#undef SFH_PACK_AT
    return sfh_check_launch("pack_s3_weights_alltaps_kernel");
  }
  const long total = n / 48;  // one thread per (lane, 8 channels) of all three planes
  hipLaunchKernelGGL(pack_s3_weights_kernel<3>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, (unsigned short*)packed, ksize, c0, c1, cout_virtual, mode, aux, total, 1.f);
  return sfh_check_launch("pack_s3_weights_kernel");
}

extern "C" int64_t sfh_packed_h2_weight_bytes(int ksize, int c0, int c1, int cout_virtual) {
  const int64_t n = sfh_packed_s3_weight_bytes(ksize, c0, c1, cout_virtual);
  return n < 0 ? n : n / 3 * 2;
}

extern "C" int sfh_pack_h2_weights(const float* w, void* packed, int ksize, int c0, int c1, int cout_virtual,
                                   int mode, int aux, int wexp, void* stream) {
  const int64_t n = sfh_packed_h2_weight_bytes(ksize, c0, c1, cout_virtual);
  SFH_REQUIRE(n > 0, "pack_h2_weights: bad geometry ks=%d c0=%d c1=%d cout=%d", ksize, c0, c1, cout_virtual);
  SFH_REQUIRE(w && packed, "pack_h2_weights: null pointer");
  SFH_REQUIRE(wexp >= -100 && wexp <= 100, "pack_h2_weights: wexp=%d out of range", wexp);
  SFH_REQUIRE((mode == 0 && ksize != 4) || (mode == 1 && ksize == 1 && c1 == 0 && cout_virtual % 256 == 0) ||
                  (mode == 2 && ksize == 4 && c1 == 0 && aux > 0 && aux <= c0 / 4) ||
                  (mode == 3 && c1 == 0 && (ksize == 1 || ksize == 3) && aux > 0 && aux <= cout_virtual) ||
                  (mode == 4 && ksize == 1 && c1 == 0 && aux > 0 && c0 == 4 * aux),
              "pack_h2_weights: bad mode/geometry (mode %d, ksize %d)", mode, ksize);
  if ((mode == 0 || mode == 3) && (ksize == 3 || ksize == 1)) {   // whole runs of the checkpoint tensor per thread
    const long tot = n / 32 / (ksize * ksize);
    const dim3 grid((unsigned)((tot + 255) / 256));
    const float wscale = ldexpf(1.f, wexp);
#define SFH_PACK_AT(KS_, M_)                                                                                     \
  hipLaunchKernelGGL((pack_s3_weights_alltaps_kernel<KS_, M_, 2>), grid, dim3(256), 0, (hipStream_t)stream, w,   \
                     (unsigned short*)packed, c0, c1, cout_virtual, aux, tot, wscale)
    if (ksize == 3 && mode == 0) SFH_PACK_AT(3, 0);
    else if (ksize == 3) SFH_PACK_AT(3, 3);
    else if (mode == 0) SFH_PACK_AT(1, 0);
    else SFH_PACK_AT(1, 3);
#undef SFH_PACK_AT
    return sfh_check_launch("pack_h2_weights_alltaps_kernel");
  }
  const long total = n / 32;  // one thread per (lane, 8 channels) of both planes
  hipLaunchKernelGGL(pack_s3_weights_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, (unsigned short*)packed, ksize, c0, c1, cout_virtual, mode, aux, total,
                     ldexpf(1.f, wexp));
  return sfh_check_launch("pack_h2_weights_kernel");
}

extern "C" int sfh_f32_to_s3(const float* src, void* dst, int64_t rows, int W, int cs, void* stream) {
  SFH_REQUIRE(src && dst && rows > 0 && W > 0 && cs > 0 && cs % 32 == 0, "f32_to_s3: cs must be a multiple of 32");
  const int xchunks = (W + 15) / 16;
  const long total = rows * (cs / 32) * xchunks * 64;
  hipLaunchKernelGGL(f32_to_s3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     src, (unsigned short*)dst, W, cs, xchunks, total);
  return sfh_check_launch("f32_to_s3_kernel");
}

extern "C" int sfh_s3_to_f32(const void* src, float* dst, int64_t rows, int W, int cs, void* stream) {
  SFH_REQUIRE(src && dst && rows > 0 && W > 0 && cs > 0 && cs % 32 == 0, "s3_to_f32: cs must be a multiple of 32");
  const long total = rows * W * cs;
  hipLaunchKernelGGL(s3_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)src, dst, W, cs, total);
  return sfh_check_launch("s3_to_f32_kernel");
}

extern "C" int sfh_f32_to_h2(const float* src, void* dst, int64_t rows, int W, int cs, int act_exp, uint32_t* overflow,
                             uint32_t* range, void* stream) {
  SFH_REQUIRE(src && dst && rows > 0 && W > 0 && cs > 0 && cs % 32 == 0, "f32_to_h2: cs must be a multiple of 32");
  SFH_REQUIRE(act_exp >= -64 && act_exp <= 64, "f32_to_h2: act_exp=%d out of range", act_exp);
  const int xchunks = (W + 15) / 16;
  const long total = rows * (cs / 32) * xchunks * 64;
  hipLaunchKernelGGL(f32_to_h2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     src, (unsigned short*)dst, W, cs, xchunks, total, ldexpf(1.f, act_exp), overflow, range);
  return sfh_check_launch("f32_to_h2_kernel");
}

extern "C" int sfh_maxpool3x3s2_split_fwd(const float* x, void* y, int batch, int H, int W, int C, int dst_fmt, int act_exp,
                                          uint32_t* overflow, uint32_t* range, void* stream) {
  SFH_REQUIRE(x && y && batch > 0 && H > 0 && W > 0 && C > 0 && C % 32 == 0, "maxpool3x3s2_split: C=%d must be a multiple of 32", C);
  SFH_REQUIRE(dst_fmt == SFH_FMT_H2 || dst_fmt == SFH_FMT_S3, "maxpool3x3s2_split: dst_fmt=%d (S3 or H2)", dst_fmt);
  SFH_REQUIRE(act_exp >= -64 && act_exp <= 64, "maxpool3x3s2_split: act_exp=%d out of range", act_exp);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int xchunks = (Wo + 15) / 16;
  const long total = (long)batch * Ho * (C / 32) * xchunks * 64;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dst_fmt == SFH_FMT_H2)
    hipLaunchKernelGGL(maxpool3x3s2_split_kernel<SFH_FMT_H2>, grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, H,
                       W, Ho, Wo, C, xchunks, total, ldexpf(1.f, act_exp), overflow, range);
  else
    hipLaunchKernelGGL(maxpool3x3s2_split_kernel<SFH_FMT_S3>, grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)y, H,
                       W, Ho, Wo, C, xchunks, total, 1.f, nullptr, nullptr);
  return sfh_check_launch("maxpool3x3s2_split_kernel");
}

extern "C" int sfh_h2_to_f32(const void* src, float* dst, int64_t rows, int W, int cs, int act_exp, void* stream) {
  SFH_REQUIRE(src && dst && rows > 0 && W > 0 && cs > 0 && cs % 32 == 0, "h2_to_f32: cs must be a multiple of 32");
  SFH_REQUIRE(act_exp >= -64 && act_exp <= 64, "h2_to_f32: act_exp=%d out of range", act_exp);
  const long total = rows * W * cs;
  hipLaunchKernelGGL(h2_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)src, dst, W, cs, total, ldexpf(1.f, -act_exp));
  return sfh_check_launch("h2_to_f32_kernel");
}

extern "C" int sfh_splitk_finish(const float* slabs, int nslabs, int64_t slab_stride, const float* shift,
                                 const void* residual, int res_fmt, int exp_res, int relu, int64_t rows, int W, int C,
                                 void* dst, int dst_fmt, int exp_dst, uint32_t* overflow, uint32_t* range, void* stream) {
  SFH_REQUIRE(slabs && shift && dst && nslabs >= 1 && nslabs <= 64 && rows > 0 && W > 0 && C > 0 && C % 32 == 0,
              "splitk_finish: bad arguments (nslabs=%d, C=%d must be a multiple of 32)", nslabs, C);
  SFH_REQUIRE(slab_stride % 16 == 0 && slab_stride >= rows * W * C * 4, "splitk_finish: slab_stride=%lld is smaller than a slab",
              (long long)slab_stride);
  SFH_REQUIRE(dst_fmt == SFH_FMT_F32 || dst_fmt == SFH_FMT_S3 || dst_fmt == SFH_FMT_H2, "splitk_finish: dst_fmt=%d", dst_fmt);
  SFH_REQUIRE(!residual || res_fmt == SFH_FMT_F32 || res_fmt == SFH_FMT_S3 || res_fmt == SFH_FMT_H2, "splitk_finish: res_fmt=%d", res_fmt);
  SFH_REQUIRE(exp_dst >= -64 && exp_dst <= 64 && exp_res >= -64 && exp_res <= 64, "splitk_finish: exponent out of range");
  const int xchunks = (W + 15) / 16;
  const long total = rows * (C / 32) * xchunks * 64;
  const dim3 grid((unsigned)((total + 255) / 256));
#define SFH_FINISH(F_)                                                                                               \
  hipLaunchKernelGGL(splitk_finish_kernel<F_>, grid, dim3(256), 0, (hipStream_t)stream, slabs, nslabs,               \
                     (long)(slab_stride / 4), shift, residual, res_fmt, ldexpf(1.f, -exp_res), relu, W, C, xchunks, \
                     total, dst, ldexpf(1.f, exp_dst), overflow, range)
  if (dst_fmt == SFH_FMT_H2) SFH_FINISH(SFH_FMT_H2);
  else if (dst_fmt == SFH_FMT_S3) SFH_FINISH(SFH_FMT_S3);
  else SFH_FINISH(SFH_FMT_F32);
#undef SFH_FINISH
  return sfh_check_launch("splitk_finish_kernel");
}

extern "C" int sfh_conv_s3_fwd(const sfh_conv_desc* dp, void* stream_) {
  SFH_REQUIRE(dp, "conv_s3_fwd: null descriptor");
  const sfh_conv_desc& d = *dp;
  hipStream_t stream = (hipStream_t)stream_;
  SFH_REQUIRE(d.src0 && d.wpacked && d.scale && d.shift && d.dst, "conv_s3_fwd: null pointer");
  SFH_REQUIRE(d.src_fmt == SFH_FMT_S3 || d.src_fmt == SFH_FMT_H2, "conv_s3_fwd: sources must be S3 or H2 tensors (src_fmt=%d)", d.src_fmt);
  SFH_REQUIRE(d.dst_fmt == SFH_FMT_F32 || d.dst_fmt == d.src_fmt,
              "conv_s3_fwd: the destination is fp32 or the sources' own split format (src_fmt=%d, dst_fmt=%d)", d.src_fmt, d.dst_fmt);
  SFH_REQUIRE(d.batch > 0 && d.H > 0 && d.W > 0, "conv_s3_fwd: empty geometry");
  SFH_REQUIRE(d.h2_exp_dst >= -64 && d.h2_exp_dst <= 64 && d.h2_exp_res >= -64 && d.h2_exp_res <= 64,
              "conv_s3_fwd: h2_exp_dst=%d / h2_exp_res=%d out of range (-64 .. 64)", d.h2_exp_dst, d.h2_exp_res);
  SFH_REQUIRE(d.cout > 0 && d.cout % 64 == 0, "conv_s3_fwd: cout=%d must be a multiple of 64", d.cout);
  SFH_REQUIRE(d.c0 > 0 && d.c0 % 32 == 0 && d.cs0 >= d.c0, "conv_s3_fwd: c0=%d must be a multiple of 32 (cs0=%d)", d.c0, d.cs0);
  SFH_REQUIRE(!d.pool0, "conv_s3_fwd: pool-on-load is not available for S3 sources (use the producer's dst_pool)");
  SFH_REQUIRE((d.h0 == d.H && d.w0 == d.W) ||
                  (d.ksize == 2 && (d.H == d.h0 || d.H == d.h0 + 1) && (d.W == d.w0 || d.W == d.w0 + 1)),
              "conv_s3_fwd: source 0 is %dx%d, frame is %dx%d", d.h0, d.w0, d.H, d.W);
  SFH_REQUIRE((!d.up_dst_h && !d.up_dst_w) || d.ksize == 2, "conv_s3_fwd: up_dst_h/w exist only for the 2x2 up-scatter conv");
  if (d.ksize == 2)
    SFH_REQUIRE((d.up_dst_h == 0 || (d.up_dst_h >= 2 * d.h0 && d.up_dst_h <= 2 * d.H)) &&
                    (d.up_dst_w == 0 || (d.up_dst_w >= 2 * d.w0 && d.up_dst_w <= 2 * d.W)),
                "conv_s3_fwd: up_dst %dx%d does not match the source %dx%d", d.up_dst_h, d.up_dst_w, d.h0, d.w0);
  SFH_REQUIRE(d.stride == 1 || (d.stride == 2 && !d.src1 && !d.dst_pool && d.out_mode == SFH_OUT_NHWC),
              "conv_s3_fwd: stride 2 supports a single source, plain output");
  if (d.src1) {
    SFH_REQUIRE(d.c1 > 0 && d.c1 % 32 == 0 && d.cs1 >= d.c1, "conv_s3_fwd: c1=%d must be a multiple of 32", d.c1);
    SFH_REQUIRE(d.h1 > 0 && d.w1 > 0 && d.pad_top1 >= 0 && d.pad_left1 >= 0 && d.pad_top1 + d.h1 <= d.H &&
                    d.pad_left1 + d.w1 <= d.W, "conv_s3_fwd: source 1 does not fit the frame");
  }
  if (d.out_mode == SFH_OUT_UPSCATTER2)
    SFH_REQUIRE((d.ksize == 1 || d.ksize == 2) && d.stride == 1 && (d.cout / 4) % 64 == 0 && !d.dst_pool &&
                    (d.ksize == 2 || !d.residual) && !d.src1,
                "conv_s3_fwd: up-scatter needs ksize 1 or 2, stride=1, cout/4 multiple of 64, one source");
  SFH_REQUIRE(d.ksize != 2 || d.out_mode == SFH_OUT_UPSCATTER2, "conv_s3_fwd: ksize 2 exists only as the up-scatter conv");
  SFH_REQUIRE(!d.residual_f32 || (d.residual && d.dst_fmt != SFH_FMT_F32), "conv_s3_fwd: residual_f32 needs a residual and a split-format dst");
  SFH_REQUIRE(!d.residual_f32 || d.ksize == 2 || d.out_mode == SFH_OUT_NHWC, "conv_s3_fwd: residual_f32 with a plain output only");
  SFH_REQUIRE(!d.shift_border || d.ksize == 2, "conv_s3_fwd: shift_border exists only for the 2x2 up-scatter conv");
  if (d.head_w)
    SFH_REQUIRE(d.ksize == 3 && d.stride == 1 && d.cout == 64 && d.out_mode == SFH_OUT_NHWC && !d.dst_pool && d.head_b &&
                    d.head_logits && d.head_nc >= 1 && d.head_nc <= 8 && (!d.head_stn || (d.head_frame && d.head_nc <= 5)),
                "conv_s3_fwd: the fused OutConv head needs a 3x3 stride-1 conv with 64 output channels and a plain output");
  SFH_REQUIRE(!d.head_skip_dst || d.head_w, "conv_s3_fwd: head_skip_dst without a head");
  SFH_REQUIRE(!d.acc_init || (d.ksize == 3 && d.stride == 1 && d.out_mode == SFH_OUT_NHWC && !(d.ksplit > 1) &&
                              (unsigned long long)d.batch * d.H * d.W * d.cout * 4ULL < 0xFFFFFFF0ULL),
              "conv_s3_fwd: acc_init needs a 3x3 stride-1 conv with a plain output and an initial tensor below 4 GiB");
  if (d.ksplit > 1)
    SFH_REQUIRE(d.ksplit <= d.c0 / 32 && !d.src1 && d.dst_fmt == SFH_FMT_F32 && !d.relu && !d.residual && !d.dst_pool &&
                    !d.head_w && d.out_mode == SFH_OUT_NHWC && d.ksplit_stride % 16 == 0 &&
                    d.ksplit_stride >= (int64_t)d.batch * ((d.H + d.ksize / 2 + (d.ksize - 1) / 2 - d.ksize) / d.stride + 1) *
                                           ((d.W + d.ksize / 2 + (d.ksize - 1) / 2 - d.ksize) / d.stride + 1) * d.dst_cs * 4,
                "conv_s3_fwd: split-K (ksplit=%d) needs one source with at least ksplit 32-channel stages, a plain fp32 "
                "destination of ksplit slabs ksplit_stride bytes apart, no ReLU / residual / pooled output / head", d.ksplit);
  // buffering policy (launch_s3): two single-buffered workgroups per CU, except grids of at most 320
  // workgroups, which take the double-buffered variant
  // 8-wave workgroups (256 pixels x 128 couts, one per CU, two LDS buffers): long K, at least 128 couts (a 2x2
  // up-scatter workgroup must stay inside one quadrant), grids of at least SFH_W8_MIN_BLOCKS such workgroups
  const int nst_all = (d.c0 + (d.src1 ? d.c1 : 0)) / 32;
  const bool w8_ok = d.stride == 1 && (d.ksize == 3 || d.ksize == 2) && d.cout % 128 == 0 && !d.head_w && nst_all >= 4 &&
                     (d.ksize != 2 || (d.cout / 4) % 128 == 0) && d.src_fmt == SFH_FMT_H2;
  // the 128-pixel x 128-cout double-buffered shape (below) also takes short K (two stages)
  const bool w8h_ok = d.stride == 1 && d.ksize == 3 && d.cout % 128 == 0 && !d.head_w && nst_all >= 2 && d.src_fmt == SFH_FMT_H2 &&
                      (d.tile == SFH_TILE_8x16 || d.tile == SFH_TILE_16x8);
  SFH_REQUIRE(d.wg_couts == 0 || d.wg_couts == 64 || (d.wg_couts == 128 && (w8_ok || w8h_ok)),
              "conv_s3_fwd: wg_couts=%d is not available for this launch (see sfh_conv_desc.wg_couts)", d.wg_couts);
  if (d.stats_partial) {
    SFH_REQUIRE(d.src_fmt == SFH_FMT_H2 && d.dst_fmt == SFH_FMT_F32 && (d.ksize == 3 || d.ksize == 1) && d.stride == 1 && !d.relu && !d.residual &&
                    !d.dst_pool && !d.head_w && !(d.ksplit > 1) && d.out_mode == SFH_OUT_NHWC && d.stats_rows >= 64 &&
                    d.stats_rows <= 65536 && (d.stats_rows & (d.stats_rows - 1)) == 0,
                "conv_s3_fwd: stats_partial needs H2 sources, a 3x3 or 1x1 stride-1 conv with a plain fp32 destination (no ReLU / "
                "residual / pooled output / head / split-K) and stats_rows a power of two in 64 .. 65536");
    SFH_REQUIRE(!d.bwd_z || (d.bwd_mi && d.dst_cs == d.cout && (!d.bwd_gamma == !d.bwd_beta) &&
                             (unsigned long long)d.batch * d.H * d.W * d.cout * 4ULL < 0xFFFFFFF0ULL),
                "conv_s3_fwd: bwd_z needs bwd_mi, bwd_gamma and bwd_beta together, dst_cs == cout and tensors below 4 GiB");
  } else {
    SFH_REQUIRE(!d.bwd_z, "conv_s3_fwd: bwd_z without stats_partial");
  }
#define SFH_S3CASE_W8(KS, ST, TILE, SH, SW, TH, TW)                                                          \
  if (w8_ok && d.ksize == KS && d.stride == ST && d.tile == TILE) {                                           \
    using CFG = S3Cfg<KS, ST, SH, SW, TH, TW, 2, 4, SFH_W128_NWM>;                                            \
    using CFGS = S3Cfg<KS, ST, SH, SW, TH, TW, 2, 4, SFH_W128_NWM, true>;                                     \
    const int Ho_ = d.H, Wo_ = d.W;                                                                           \
    int zr_ = CFG::PAD;                                                                                       \
    if ((Ho_ + zr_) & 1) ++zr_;                                                                               \
    if (zr_ == 0) zr_ = (Ho_ & 1) ? 1 : 0;                                                                    \
    const long tiles_ = (long)sfh_cdiv(Wo_, TW) * sfh_cdiv(d.batch * (Ho_ + zr_), TH);                        \
    if (d.wg_couts == 128 || (d.wg_couts == 0 && tiles_ * (d.cout / 128) >= SFH_W8_MIN_BLOCKS)) {            \
      if constexpr (KS == 3) {                                                                                \
        if (d.stats_partial) return launch_s3<CFGS, (SFH_W128_NWM == 2)>(d, stream);                          \
      }                                                                                                       \
      return launch_s3<CFG, (SFH_W128_NWM == 2)>(d, stream);                                                  \
    }                                                                                                         \
  }
  SFH_S3CASE_W8(3, 1, SFH_TILE_8x32, 1, 16, 8, 32)
  SFH_S3CASE_W8(3, 1, SFH_TILE_16x16, 1, 16, 16, 16)
  SFH_S3CASE_W8(3, 1, SFH_TILE_32x8, 2, 8, 32, 8)
  SFH_S3CASE_W8(2, 1, SFH_TILE_8x32, 1, 16, 8, 32)
  SFH_S3CASE_W8(2, 1, SFH_TILE_16x16, 1, 16, 16, 16)
  if (SFH_W128_NWM == 2) {  // (the 4-wave 128-cout instance of this shape spills registers)
    SFH_S3CASE_W8(2, 1, SFH_TILE_32x8, 2, 8, 32, 8)
  }
#undef SFH_S3CASE_W8
  // 128-pixel x 128-cout workgroups (1 x 4 waves, half-size tile) with TWO LDS buffers of 24.5 KB: 151 VGPRs, i.e. three
  // workgroups per CU, each with the next stage's DMA under its own MFMAs and one barrier per stage.  Explicit request only
  // (wg_couts = 128 with a half-size tile): +3 % on 128- and 256-channel layers at 180x320 / 90x160, -2 .. -5 % on
  // 512-channel ones (profiles/r03_conv_rate_probe_w8half.txt) - the engine asks for it where it wins.
  if (w8h_ok && d.wg_couts == 128 && !d.stats_partial) {
    if (d.tile == SFH_TILE_8x16) return launch_s3<S3Cfg<3, 1, 1, 16, 8, 16, 2, 4, 1>, true>(d, stream);
    if (d.tile == SFH_TILE_16x8) return launch_s3<S3Cfg<3, 1, 2, 8, 16, 8, 2, 4, 1>, true>(d, stream);
  }
#define SFH_S3CASE(KS, ST, TILE, SH, SW, TH, TW)                      \
  if (d.ksize == KS && d.stride == ST && d.tile == TILE) {             \
    if (d.src_fmt == SFH_FMT_H2) {                                     \
      using CFG = S3Cfg<KS, ST, SH, SW, TH, TW, 2>;                    \
      if constexpr ((KS == 3 || KS == 1) && ST == 1) {                 \
        using CFGS = S3Cfg<KS, ST, SH, SW, TH, TW, 2, 2, 2, true>;     \
        if (d.stats_partial) return launch_s3<CFGS, false>(d, stream); \
      }                                                                \
      return launch_s3<CFG, false>(d, stream);                         \
    }                                                                  \
    using CFG = S3Cfg<KS, ST, SH, SW, TH, TW, 3>;                      \
    return launch_s3<CFG, false>(d, stream);                           \
  }
  SFH_S3CASE(3, 1, SFH_TILE_8x32, 1, 16, 8, 32)
  SFH_S3CASE(3, 1, SFH_TILE_16x16, 1, 16, 16, 16)
  SFH_S3CASE(3, 1, SFH_TILE_32x8, 2, 8, 32, 8)
  SFH_S3CASE(1, 1, SFH_TILE_8x32, 1, 16, 8, 32)
  SFH_S3CASE(1, 1, SFH_TILE_16x16, 1, 16, 16, 16)
  SFH_S3CASE(1, 1, SFH_TILE_32x8, 2, 8, 32, 8)
  SFH_S3CASE(2, 1, SFH_TILE_8x32, 1, 16, 8, 32)
  SFH_S3CASE(2, 1, SFH_TILE_16x16, 1, 16, 16, 16)
  SFH_S3CASE(2, 1, SFH_TILE_32x8, 2, 8, 32, 8)
  // half-size tiles (8 pixel groups) for small feature maps: twice the workgroups
  SFH_S3CASE(3, 1, SFH_TILE_8x16, 1, 16, 8, 16)
  SFH_S3CASE(3, 1, SFH_TILE_16x8, 2, 8, 16, 8)
  // 4x4 stem over the space-to-depth input
  SFH_S3CASE(4, 1, SFH_TILE_8x32, 1, 16, 8, 32)
  SFH_S3CASE(4, 1, SFH_TILE_16x16, 1, 16, 16, 16)
  SFH_S3CASE(4, 1, SFH_TILE_32x8, 2, 8, 32, 8)
  // stride 2 (ResNet stage transitions): 8-group tiles keep the halo within LDS
  SFH_S3CASE(3, 2, SFH_TILE_8x16, 1, 16, 8, 16)
  SFH_S3CASE(3, 2, SFH_TILE_16x8, 2, 8, 16, 8)
  SFH_S3CASE(1, 2, SFH_TILE_8x16, 1, 16, 8, 16)
  SFH_S3CASE(1, 2, SFH_TILE_16x8, 2, 8, 16, 8)
#undef SFH_S3CASE
  sfh_set_error("conv_s3_fwd: unsupported ksize=%d stride=%d tile=%d", d.ksize, d.stride, d.tile);
  return SFH_E_ARG;
}
