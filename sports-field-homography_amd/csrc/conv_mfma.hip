// conv_mfma.hip - fp32 implicit-GEMM convolution on the gfx950 matrix cores.
//
// One kernel family covers the conv-shaped work of the hot path:
//   * conv3x3 pad1 + bias + BatchNorm(eval) + ReLU   (DoubleConv, unet/unet_parts.py:14-21)
//   * MaxPool2d(2) fused into the consumer's loads    (Down, unet/unet_parts.py:33)
//   * F.pad + torch.cat([skip, up]) as a 2-source load (Up, unet/unet_parts.py:59-67)
//   * ConvTranspose2d k2 s2 as a 1x1 GEMM with a scatter epilogue (Up, unet/unet_parts.py:52)
//   * stride-2 / residual variants of BasicBlock      (models/resnet.py:64-82)
//
// Arithmetic: v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate) - bit-for-bit an fp32
// fmaf chain, which is what the 1e-4 / exact-argmax parity target of the path needs
// (bf16 inputs would not hold it through ~45 layers).  Roofline: fp32 matrix peak
// 157.3 TFLOP/s (MI355X_MICROARCH.md).
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = (tap, cin).
//   A operand = weights  (row = cout within a 16-group, k = lane>>4)
//   B operand = pixels   (col = pixel within a 16-group, k = lane>>4)
//   D: lane holds pixel (lane&15) and couts 4*(lane>>4)..+3  -> one 16-byte NHWC store.
// One ds_read_b128 of a pixel's 4 consecutive channels (or of 4 consecutive cin of one
// cout) feeds 4 MFMAs: register j of lane-group g is channel 4g+j of the 16-channel stage.
//
// Workgroup = 256 threads = 4 waves; block tile = NSUBT pixel-groups x 64 couts; wave w
// owns pixel-groups [w*MT_M, (w+1)*MT_M) x all 4 cout-groups (MT_M*4 accumulators).
// LDS per stage (16*NSUB input channels):
//   halo  [4*NSUB planes][HPIXP pixels][4 ch]   (plane stride multiple of 256 B ->
//                                                conflict-free ds_read_b128 for 1x16 groups)
//   wts   [taps][4 cout-groups][64 lanes][4]    (fragment order, linear copy of the
//                                                pre-packed global image)
// Stages are register-prefetched: the global loads of stage s+1 are issued before the
// MFMA block of stage s and written to LDS after it.  Two workgroups per CU (<= 60 KB
// LDS, <= 256 VGPRs) overlap each other's staging with MFMA issue.
//
// Row space: for stride 1 the output rows of all frames are flattened with ONE shared
// zero row between consecutive frames (row index r = b*(H+1) + y, y == H is the zero
// row), so row tiles may straddle frames and only the last tile is partial.
#include <stdlib.h>

#include "common.h"
#include "conv_epilogue.h"

// Diagnostic build only (-DSFH_DIAG_STAMPS, libsfh_amd_diag.so): the stamp macros live in common.h; this
// translation unit owns the accumulator and its reader.  Never compiled into the shipped library.
#ifdef SFH_DIAG_STAMPS
__device__ unsigned long long g_stamps[16];
extern "C" int sfh_debug_read_stamps(unsigned long long* host_out, int reset) {
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -2;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -2;
  }
  return 0;
}
#endif

namespace {

template <int KS_, int STRIDE_, int NSUB_, int SH_, int SW_, int TH_, int TW_>
struct ConvCfg {
  static constexpr int KS = KS_, STRIDE = STRIDE_, NSUB = NSUB_;
  static constexpr int SH = SH_, SW = SW_, TH = TH_, TW = TW_;
  // padding before / after: 3x3 -> 1/1, 1x1 -> 0/0, 4x4 (space-to-depth stem) -> 2/1
  static constexpr int PAD = KS / 2;
  static constexpr int PADA = (KS - 1) / 2;
  // zero rows shared between consecutive frames in the flattened row space (>= pad)
  static constexpr int ZROWS = PAD;
  static constexpr int NTAP = KS * KS * NSUB;
  // taps per LDS stage: the 16-tap 4x4 stem stages its weights in two halves
  static constexpr int TPS = NTAP > 9 ? NTAP / 2 : NTAP;
  static constexpr int TG = NTAP / TPS;
  static constexpr int PLANES = 4 * NSUB;
  static constexpr int CKS = 16 * NSUB;
  static constexpr int HH = (TH - 1) * STRIDE + KS;
  static constexpr int HW = (TW - 1) * STRIDE + KS;
  static constexpr int HPIX = HH * HW;
  static constexpr int HPIXP = (HPIX + 15) / 16 * 16;
  static constexpr int HSLOTS = PLANES * HPIXP;
  static constexpr int NSL = (HSLOTS + 255) / 256;
  static constexpr int SUBX = TW / SW;
  static constexpr int NSUBT = (TH / SH) * SUBX;
  static constexpr int MT_M = NSUBT / 4;
  static constexpr int LDS_BYTES = (HSLOTS + TPS * 256) * 16;
  static constexpr bool FLATROWS = (STRIDE == 1);
  static_assert(SH * SW == 16, "pixel group must hold 16 pixels");
  static_assert(NSUBT % 4 == 0, "tile must split over 4 waves");
  static_assert(TH % SH == 0 && TW % SW == 0, "tile/group mismatch");
};

struct ConvGeom {
  int tiles_x, tiles_y, ntiles, nblk_n;
  int Ho, Wo, rows_total;
  int rows_per_img;     // flat row space: H + shared zero rows
  unsigned rows_magic;  // floor(2^32 / rows_per_img) + 1: r / rows_per_img == umulhi(r, magic)
  unsigned bytes0, bytes1;  // byte sizes of the two source tensors (buffer range check)
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0xFFFFFFF0u;  // voffset of a halo slot that must read zeros

// 16-byte buffer load: out-of-range offsets (kOOB) return zeros without a branch; the
// per-stage channel advance rides in the scalar offset, so staging needs no vector ALU work
// (fp32 MFMA and VALU instructions contend for the same SIMD issue: every VALU instruction
// in the stage loop is matrix time lost).
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0));
}

template <class C>
__device__ __forceinline__ unsigned halo_voffset(const sfh_conv_desc& d, const ConvGeom& g, int which,
                                                 int slot, int r0, int x0) {
  // slot -> (channel plane, halo pixel); returns the byte offset of the pixel's channel quad
  // of the FIRST stage in the source tensor, or kOOB.  r0: first output row of the tile (flat
  // row space, or img*2^16 + y for the per-image policy); x0: first output column.
  if (slot >= C::HSLOTS) return kOOB;
  const int plane = slot / C::HPIXP, p = slot - plane * C::HPIXP;
  if (p >= C::HPIX) return kOOB;
  const int cs = which == 0 ? d.cs0 : d.cs1;
  if (cs < C::CKS && 4 * plane >= cs) return kOOB;  // narrow source (e.g. RGB stored as 4)
  const int hy = p / C::HW, hx = p - hy * C::HW;
  int b, y;
  if (C::FLATROWS) {
    const int r = r0 - C::PAD + hy;
    if (r < 0) return kOOB;
    b = (int)__umulhi((unsigned)r, g.rows_magic);
    y = r - b * g.rows_per_img;
    if (b >= d.batch || y >= d.H) return kOOB;
  } else {
    b = r0 >> 16;
    y = (r0 & 0xFFFF) * C::STRIDE - C::PAD + hy;
    if (y < 0 || y >= d.H) return kOOB;
  }
  const int x = x0 * C::STRIDE - C::PAD + hx;
  if (x < 0 || x >= d.W) return kOOB;
  unsigned pix;
  if (which == 0) {
    pix = d.pool0 ? (unsigned)((b * d.h0 + 2 * y) * d.w0 + 2 * x) : (unsigned)((b * d.h0 + y) * d.w0 + x);
  } else {
    const int ys = y - d.pad_top1, xs = x - d.pad_left1;
    if (ys < 0 || ys >= d.h1 || xs < 0 || xs >= d.w1) return kOOB;
    pix = (unsigned)((b * d.h1 + ys) * d.w1 + xs);
  }
  return (pix * (unsigned)cs + 4u * plane) * 4u;
}

struct TileCoord {
  int nb, r0, x0;
  bool live;
};

// XCD-aware block -> (tile, cout block): blocks b and b+8 share an XCD (L2); keep the cout
// blocks of one pixel tile on one XCD so the halo is re-read from that L2.
template <class C>
__device__ __forceinline__ TileCoord decode_block(const ConvGeom& g) {
  TileCoord t;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, k = bid >> 3;
  t.nb = k % g.nblk_n;
  const int tile = (k / g.nblk_n) * 8 + xcd;
  t.live = tile < g.ntiles;
  const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
  t.x0 = tx * C::TW;
  if (C::FLATROWS) {
    t.r0 = ty * C::TH;
  } else {
    const int img = ty / g.tiles_y;  // tiles_y = tiles per image
    t.r0 = (img << 16) | ((ty - img * g.tiles_y) * C::TH);
  }
  return t;
}

template <class C>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const sfh_conv_desc d, const ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  f32x4* const halo = reinterpret_cast<f32x4*>(smem_f);
  f32x4* const wlds = halo + C::HSLOTS;

  SFH_STAMP_INIT();
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int lq = lane & 15;  // pixel (B operand col) / cout (A operand row) within a group
  const int lg = lane >> 4;  // k index of the MFMA = channel quad of the stage

  // XCD-aware block -> (tile, cout block): blocks b and b+8 share an XCD (L2); keep the
  // cout blocks of one pixel tile on one XCD so the halo is re-read from that L2.
  const int bid = blockIdx.x;
  const int xcd = bid & 7, k = bid >> 3;
  const int nb = k % g.nblk_n;
  const int tile = (k / g.nblk_n) * 8 + xcd;
  if (tile >= g.ntiles) return;
  const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
  const int x0 = tx * C::TW;
  int r0;
  if (C::FLATROWS) {
    r0 = ty * C::TH;
  } else {
    const int tpi = g.tiles_y;  // tiles per image
    const int img = ty / tpi;
    r0 = (img << 16) | ((ty - img * tpi) * C::TH);
  }
  const int n0 = nb * 64;

  // a stage = (16*NSUB-channel chunk, tap group); chunks of source 0 come first
  const int nst0 = (d.c0 + C::CKS - 1) / C::CKS * C::TG;
  const int nst1 = d.src1 ? (d.c1 + C::CKS - 1) / C::CKS * C::TG : 0;
  const int nst = nst0 + nst1;

  // buffer descriptors (wave-uniform: kernel arguments and blockIdx only)
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.src0), 0, (int)g.bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.src1 ? d.src1 : d.src0), 0, (int)(d.src1 ? g.bytes1 : 0u), 0x00020000);
  const unsigned wstage_bytes = C::TPS * 4096u;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.wpacked) + (size_t)nb * nst * (C::TPS * 1024), 0, (int)(nst * wstage_bytes),
      0x00020000);
  const unsigned wvoff = tid * 16u;
  // pool-on-load: the 2x2 window's other three pixels as scalar byte deltas
  const unsigned pd1 = d.cs0 * 4u, pd2 = (unsigned)d.w0 * d.cs0 * 4u;

  // ---- per-thread halo slot byte offsets (recomputed when the source switches) ----------
  unsigned hoff[C::NSL];
  int which = 0;
#pragma unroll
  for (int i = 0; i < C::NSL; ++i) hoff[i] = halo_voffset<C>(d, g, 0, tid + 256 * i, r0, x0);

  f32x4 hreg[C::NSL];
  f32x4 wreg[C::TPS];

  auto load_stage = [&](int st) {
    const bool first = st < nst0;
    const unsigned cb = (unsigned)((first ? st : st - nst0) / C::TG) * (C::CKS * 4u);  // channel byte offset
    if (C::TG == 1 || st % C::TG == 0) {  // the halo is shared by the tap groups of a chunk
      if (first && d.pool0) {
#pragma unroll
        for (int i = 0; i < C::NSL; ++i) {
          const f32x4 a0 = bload(rs0, hoff[i], cb), a1 = bload(rs0, hoff[i], cb + pd1);
          const f32x4 a2 = bload(rs0, hoff[i], cb + pd2), a3 = bload(rs0, hoff[i], cb + pd2 + pd1);
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = sfh_max_nan(sfh_max_nan(a0[j], a1[j]), sfh_max_nan(a2[j], a3[j]));
          hreg[i] = v;
        }
      } else if (first) {
#pragma unroll
        for (int i = 0; i < C::NSL; ++i) hreg[i] = bload(rs0, hoff[i], cb);
      } else {
#pragma unroll
        for (int i = 0; i < C::NSL; ++i) hreg[i] = bload(rs1, hoff[i], cb);
      }
    }
    const unsigned wb = (unsigned)st * wstage_bytes;
#pragma unroll
    for (int t = 0; t < C::TPS; ++t) wreg[t] = bload(rsw, wvoff, wb + t * 4096u);
  };

  // ---- per-lane LDS read bases (B operand = pixels) ----------------------------------
  int pixbase[C::MT_M];
#pragma unroll
  for (int mi = 0; mi < C::MT_M; ++mi) {
    const int s = wv * C::MT_M + mi;
    const int sy = s / C::SUBX, sx = s - sy * C::SUBX;
    const int oy = sy * C::SH + lq / C::SW, ox = sx * C::SW + lq % C::SW;
    pixbase[mi] = lg * C::HPIXP + oy * C::STRIDE * C::HW + ox * C::STRIDE;
  }

  f32x4 acc[4][C::MT_M];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < C::MT_M; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_stage(0);
  SFH_STAMP(0);  // prologue
  for (int st = 0; st < nst; ++st) {
    // Staging code runs at raised priority: with two waves per SIMD it otherwise loses VALU /
    // SALU issue arbitration to the co-resident wave's MFMA stream (measured: ~45 % of wave
    // time spent issuing ~300 staging instructions).  The MFMA stream has issue slack (8 of
    // every 32 cycles), so it loses nothing.
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();  // all waves finished reading the previous stage from LDS
    SFH_STAMP(1);  // barrier 1 (skew between the waves of the block)
    if (C::TG == 1 || st % C::TG == 0) {
#pragma unroll
      for (int i = 0; i < C::NSL; ++i) {
        const int s = tid + 256 * i;
        if (s < C::HSLOTS) halo[s] = hreg[i];
      }
    }
#pragma unroll
    for (int t = 0; t < C::TPS; ++t) wlds[t * 256 + tid] = wreg[t];
    SFH_STAMP(2);  // wait for prefetched data + LDS writes
    __syncthreads();
    SFH_STAMP(3);  // barrier 2

    if (st + 1 < nst) {
      if (st + 1 == nst0 && which == 0) {  // switch to source 1: recompute slot offsets
        which = 1;
#pragma unroll
        for (int i = 0; i < C::NSL; ++i) hoff[i] = halo_voffset<C>(d, g, 1, tid + 256 * i, r0, x0);
      }
      load_stage(st + 1);
    }
    __builtin_amdgcn_s_setprio(0);
    SFH_STAMP(4);  // address arithmetic + issue of the next stage's global loads

    // MFMA block of this stage.  The LDS operand reads of tap t+1 are issued before the
    // 16*MT_M MFMAs of tap t (two register sets, static indices) so their latency hides
    // under ~2000 cycles of matrix work instead of stalling each tap boundary.
    const int tbase = (C::TG == 1) ? 0 : (st % C::TG) * C::TPS;
    f32x4 xv[2][C::MT_M], wv4[2][4];
    auto ld_tap = [&](int tl, int buf) {
      constexpr int KK = C::KS * C::KS;
      const int tgl = tbase + tl;  // tap index within the chunk
      const int sub = tgl / KK, kk = tgl % KK;
      const int toff = sub * 4 * C::HPIXP + (kk / C::KS) * C::HW + (kk % C::KS);
#pragma unroll
      for (int mi = 0; mi < C::MT_M; ++mi) xv[buf][mi] = halo[pixbase[mi] + toff];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) wv4[buf][ni] = wlds[(tl * 4 + ni) * 64 + lane];
    };
    ld_tap(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, C::MT_M + 4, 0);  // DS_READ: tap 0 operands
#pragma unroll
    for (int tl = 0; tl < C::TPS; ++tl) {
      const int cur = tl & 1;
      if (tl + 1 < C::TPS) ld_tap(tl + 1, cur ^ 1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int mi = 0; mi < C::MT_M; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv4[cur][ni][j], xv[cur][mi][j],
                                                               acc[ni][mi], 0, 0, 0);
      // pin the interleave: one LDS read of the NEXT tap behind each of the first MFMAs of this tap
      constexpr int NMF = 16 * C::MT_M, NRD = C::MT_M + 4;
      if (tl + 1 < C::TPS) {
#pragma unroll
        for (int i = 0; i < NRD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS_READ
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NMF - NRD, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
      }
    }
    SFH_STAMP(5);  // MFMA block (incl. matrix-pipe sharing with the co-resident wave)
  }

  sfh_conv_epilogue<C, 4, C::MT_M>(d, g, acc, n0, wv * C::MT_M, r0, x0, lq, lg);
  SFH_STAMP(6);  // epilogue
  SFH_STAMP_FLUSH();
}

// ------------------------------------------------------------------------------------------
// First UNet layer (3 input channels stored as 4): tap-packed K.  The generic kernel spends a
// 16-channel stage (4 MFMA k-steps) per tap on 3 real channels; here one MFMA k-step (k = 4) IS
// one tap: k index = channel (r, g, b, 0), nine k-steps in total.  All weights of the 64 couts
// (9 taps x 4 cout groups, one float per lane each) stay in registers; the halo is 5.4 KB of LDS.
struct C4Cfg {  // tile geometry seen by the shared epilogue: 8 rows x 32 cols, 1x16 pixel groups
  static constexpr int SUBX = 2, SH = 1, SW = 16, TH = 8, TW = 32, KS = 3;
  static constexpr bool FLATROWS = true;
  static constexpr int HW = 34, HPIX = 10 * 34;
};

__global__ __launch_bounds__(256, 2) void conv3x3_c4_kernel(const sfh_conv_desc d, const ConvGeom g) {
  __shared__ __attribute__((aligned(16))) f32x4 halo[C4Cfg::HPIX + 12];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int lq = lane & 15, lg = lane >> 4;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, k = bid >> 3;
  const int nb = k % g.nblk_n;
  const int tile = (k / g.nblk_n) * 8 + xcd;
  if (tile >= g.ntiles) return;
  const int ty = tile / g.tiles_x, tx = tile - ty * g.tiles_x;
  const int x0 = tx * C4Cfg::TW, r0 = ty * C4Cfg::TH;
  const int n0 = nb * 64;

  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.src0), 0, (int)g.bytes0, 0x00020000);
  // halo: one float4 (the pixel's 4 stored channels) per slot, zeros outside the frame
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = tid + 256 * i;
    if (p < C4Cfg::HPIX) {
      const int hy = p / C4Cfg::HW, hx = p - hy * C4Cfg::HW;
      const int r = r0 - 1 + hy, x = x0 - 1 + hx;
      unsigned off = kOOB;
      if (r >= 0 && x >= 0 && x < d.W) {
        const int b = (int)__umulhi((unsigned)r, g.rows_magic);
        const int y = r - b * g.rows_per_img;
        if (b < d.batch && y < d.H) off = (unsigned)((b * d.H + y) * d.W + x) * 16u;
      }
      halo[p] = bload(rs0, off, 0);
    }
  }
  // weights: packed [nb][tap 9][cout group 4][lane 64] floats (lane = 16*channel + cout)
  const float* wp = d.wpacked + (size_t)nb * (9 * 256) + lane;
  float wr[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) wr[t][ni] = wp[(t * 4 + ni) * 64];
  __syncthreads();

  f32x4 acc[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* hf = reinterpret_cast<const float*>(halo);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float xv[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int s = wv * 4 + mi;
      const int pix = (s / 2 + t / 3) * C4Cfg::HW + (s % 2) * 16 + lq + t % 3;
      xv[mi] = hf[pix * 4 + lg];
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t][ni], xv[mi], acc[ni][mi], 0, 0, 0);
  }
  sfh_conv_epilogue<C4Cfg, 4, 4, 3>(d, g, acc, n0, wv * 4, r0, x0, lq, lg);  // S3 or H2 destination
}

__global__ void pack_c4_weights_kernel(const float* __restrict__ w, float* __restrict__ packed, int cin,
                                       int cout, int total) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, ng = (idx >> 6) & 3, t = (idx >> 8) % 9, nb = idx / (9 * 256);
  const int co = nb * 64 + ng * 16 + (lane & 15), c = lane >> 4;
  packed[idx] = (c < cin && co < cout) ? w[((size_t)(co * cin + c) * 3 + t / 3) * 3 + t % 3] : 0.f;
}

// ------------------------------------------------------------------ weight packing
// packed[nb][stage][tap][ng(4)][lane(64)][j(4)], tap = sub*KS*KS + ky*KS + kx:
//   cout = nb*64 + ng*16 + (lane&15);  channel-in-source = stage_local*CKS + 16*sub + 4*(lane>>4) + j
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed,
                                    int ks, int nsub, int c0, int c1, int coutv, int transposed,
                                    int aux, long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 per thread
  if (idx >= total4) return;
  const int cks = 16 * nsub, kk2 = ks * ks, ntap = kk2 * nsub;
  const int nst0 = (c0 + cks - 1) / cks, nst1 = c1 > 0 ? (c1 + cks - 1) / cks : 0;
  const int nst = nst0 + nst1;
  long r = idx;
  const int lane = r & 63; r >>= 6;
  const int ng = r & 3; r >>= 2;
  const int tap = r % ntap; r /= ntap;
  const int st = r % nst;
  const int nb = r / nst;
  const int sub = tap / kk2, kk = tap % kk2;
  const int ky = kk / ks, kx = kk % ks;
  const int cv = nb * 64 + ng * 16 + (lane & 15);
  const int cin_total = c0 + c1;
  f32x4 out = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int cl = (st < nst0 ? st : st - nst0) * cks + 16 * sub + 4 * (lane >> 4) + j;
    const int lim = st < nst0 ? c0 : c1;
    if (cl >= lim) continue;
    const int cin = st < nst0 ? cl : c0 + cl;
    float v;
    if (transposed == 2) {  // 7x7 stride-2 stem as a 4x4 conv over the 2x2 space-to-depth input
      const int csd = c0 >> 2;                  // padded channels of the un-shuffled source
      const int par = cl / csd, c = cl - par * csd;
      const int ky7 = 2 * ky + (par >> 1) - 1, kx7 = 2 * kx + (par & 1) - 1;
      if (c >= aux || ky7 < 0 || ky7 > 6 || kx7 < 0 || kx7 > 6) continue;
      v = w[(((size_t)cv * aux + c) * 7 + ky7) * 7 + kx7];
    } else if (transposed == 3) {  // backward-data of Conv2d: input = dz (orig cout), output = dx (orig cin = aux)
      if (cv >= aux) continue;
      v = w[(((size_t)cin * aux + cv) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
    } else if (transposed == 4) {  // backward-data of ConvTranspose2d as a 1x1 conv over space-to-depth(dY)
      const int qd = cin / aux, co = cin - qd * aux;  // aux = orig cout; input channel = (py*2+px)*cout + co
      v = w[(((size_t)cv * aux + co) * 2 + (qd >> 1)) * 2 + (qd & 1)];
    } else if (transposed) {  // ConvTranspose2d weight (cin, cout, 2, 2); cv = (dy*2+dx)*cout + co
      const int cout = coutv >> 2;
      const int qd = cv / cout, co = cv - qd * cout;
      v = w[(((size_t)cin * cout + co) * 2 + (qd >> 1)) * 2 + (qd & 1)];
    } else {  // Conv2d weight (cout, cin, ks, ks)
      v = w[(((size_t)cv * cin_total + cin) * ks + ky) * ks + kx];
    }
    out[j] = v;
  }
  reinterpret_cast<f32x4*>(packed)[idx] = out;
}

int nsub_for(int ksize) { return ksize == 1 ? 2 : 1; }
bool ksize_ok(int k) { return k == 1 || k == 3 || k == 4; }

template <class C>
int launch_conv(const sfh_conv_desc& d, hipStream_t stream) {
  ConvGeom g;
  g.Ho = (d.H + C::PAD + C::PADA - C::KS) / C::STRIDE + 1;
  g.Wo = (d.W + C::PAD + C::PADA - C::KS) / C::STRIDE + 1;
  g.tiles_x = sfh_cdiv(g.Wo, C::TW);
  if (C::FLATROWS) {
    g.rows_per_img = g.Ho + C::ZROWS;
    g.rows_total = d.batch * g.rows_per_img;
    g.rows_magic = (unsigned)((1ULL << 32) / (unsigned)g.rows_per_img) + 1u;
    SFH_REQUIRE((unsigned long long)(g.rows_total + 64) * g.rows_per_img < (1ULL << 32),
                "conv_fwd: flattened row space too large for the reciprocal division");
    g.tiles_y = sfh_cdiv(g.rows_total, C::TH);
    g.ntiles = g.tiles_x * g.tiles_y;
  } else {
    g.rows_total = 0;
    g.rows_per_img = g.Ho;
    g.rows_magic = 0;
    g.tiles_y = sfh_cdiv(g.Ho, C::TH);  // per image
    g.ntiles = g.tiles_x * g.tiles_y * d.batch;
    SFH_REQUIRE(g.Ho < 65536 && d.batch < 32768, "stride-2 conv: geometry too large");
  }
  const unsigned long long b0 = 4ULL * d.batch * d.h0 * d.w0 * d.cs0;
  const unsigned long long b1 = d.src1 ? 4ULL * d.batch * d.h1 * d.w1 * d.cs1 : 0ULL;
  SFH_REQUIRE(b0 < kOOB && b1 < kOOB,
              "conv_fwd: a source tensor of %llu bytes exceeds the 4 GiB buffer-descriptor range; split the batch",
              b0 > b1 ? b0 : b1);
  g.bytes0 = (unsigned)b0;
  g.bytes1 = (unsigned)b1;
  g.nblk_n = d.cout / 64;
  const long nblocks = (long)sfh_cdiv(g.ntiles, 8) * 8 * g.nblk_n;
  SFH_REQUIRE(nblocks < (1L << 31), "conv grid too large");
  sfh_allow_big_lds(reinterpret_cast<const void*>(&conv_mfma_kernel<C>));
  hipLaunchKernelGGL(conv_mfma_kernel<C>, dim3((unsigned)nblocks), dim3(256), C::LDS_BYTES, stream, d, g);
  return sfh_check_launch("conv_mfma_kernel");
}

}  // namespace

extern "C" int64_t sfh_packed_weight_floats(int ksize, int c0, int c1, int cout_virtual) {
  if (!ksize_ok(ksize) || c0 <= 0 || c1 < 0 || cout_virtual <= 0 || cout_virtual % 64)
    return -1;
  const int nsub = nsub_for(ksize), cks = 16 * nsub;
  const int64_t nst = (c0 + cks - 1) / cks + (c1 > 0 ? (c1 + cks - 1) / cks : 0);
  return (int64_t)(cout_virtual / 64) * nst * (ksize * ksize * nsub) * 1024;
}

extern "C" int sfh_pack_conv_weights(const float* w, float* packed, int ksize, int c0, int c1,
                                     int cout_virtual, int transposed, int aux, void* stream) {
  const int64_t n = sfh_packed_weight_floats(ksize, c0, c1, cout_virtual);
  SFH_REQUIRE(n > 0, "pack_conv_weights: bad geometry ks=%d c0=%d c1=%d cout=%d", ksize, c0, c1,
              cout_virtual);
  SFH_REQUIRE(w && packed, "pack_conv_weights: null pointer");
  SFH_REQUIRE(transposed >= 0 && transposed <= 4, "pack_conv_weights: bad mode %d", transposed);
  SFH_REQUIRE(transposed != 3 || (c1 == 0 && (ksize == 1 || ksize == 3) && aux > 0 && aux <= cout_virtual),
              "pack_conv_weights: mode 3 (backward-data) needs c0 = orig cout, c1 = 0, aux = orig cin <= cout_virtual");
  SFH_REQUIRE(transposed != 4 || (ksize == 1 && c1 == 0 && aux > 0 && c0 == 4 * aux),
              "pack_conv_weights: mode 4 (ConvTranspose2d backward-data) needs ksize=1, c0 = 4*orig cout, aux = orig cout");
  SFH_REQUIRE(transposed != 1 || (ksize == 1 && c1 == 0 && cout_virtual % 256 == 0),
              "pack_conv_weights: transposed needs ksize=1, c1=0, cout multiple of 64");
  SFH_REQUIRE(transposed != 2 || (ksize == 4 && c1 == 0 && c0 % 16 == 0 && aux > 0 && aux <= c0 / 4),
              "pack_conv_weights: stem mode needs ksize=4, c0 = 4*padded cin, aux = real cin");
  SFH_REQUIRE(c1 == 0 || c0 % (16 * nsub_for(ksize)) == 0,
              "pack_conv_weights: c0 must be a multiple of the stage width when c1 > 0");
  const long total4 = n / 4;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, packed, ksize, nsub_for(ksize), c0, c1, cout_virtual,
                     transposed, aux, total4);
  return sfh_check_launch("pack_weights_kernel");
}

extern "C" int sfh_pack_c4_weights(const float* w, float* packed, int cin, int cout, void* stream) {
  SFH_REQUIRE(w && packed && cin >= 1 && cin <= 4 && cout > 0 && cout % 64 == 0,
              "pack_c4_weights: needs 1..4 input channels and a multiple of 64 output channels");
  const int total = (cout / 64) * 9 * 256;
  hipLaunchKernelGGL(pack_c4_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, packed, cin, cout, total);
  return sfh_check_launch("pack_c4_weights_kernel");
}

extern "C" int sfh_conv3x3_c4_fwd(const sfh_conv_desc* dp, void* stream_) {
  SFH_REQUIRE(dp, "conv3x3_c4_fwd: null descriptor");
  const sfh_conv_desc& d = *dp;
  SFH_REQUIRE(d.src0 && d.wpacked && d.scale && d.shift && d.dst, "conv3x3_c4_fwd: null pointer");
  SFH_REQUIRE(d.ksize == 3 && d.stride == 1 && d.cs0 == 4 && d.c0 <= 4 && !d.src1 && !d.pool0 && !d.dst_pool &&
                  d.src_fmt == SFH_FMT_F32 && d.out_mode == SFH_OUT_NHWC && d.h0 == d.H && d.w0 == d.W,
              "conv3x3_c4_fwd: needs a single fp32 NHWC source with 4 stored channels, 3x3 stride 1");
  SFH_REQUIRE(d.cout > 0 && d.cout % 64 == 0 && d.batch > 0 && d.H > 0 && d.W > 0, "conv3x3_c4_fwd: bad geometry");
  SFH_REQUIRE(d.h2_exp_dst >= -64 && d.h2_exp_dst <= 64, "conv3x3_c4_fwd: h2_exp_dst=%d out of range (-64 .. 64)", d.h2_exp_dst);
  ConvGeom g;
  g.Ho = d.H;
  g.Wo = d.W;
  g.tiles_x = sfh_cdiv(g.Wo, C4Cfg::TW);
  g.rows_per_img = g.Ho + 1 + (g.Ho & 1 ? 0 : 1);  // even rows per frame
  g.rows_total = d.batch * g.rows_per_img;
  g.rows_magic = (unsigned)((1ULL << 32) / (unsigned)g.rows_per_img) + 1u;
  SFH_REQUIRE((unsigned long long)(g.rows_total + 64) * g.rows_per_img < (1ULL << 32), "conv3x3_c4_fwd: too many rows");
  g.tiles_y = sfh_cdiv(g.rows_total, C4Cfg::TH);
  g.ntiles = g.tiles_x * g.tiles_y;
  const unsigned long long b0 = 16ULL * d.batch * d.H * d.W;
  SFH_REQUIRE(b0 < kOOB, "conv3x3_c4_fwd: source exceeds the 4 GiB descriptor range");
  g.bytes0 = (unsigned)b0;
  g.bytes1 = 0;
  g.nblk_n = d.cout / 64;
  const long nblocks = (long)sfh_cdiv(g.ntiles, 8) * 8 * g.nblk_n;
  hipLaunchKernelGGL(conv3x3_c4_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream_, d, g);
  return sfh_check_launch("conv3x3_c4_kernel");
}

extern "C" int sfh_conv_fwd(const sfh_conv_desc* dp, void* stream_) {
  SFH_REQUIRE(dp, "conv_fwd: null descriptor");
  const sfh_conv_desc& d = *dp;
  hipStream_t stream = (hipStream_t)stream_;
  SFH_REQUIRE(d.src0 && d.wpacked && d.scale && d.shift && d.dst, "conv_fwd: null pointer");
  SFH_REQUIRE(d.batch > 0 && d.H > 0 && d.W > 0, "conv_fwd: empty geometry");
  SFH_REQUIRE(d.cout > 0 && d.cout % 64 == 0, "conv_fwd: cout=%d must be a multiple of 64", d.cout);
  SFH_REQUIRE(d.cs0 % 4 == 0 && d.cs0 >= 4 && d.dst_cs % 4 == 0, "conv_fwd: channel strides must be multiples of 4");
  SFH_REQUIRE(d.cs0 < 16 || d.cs0 % 16 == 0, "conv_fwd: cs0=%d must be < 16 or a multiple of 16", d.cs0);
  SFH_REQUIRE(!d.src1 || d.cs1 < 16 || d.cs1 % 16 == 0, "conv_fwd: cs1=%d must be < 16 or a multiple of 16", d.cs1);
  SFH_REQUIRE(d.c0 > 0 && d.c0 <= d.cs0, "conv_fwd: c0=%d cs0=%d", d.c0, d.cs0);
  const int nsub = nsub_for(d.ksize);
  if (d.src1) {
    SFH_REQUIRE(d.c0 % (16 * nsub) == 0, "conv_fwd: c0 must be a multiple of %d with two sources", 16 * nsub);
    SFH_REQUIRE(d.cs1 % 4 == 0 && d.c1 > 0 && d.c1 <= d.cs1, "conv_fwd: c1=%d cs1=%d", d.c1, d.cs1);
    SFH_REQUIRE(d.h1 > 0 && d.w1 > 0 && d.pad_top1 >= 0 && d.pad_left1 >= 0 &&
                    d.pad_top1 + d.h1 <= d.H && d.pad_left1 + d.w1 <= d.W,
                "conv_fwd: source 1 (%dx%d at %d,%d) does not fit the %dx%d frame", d.h1, d.w1,
                d.pad_top1, d.pad_left1, d.H, d.W);
  }
  if (d.pool0)
    SFH_REQUIRE(d.h0 / 2 == d.H && d.w0 / 2 == d.W, "conv_fwd: pool0 needs floor(h0/2)==H, floor(w0/2)==W");
  else
    SFH_REQUIRE(d.h0 == d.H && d.w0 == d.W, "conv_fwd: source 0 is %dx%d, frame is %dx%d", d.h0, d.w0, d.H, d.W);
  SFH_REQUIRE(d.src_fmt == SFH_FMT_F32, "conv_fwd: the fp32 kernel reads fp32 NHWC sources (src_fmt=%d)", d.src_fmt);
  SFH_REQUIRE(!d.dst_pool, "conv_fwd: fused pool output is provided by sfh_conv_s3_fwd only");
  SFH_REQUIRE(d.dst_fmt == SFH_FMT_F32 || d.dst_fmt == SFH_FMT_S3, "conv_fwd: the fp32 kernel writes fp32 or S3 (dst_fmt=%d)", d.dst_fmt);
  if (d.out_mode == SFH_OUT_UPSCATTER2)
    SFH_REQUIRE(d.ksize == 1 && d.stride == 1 && (d.cout / 4) % 64 == 0 && !d.residual,
                "conv_fwd: up-scatter needs ksize=1, stride=1, cout/4 multiple of 64");
  else
    SFH_REQUIRE(d.out_mode == SFH_OUT_NHWC, "conv_fwd: bad out_mode %d", d.out_mode);

#define SFH_CASE(KS, ST, TILE, SH, SW, TH, TW)                                   \
  if (d.ksize == KS && d.stride == ST && d.tile == TILE)                         \
    return launch_conv<ConvCfg<KS, ST, (KS == 1 ? 2 : 1), SH, SW, TH, TW>>(d, stream);
  SFH_CASE(3, 1, SFH_TILE_8x32, 1, 16, 8, 32)
  SFH_CASE(3, 1, SFH_TILE_16x16, 1, 16, 16, 16)
  SFH_CASE(3, 1, SFH_TILE_32x8, 2, 8, 32, 8)
  SFH_CASE(1, 1, SFH_TILE_8x32, 1, 16, 8, 32)
  SFH_CASE(1, 1, SFH_TILE_16x16, 1, 16, 16, 16)
  SFH_CASE(1, 1, SFH_TILE_32x8, 2, 8, 32, 8)
  // 4x4 stem (7x7 s2 over the space-to-depth input)
  SFH_CASE(4, 1, SFH_TILE_8x32, 1, 16, 8, 32)
  SFH_CASE(4, 1, SFH_TILE_16x16, 1, 16, 16, 16)
  SFH_CASE(4, 1, SFH_TILE_32x8, 2, 8, 32, 8)
  // stride 2 (ResNet stage transitions): half-size tiles keep the halo within LDS
  SFH_CASE(3, 2, SFH_TILE_8x32, 1, 16, 4, 32)
  SFH_CASE(3, 2, SFH_TILE_16x16, 1, 16, 8, 16)
  SFH_CASE(3, 2, SFH_TILE_32x8, 2, 8, 16, 8)
  SFH_CASE(1, 2, SFH_TILE_8x32, 1, 16, 4, 32)
  SFH_CASE(1, 2, SFH_TILE_16x16, 1, 16, 8, 16)
  SFH_CASE(1, 2, SFH_TILE_32x8, 2, 8, 16, 8)
#undef SFH_CASE
  sfh_set_error("conv_fwd: unsupported ksize=%d stride=%d tile=%d", d.ksize, d.stride, d.tile);
  return SFH_E_ARG;
}
