// conv_c4h2.hip - the UNet's first layer (DoubleConv conv 0: 3 -> 64 channels, unet/unet_parts.py:15) on the fp16
// matrix cores, "f16x3" arithmetic like every other layer of that mode (csrc/conv_s3.hip).
//
// The layer is bound by its OUTPUT (0.94 GB of H2 activations per batch of 16 at 640x360 against 59 MB of input); the
// fp32-MFMA kernel it replaces (conv3x3_c4_kernel: one v_mfma_f32_16x16x4_f32 per tap, 32 cycles each) kept the matrix pipe
// busy for 0.13 of its 0.29 ms.  Here the K axis of one MFMA is (tap, channel): k = 4 * tap + channel, 36 real values in
// two k-steps of 32, three fp16 products per step - 96 MFMAs of 16 cycles per wave instead of 144 of 32.
//
// Frame tensor ("FH2"): the producer (sfh_frame_to_h2) splits the frame once into two fp16 planes per pixel,
//   16 bytes per pixel = [plane0 of channels 0..3 | plane1 of channels 0..3]     (u = x * 2^e, as SFH_FMT_H2)
// so a lane's B operand for the taps (t, t + 1) is two 16-byte LDS reads and no conversion: plane p of the operand is
// {read(t).p, read(t + 1).p}.  Out-of-frame halo slots and the padding taps 9 .. 15 read zeros (a NaN pixel must not
// reach outputs outside its 3x3 receptive field through a zero weight).
#include <stdlib.h>

#include "common.h"
#include "conv_epilogue.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr unsigned kOOB = 0xFFFFFFF0u;

struct C4HCfg {  // tile geometry seen by the shared epilogue: 8 rows x 32 cols, 1x16 pixel groups
  static constexpr int SUBX = 2, SH = 1, SW = 16, TH = 8, TW = 32, KS = 3;
  static constexpr bool FLATROWS = true;
  static constexpr int HW = 34, HPIX = 10 * 34;
};

struct C4HGeom {
  int tiles_x, ntiles, nblk_n;
  int Ho, Wo, rows_total, rows_per_img;
  unsigned rows_magic;
  unsigned bytes0;
};

__device__ __forceinline__ f16x8 as_hf(const u32x4& v) { return __builtin_bit_cast(f16x8, v); }

__global__ __launch_bounds__(256, 2) void conv3x3_c4h2_kernel(const sfh_conv_desc d, const C4HGeom g) {
  // halo slots + one slot that stays zero (the padding taps and the k-step-1 lanes without a tap read it)
  __shared__ __attribute__((aligned(16))) u32x4 halo[C4HCfg::HPIX + 1];
  constexpr int ZSLOT = C4HCfg::HPIX;
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
  const int x0 = tx * C4HCfg::TW, r0 = ty * C4HCfg::TH;
  const int n0 = nb * 64;

  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.src0), 0, (int)g.bytes0, 0x00020000);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = tid + 256 * i;
    if (p < C4HCfg::HPIX) {
      const int hy = p / C4HCfg::HW, hx = p - hy * C4HCfg::HW;
      const int r = r0 - 1 + hy, x = x0 - 1 + hx;
      unsigned off = kOOB;
      if (r >= 0 && x >= 0 && x < d.W) {
        const int b = (int)__umulhi((unsigned)r, g.rows_magic);
        const int y = r - b * g.rows_per_img;
        if (b < d.batch && y < d.H) off = (unsigned)((b * d.H + y) * d.W + x) * 16u;
      }
      halo[p] = __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)off, 0, 0);
    }
  }
  if (tid == 0) halo[ZSLOT] = (u32x4){0u, 0u, 0u, 0u};
  // weights: packed [nb][k-step 2][plane 2][cout group 4][lane 64][8 x fp16]
  const u32x4* wp = reinterpret_cast<const u32x4*>(d.wpacked) + (size_t)nb * (2 * 2 * 4 * 64) + lane;
  u32x4 wr[2][2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) wr[s][p][ni] = wp[((s * 2 + p) * 4 + ni) * 64];
  __syncthreads();

  // k = 32 * s + 8 * lg + j: taps (8s + 2lg, 8s + 2lg + 1), channel j & 3.  Step 0: taps 2lg, 2lg + 1 (all real);
  // step 1: tap 8 for lg == 0 only, everything else is padding.
  const int ta = 2 * lg, tb = 2 * lg + 1;
  const int oa = (ta / 3) * C4HCfg::HW + ta % 3, ob = (tb / 3) * C4HCfg::HW + tb % 3;
  constexpr int O8 = 2 * C4HCfg::HW + 2;

  f32x4 acc[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // the kept partial products, smallest first (as conv_s3.hip): w0 x1 + w1 x0 + w0 x0
  constexpr int PW[3] = {0, 1, 0}, PX[3] = {1, 0, 0};
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int sg = wv * 4 + mi;
    const int pix = (sg / 2) * C4HCfg::HW + (sg % 2) * 16 + lq;
    const u32x4 a = halo[pix + oa], b = halo[pix + ob];
    const u32x4 c = halo[lg == 0 ? pix + O8 : ZSLOT];
    u32x4 x[2][2];   // [k-step][plane]
    x[0][0] = (u32x4){a[0], a[1], b[0], b[1]};
    x[0][1] = (u32x4){a[2], a[3], b[2], b[3]};
    x[1][0] = (u32x4){c[0], c[1], 0u, 0u};
    x[1][1] = (u32x4){c[2], c[3], 0u, 0u};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_hf(wr[s][PW[q]][ni]), as_hf(x[s][PX[q]]), acc[ni][mi], 0, 0, 0);
  }
  sfh_conv_epilogue<C4HCfg, 4, 4, 2>(d, g, acc, n0, wv * 4, r0, x0, lq, lg);   // H2 (or fp32) destination
}

// packed[nb][s 2][plane 2][ni 4][lane 64][j 8] fp16 planes of w * 2^wexp; cout = nb*64 + ni*16 + (lane & 15),
// k = 32 s + 8 (lane >> 4) + j: tap = k >> 2 (taps 9 .. 15: zero), channel = k & 3 (channels >= cin: zero)
__global__ void pack_c4h2_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed, int cin,
                                         int cout, int total, float wscale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // one (nb, s, ni, lane): 8 values, both planes
  if (idx >= total) return;
  const int lane = idx & 63, ni = (idx >> 6) & 3, s = (idx >> 8) & 1, nb = idx >> 9;
  const int co = nb * 64 + ni * 16 + (lane & 15);
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  u16x8 p0, p1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int kk = 32 * s + 8 * (lane >> 4) + j, tap = kk >> 2, c = kk & 3;
    const float v = (tap < 9 && c < cin && co < cout) ? w[((size_t)(co * cin + c) * 3 + tap / 3) * 3 + tap % 3] : 0.f;
    const float u = fminf(fmaxf(v * wscale, -65504.f), 65504.f);
    const _Float16 h0 = (_Float16)u;
    const _Float16 h1 = (_Float16)(u - (float)h0);
    p0[j] = __builtin_bit_cast(unsigned short, h0);
    p1[j] = __builtin_bit_cast(unsigned short, h1);
  }
  const size_t base = (((size_t)(nb * 2 + s) * 2) * 4 + ni) * 64 + lane;       // in 16-byte units, plane 0
  *reinterpret_cast<u16x8*>(packed + base * 8) = p0;
  *reinterpret_cast<u16x8*>(packed + (base + 4 * 64) * 8) = p1;
}

// (B,C,H,W) fp32 -> fp32 NHWC with 4 stored channels (optional) and the FH2 frame tensor (16 bytes per pixel); every lane
// reaches sfh_h2_report
__global__ __launch_bounds__(256) void frame_to_h2_kernel(const float* __restrict__ src, float* __restrict__ nhwc4,
                                                          u32x4* __restrict__ fh2, int C, int HW, long npix, float scale,
                                                          unsigned* overflow, unsigned* range) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned over = 0u;
  if (p < npix) {
    const long b = p / HW, i = p - b * HW;
    const float* s = src + b * (long)C * HW + i;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = j < C ? s[(long)j * HW] : 0.f;
    if (nhwc4) *reinterpret_cast<f32x4*>(nhwc4 + p * 4) = v;
    sfh_u32x2 pl[2];
    sfh_split4_h2(v, scale, pl, over);
    fh2[p] = (u32x4){pl[0][0], pl[0][1], pl[1][0], pl[1][1]};
  }
  sfh_h2_report(over, overflow, range);
}

}  // namespace

extern "C" int64_t sfh_packed_c4h2_weight_bytes(int cout) { return cout > 0 && cout % 64 == 0 ? (int64_t)(cout / 64) * 16384 : -1; }

extern "C" int sfh_pack_c4h2_weights(const float* w, void* packed, int cin, int cout, int wexp, void* stream) {
  SFH_REQUIRE(w && packed && cin >= 1 && cin <= 4 && cout > 0 && cout % 64 == 0,
              "pack_c4h2_weights: needs 1..4 input channels and a multiple of 64 output channels");
  SFH_REQUIRE(wexp >= -100 && wexp <= 100, "pack_c4h2_weights: wexp=%d out of range", wexp);
  const int total = (cout / 64) * 512;
  hipLaunchKernelGGL(pack_c4h2_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (unsigned short*)packed, cin, cout, total, ldexpf(1.f, wexp));
  return sfh_check_launch("pack_c4h2_weights_kernel");
}

extern "C" int sfh_frame_to_h2(const float* src_nchw, float* dst_nhwc4, void* dst_fh2, int batch, int C, int H, int W,
                               int act_exp, uint32_t* overflow, uint32_t* range, void* stream) {
  SFH_REQUIRE(src_nchw && dst_fh2 && batch > 0 && C >= 1 && C <= 4 && H > 0 && W > 0, "frame_to_h2: bad argument");
  SFH_REQUIRE(act_exp >= -64 && act_exp <= 64, "frame_to_h2: act_exp=%d out of range", act_exp);
  const long npix = (long)batch * H * W;
  hipLaunchKernelGGL(frame_to_h2_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src_nchw,
                     dst_nhwc4, (u32x4*)dst_fh2, C, H * W, npix, ldexpf(1.f, act_exp), overflow, range);
  return sfh_check_launch("frame_to_h2_kernel");
}

extern "C" int sfh_conv3x3_c4h2_fwd(const sfh_conv_desc* dp, void* stream_) {
  SFH_REQUIRE(dp, "conv3x3_c4h2_fwd: null descriptor");
  const sfh_conv_desc& d = *dp;
  SFH_REQUIRE(d.src0 && d.wpacked && d.scale && d.shift && d.dst, "conv3x3_c4h2_fwd: null pointer");
  SFH_REQUIRE(!d.src1 && !d.pool0 && d.ksize == 3 && d.stride == 1 && d.h0 == d.H && d.w0 == d.W && !d.dst_pool &&
                  d.out_mode == SFH_OUT_NHWC && !d.residual && !d.head_w && !d.acc_init && !(d.ksplit > 1) && !d.stats_partial,
              "conv3x3_c4h2_fwd: needs one FH2 frame source (sfh_frame_to_h2), 3x3 stride 1, a plain output");
  SFH_REQUIRE(d.dst_fmt == SFH_FMT_H2 || d.dst_fmt == SFH_FMT_F32, "conv3x3_c4h2_fwd: the destination is H2 or fp32 (dst_fmt=%d)", d.dst_fmt);
  SFH_REQUIRE(d.cout > 0 && d.cout % 64 == 0 && d.batch > 0 && d.H > 0 && d.W > 0, "conv3x3_c4h2_fwd: bad geometry");
  SFH_REQUIRE(d.h2_exp_dst >= -64 && d.h2_exp_dst <= 64, "conv3x3_c4h2_fwd: h2_exp_dst=%d out of range (-64 .. 64)", d.h2_exp_dst);
  SFH_REQUIRE(d.h2_exp_src >= -64 && d.h2_exp_src <= 64, "conv3x3_c4h2_fwd: h2_exp_src=%d out of range (-64 .. 64)", d.h2_exp_src);
  // the kernel hard-codes 16 bytes per source pixel (two fp16 planes of four channels): anything else would be misread silently
  SFH_REQUIRE(d.src_fmt == SFH_FMT_FH2, "conv3x3_c4h2_fwd: src_fmt=%d, expected SFH_FMT_FH2 (the frame tensor of sfh_frame_to_h2; "
              "an fp32 NHWC4 frame belongs to sfh_conv3x3_c4_fwd)", d.src_fmt);
  SFH_REQUIRE(d.c0 >= 1 && d.c0 <= 4 && d.cs0 == 4, "conv3x3_c4h2_fwd: c0=%d cs0=%d, expected 1..4 channels stored as 4", d.c0, d.cs0);
  C4HGeom g;
  g.Ho = d.H;
  g.Wo = d.W;
  g.tiles_x = sfh_cdiv(g.Wo, C4HCfg::TW);
  int zr = 1;
  if ((g.Ho + zr) & 1) ++zr;
  g.rows_per_img = g.Ho + zr;
  g.rows_total = d.batch * g.rows_per_img;
  g.rows_magic = (unsigned)((1ULL << 32) / (unsigned)g.rows_per_img) + 1u;
  SFH_REQUIRE((unsigned long long)(g.rows_total + 64) * g.rows_per_img < (1ULL << 32), "conv3x3_c4h2_fwd: too many rows");
  g.ntiles = g.tiles_x * sfh_cdiv(g.rows_total, C4HCfg::TH);
  const unsigned long long b0 = 16ULL * d.batch * d.H * d.W;
  SFH_REQUIRE(b0 < kOOB, "conv3x3_c4h2_fwd: source exceeds the 4 GiB descriptor range");
  g.bytes0 = (unsigned)b0;
  g.nblk_n = d.cout / 64;
  const long nblocks = (long)sfh_cdiv(g.ntiles, 8) * 8 * g.nblk_n;
  SFH_REQUIRE(nblocks < (1L << 31), "conv3x3_c4h2_fwd: grid too large");
  hipLaunchKernelGGL(conv3x3_c4h2_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream_, d, g);
  return sfh_check_launch("conv3x3_c4h2_kernel");
}
