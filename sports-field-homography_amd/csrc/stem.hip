// stem.hip - the ResNetSTN stem (7x7, stride 2, padding 3, <= 8 input channels -> 64; models/resnet.py:172,
// 241-243 conv0 + bn1 + relu) on the bf16 matrix cores with the split-bf16 (bf16x6) arithmetic of conv_s3.hip.
//
// The generic kernels see this layer as a 4x4 conv over a space-to-depth copy of the input (K = 16 taps x
// 32 channels, a third of it padding; 0.55 ms per batch on the fp32 MFMA kernel, 1.6 ms on the generic
// split-bf16 one).  Here K is packed by tap: one v_mfma_f32_16x16x32_bf16 covers 4 taps x 8 channels (lane
// group k = lane/16 owns one tap), 13 k-steps for the 49 taps; the 8 channels of a pixel are one 16-byte
// LDS element per plane.  The input (fp32 NHWC, 8 stored channels = cat((logits, frame)) zero-padded) is
// split into its three bf16 planes while it is staged, so no converted copy and no space-to-depth pass exist.
// Workgroup = 8 x 32 output pixels x 64 couts (4 waves: 2 pixel halves x 2 cout halves), halo 21 x 69 input
// pixels = 70 KB of LDS, two workgroups per CU.  Epilogue: conv_epilogue.h (BatchNorm scale/shift, ReLU, fp32 NHWC).
#include "common.h"
#include "conv_epilogue.h"

typedef unsigned int st_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 st_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 st_f16x8 __attribute__((ext_vector_type(8)));

namespace {

struct StemCfg {  // what the shared epilogue needs to know about the tile
  static constexpr int KS = 7, SUBX = 2, SH = 1, SW = 16, TH = 8, TW = 32;
  static constexpr bool FLATROWS = false;
  static constexpr int HH = 2 * (TH - 1) + 7, HW = 2 * (TW - 1) + 7;  // 21 x 69 input pixels
  static constexpr int HPIX = HH * HW, HPIXP = (HPIX + 15) / 16 * 16;
  static constexpr int LDS_BYTES = 3 * HPIXP * 16;   // three planes; the two-plane instance uses two thirds of it
  static constexpr int NSTEP = 13;  // ceil(49 taps / 4 per MFMA)
};

struct StemGeom {
  int Ho, Wo, rows_total, rows_per_img;
  unsigned rows_magic;
  int tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ st_bf16x8 st_bf(const st_u32x4& v) { return __builtin_bit_cast(st_bf16x8, v); }

// NP = 3: the input is split into three bf16 planes, six products (bf16x6); NP = 2: two fp16 planes of v * 2^2, three
// products (f16x3: weights packed as planes of w * 2^wexp, the caller folds 2^-(wexp + 2) into `scale`)
template <int NP>
__global__ __launch_bounds__(256, 2) void stem7x7_kernel(const sfh_conv_desc d, const StemGeom g) {
  using C = StemCfg;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  st_u32x4* const lds = reinterpret_cast<st_u32x4*>(smem_f);   // [3 planes][HPIXP] x 16 B
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv >> 1, wn = wv & 1;
  const int lq = lane & 15, lg = lane >> 4;
  const int tile = blockIdx.x;
  const int tpi = g.tiles_x * g.tiles_y;
  const int img = tile / tpi, tl = tile - img * tpi;
  const int ty = tl / g.tiles_x, tx = tl - ty * g.tiles_x;
  const int y0 = ty * C::TH, x0 = tx * C::TW;

  // ---- stage the input halo: fp32 (8 channels) -> three bf16 planes, zeros outside the frame
  const float* src = d.src0 + (long)img * d.H * d.W * 8;
  unsigned over = 0u;   // NP = 2: see sfh_split4_h2
  const float in_scale = sfh_h2_pow2(d.h2_exp_src);   // the planes carry x * 2^h2_exp_src (folded into `scale` by the caller)
  for (int p = tid; p < C::HPIX; p += 256) {
    const int hy = p / C::HW, hx = p - hy * C::HW;
    const int y = 2 * y0 - 3 + hy, x = 2 * x0 - 3 + hx;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < d.H && x >= 0 && x < d.W) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + ((long)y * d.W + x) * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(src + ((long)y * d.W + x) * 8 + 4);
      v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    if constexpr (NP == 3) {
      st_u32x4 pl[3];
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = v[j];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const unsigned w = sfh_cvt_pk(r[2 * h], r[2 * h + 1]);
          pl[q][h] = w;
          r[2 * h] -= __builtin_bit_cast(float, w << 16);
          r[2 * h + 1] -= __builtin_bit_cast(float, w & 0xFFFF0000u);
        }
        lds[q * C::HPIXP + p] = pl[q];
      }
    } else {
      sfh_u32x2 pa[2], pb[2];
      sfh_split4_h2((f32x4){v[0], v[1], v[2], v[3]}, in_scale, pa, over);
      sfh_split4_h2((f32x4){v[4], v[5], v[6], v[7]}, in_scale, pb, over);
      lds[p] = (st_u32x4){pa[0][0], pa[0][1], pb[0][0], pb[0][1]};
      lds[C::HPIXP + p] = (st_u32x4){pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
    }
  }
  if constexpr (NP == 2) {
    sfh_h2_report(over, d.h2_overflow, d.h2_range);   // (all lanes are back from the staging loop)
  }

  // ---- weights: packed [step 13][plane NP][cout subtile 4][lane 64] x 16 B; this wave: subtiles 2*wn, 2*wn+1
  const st_u32x4* wp = reinterpret_cast<const st_u32x4*>(d.wpacked) + (2 * wn) * 64 + lane;
  auto load_w = [&](st_u32x4 (&w)[NP][2], int s) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) w[p][ni] = wp[((s * NP + p) * 4 + ni) * 64];
  };
  st_u32x4 wa[NP][2], wb[NP][2];
  load_w(wa, 0);

  f32x4 acc[2][8];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // lane's pixel in pixel group mi of this wave: row wm*4 + mi/2, col 16*(mi%2) + lq; halo pixel of tap
  // (ky,kx): (2*row + ky) * HW + 2*col + kx.  The tap of a lane is 4*s + lg.
  const int pix0 = (2 * (wm * 4)) * C::HW + 2 * lq;
  constexpr int NPROD = NP == 3 ? 6 : 3;
  constexpr int PW[6] = {0, 1, NP == 3 ? 2 : 0, 0, 1, 0}, PX[6] = {NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0, 0};
#pragma unroll
  for (int s = 0; s < C::NSTEP; ++s) {
    st_u32x4 (&wc)[NP][2] = (s & 1) ? wb : wa;
    st_u32x4 (&wn_)[NP][2] = (s & 1) ? wa : wb;
    if (s + 1 < C::NSTEP) load_w(wn_, s + 1);
    int t = 4 * s + lg;
    if (t > 48) t = 48;                      // padding taps carry zero weights: any valid address will do
    const int toff = (t / 7) * C::HW + (t % 7);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int moff = (2 * (mi >> 1)) * C::HW + 32 * (mi & 1);
      st_u32x4 xq[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) xq[p] = lds[p * C::HPIXP + pix0 + moff + toff];
#pragma unroll
      for (int k6 = 0; k6 < NPROD; ++k6)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if constexpr (NP == 3)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(st_bf(wc[PW[k6]][ni]), st_bf(xq[PX[k6]]), acc[ni][mi], 0, 0, 0);
          else
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(st_f16x8, wc[PW[k6]][ni]),
                                                                __builtin_bit_cast(st_f16x8, xq[PX[k6]]), acc[ni][mi], 0, 0, 0);
        }
    }
  }
  sfh_conv_epilogue<C, 2, 8>(d, g, acc, 32 * wn, wm * 8, (img << 16) | y0, x0, lq, lg);
}

// packed[s][plane][subtile][lane][j]: cout = subtile*16 + (lane & 15), tap = 4*s + (lane >> 4), channel j
template <int NP>
__global__ void pack_stem_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed, int cin,
                                         int total, float wscale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // one (s, subtile, lane)
  if (idx >= total) return;
  const int lane = idx & 63, sub = (idx >> 6) & 3, s = idx >> 8;
  const int co = sub * 16 + (lane & 15), t = 4 * s + (lane >> 4);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = (t < 49 && j < cin) ? w[((long)co * cin + j) * 49 + t] : 0.f;
    const long base = ((long)(s * NP) * 4 + sub) * 64 + lane;   // plane 0 element index (16-byte units)
    if constexpr (NP == 3) {
      const __bf16 v0 = (__bf16)v;
      const float r1 = v - (float)v0;
      const __bf16 v1 = (__bf16)r1;
      const __bf16 v2 = (__bf16)(r1 - (float)v1);
      packed[(base + 0 * 256) * 8 + j] = __builtin_bit_cast(unsigned short, v0);
      packed[(base + 1 * 256) * 8 + j] = __builtin_bit_cast(unsigned short, v1);
      packed[(base + 2 * 256) * 8 + j] = __builtin_bit_cast(unsigned short, v2);
    } else {
      const float u = fminf(fmaxf(v * wscale, -65504.f), 65504.f);
      const _Float16 h0 = (_Float16)u;
      const _Float16 h1 = (_Float16)(u - (float)h0);
      packed[(base + 0 * 256) * 8 + j] = __builtin_bit_cast(unsigned short, h0);
      packed[(base + 1 * 256) * 8 + j] = __builtin_bit_cast(unsigned short, h1);
    }
  }
}

}  // namespace

extern "C" int64_t sfh_packed_stem_weight_bytes(void) { return (int64_t)StemCfg::NSTEP * 3 * 4 * 64 * 16; }

extern "C" int sfh_pack_stem_weights(const float* w, void* packed, int cin, int fmt, int wexp, void* stream) {
  SFH_REQUIRE(w && packed && cin >= 1 && cin <= 8, "pack_stem_weights: 1..8 input channels");
  SFH_REQUIRE(fmt == SFH_FMT_S3 || (fmt == SFH_FMT_H2 && wexp >= -100 && wexp <= 100), "pack_stem_weights: fmt=%d wexp=%d", fmt, wexp);
  const int total = StemCfg::NSTEP * 4 * 64;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (fmt == SFH_FMT_H2)
    hipLaunchKernelGGL(pack_stem_weights_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)packed, cin,
                       total, ldexpf(1.f, wexp));
  else
    hipLaunchKernelGGL(pack_stem_weights_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)packed, cin,
                       total, 1.f);
  return sfh_check_launch("pack_stem_weights_kernel");
}

extern "C" int sfh_stem7x7_fwd(const sfh_conv_desc* dp, void* stream) {
  SFH_REQUIRE(dp, "stem7x7_fwd: null descriptor");
  const sfh_conv_desc& d = *dp;
  SFH_REQUIRE(d.src0 && d.wpacked && d.scale && d.shift && d.dst, "stem7x7_fwd: null pointer");
  SFH_REQUIRE(d.cs0 == 8 && d.c0 >= 1 && d.c0 <= 8 && d.src_fmt == SFH_FMT_F32, "stem7x7_fwd: fp32 NHWC source with 8 stored channels");
  SFH_REQUIRE(d.cout == 64 && d.dst_cs >= 64 && d.dst_fmt == SFH_FMT_F32 && d.out_mode == SFH_OUT_NHWC && !d.residual &&
                  !d.dst_pool && !d.src1, "stem7x7_fwd: 64 output channels, plain fp32 NHWC destination");
  SFH_REQUIRE(d.batch > 0 && d.batch < 32768 && d.H > 0 && d.W > 0 && d.h0 == d.H && d.w0 == d.W, "stem7x7_fwd: bad geometry");
  StemGeom g;
  g.Ho = (d.H + 6 - 7) / 2 + 1;
  g.Wo = (d.W + 6 - 7) / 2 + 1;
  SFH_REQUIRE(g.Ho < 65536, "stem7x7_fwd: frame too tall");
  g.rows_total = 0; g.rows_per_img = g.Ho; g.rows_magic = 0;
  g.tiles_x = sfh_cdiv(g.Wo, StemCfg::TW);
  g.tiles_y = sfh_cdiv(g.Ho, StemCfg::TH);
  g.ntiles = g.tiles_x * g.tiles_y * d.batch;
  SFH_REQUIRE((unsigned long long)d.batch * g.Ho * g.Wo * d.dst_cs * 4ULL < 0xFFFFFFF0ULL, "stem7x7_fwd: destination exceeds 4 GiB");
  SFH_REQUIRE(d.split_arith == 0 || d.split_arith == SFH_FMT_S3 || d.split_arith == SFH_FMT_H2, "stem7x7_fwd: split_arith=%d", d.split_arith);
  SFH_REQUIRE(d.h2_exp_src >= -64 && d.h2_exp_src <= 64, "stem7x7_fwd: h2_exp_src=%d out of range (-64 .. 64)", d.h2_exp_src);
  if (d.split_arith == SFH_FMT_H2) {
    sfh_allow_big_lds(reinterpret_cast<const void*>(&stem7x7_kernel<2>));
    hipLaunchKernelGGL(stem7x7_kernel<2>, dim3((unsigned)g.ntiles), dim3(256), StemCfg::LDS_BYTES / 3 * 2, (hipStream_t)stream, d, g);
  } else {
    sfh_allow_big_lds(reinterpret_cast<const void*>(&stem7x7_kernel<3>));
    hipLaunchKernelGGL(stem7x7_kernel<3>, dim3((unsigned)g.ntiles), dim3(256), StemCfg::LDS_BYTES, (hipStream_t)stream, d, g);
  }
  return sfh_check_launch("stem7x7_kernel");
}
