// conv_epilogue.h - the epilogue shared by the fp32 (conv_mfma.hip) and split-bf16 (conv_s3.hip)
// convolution kernels:  y = acc*scale + shift (+ residual) (ReLU)  ->  fp32 NHWC or S3 tensor,
// plain / transposed-conv scatter, optional fused 2x2 max-pool output (MaxPool2d(2) of Down,
// unet/unet_parts.py:33, written by the PRODUCER so that consumers never re-read 4x the data).
//
// Accumulator layout (both kernels): acc[ni][mi] holds, for pixel (lane & 15) of pixel group
// msub0+mi, the 4 consecutive couts n0 + ni*16 + 4*(lane >> 4) + {0..3}.
#pragma once
#include "common.h"

typedef unsigned int sfh_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sfh_bf16_bits_to_f32(unsigned short h) {
  return __builtin_bit_cast(float, (unsigned)h << 16);
}

// split 4 fp32 into three bf16 planes and store 8 bytes per plane at element offset `e`
// (plane stride `ps` elements)
__device__ __forceinline__ void sfh_store_s3(unsigned short* __restrict__ base, size_t e, size_t ps,
                                             const f32x4& v) {
  unsigned short h[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 v0 = (__bf16)v[j];
    const float r1 = v[j] - (float)v0;
    const __bf16 v1 = (__bf16)r1;
    const __bf16 v2 = (__bf16)(r1 - (float)v1);
    h[0][j] = __builtin_bit_cast(unsigned short, v0);
    h[1][j] = __builtin_bit_cast(unsigned short, v1);
    h[2][j] = __builtin_bit_cast(unsigned short, v2);
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    sfh_u32x2 w;
    w[0] = (unsigned)h[p][0] | ((unsigned)h[p][1] << 16);
    w[1] = (unsigned)h[p][2] | ((unsigned)h[p][3] << 16);
    *reinterpret_cast<sfh_u32x2*>(base + e + p * ps) = w;
  }
}

__device__ __forceinline__ f32x4 sfh_load_s3(const unsigned short* __restrict__ base, size_t e, size_t ps) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 2; p >= 0; --p) {
    const sfh_u32x2 w = *reinterpret_cast<const sfh_u32x2*>(base + e + p * ps);
    v[0] += __builtin_bit_cast(float, w[0] << 16);
    v[1] += __builtin_bit_cast(float, w[0] & 0xFFFF0000u);
    v[2] += __builtin_bit_cast(float, w[1] << 16);
    v[3] += __builtin_bit_cast(float, w[1] & 0xFFFF0000u);
  }
  return v;
}

// CFG supplies SUBX, SH, SW, FLATROWS; G supplies Ho, Wo, rows_total, rows_per_img, rows_magic.
template <class CFG, int NI, int MT, class G>
__device__ __forceinline__ void sfh_conv_epilogue(const sfh_conv_desc& d, const G& g, f32x4 (&acc)[NI][MT],
                                                  int n0, int msub0, int r0, int x0, int lq, int lg) {
  const bool s3 = d.dst_fmt == SFH_FMT_S3;
  f32x4 sc[NI], sh[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int co = n0 + ni * 16 + 4 * lg;
    sc[ni] = *reinterpret_cast<const f32x4*>(d.scale + co);
    sh[ni] = *reinterpret_cast<const f32x4*>(d.shift + co);
  }
  int pb[MT], py[MT], px[MT];
  bool pok[MT];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int s = msub0 + mi;
    const int sy = s / CFG::SUBX, sx = s - sy * CFG::SUBX;
    const int oy = sy * CFG::SH + lq / CFG::SW, ox = sx * CFG::SW + lq % CFG::SW;
    const int x = x0 + ox;
    int b, y;
    bool ok = x < g.Wo;
    if (CFG::FLATROWS) {
      const int r = r0 + oy;
      b = (int)__umulhi((unsigned)r, g.rows_magic);
      y = r - b * g.rows_per_img;
      ok = ok && r < g.rows_total && y < g.Ho;
    } else {
      b = r0 >> 16;
      y = (r0 & 0xFFFF) + oy;
      ok = ok && y < g.Ho;
    }
    pb[mi] = b; py[mi] = y; px[mi] = x; pok[mi] = ok;
  }
  const size_t cs = (size_t)d.dst_cs;
  // ---- pass 1: finish the values in place and store them
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      int co = n0 + ni * 16 + 4 * lg;
      size_t pix;
      if (d.out_mode == SFH_OUT_UPSCATTER2) {
        const int cr = d.cout >> 2;
        const int qd = co / cr;
        co -= qd * cr;
        pix = ((size_t)(pb[mi] * 2 * g.Ho + 2 * py[mi] + (qd >> 1)) * (2 * g.Wo) + 2 * px[mi] + (qd & 1));
      } else {
        pix = ((size_t)(pb[mi] * g.Ho + py[mi]) * g.Wo + px[mi]);
      }
      f32x4 v = acc[ni][mi];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] * sc[ni][j] + sh[ni][j];
      if (d.residual && pok[mi]) {
        f32x4 rr;
        if (s3)
          rr = sfh_load_s3(reinterpret_cast<const unsigned short*>(d.residual), pix * 3 * cs + co, cs);
        else
          rr = *reinterpret_cast<const f32x4*>(d.residual + pix * cs + co);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += rr[j];
      }
      if (d.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      acc[ni][mi] = v;
      if (pok[mi]) {
        if (s3)
          sfh_store_s3(reinterpret_cast<unsigned short*>(d.dst), pix * 3 * cs + co, cs, v);
        else
          *reinterpret_cast<f32x4*>(d.dst + pix * cs + co) = v;
      }
    }
  }
  // ---- pass 2: fused MaxPool2d(2) output (floor): rows (y, y+1) x cols (x, x+1), y and x even
  if (d.dst_pool) {
    const int Hp = g.Ho >> 1, Wp = g.Wo >> 1;
    const size_t pcs = (size_t)d.pool_cs;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      // vertical partner: the pixel group one tile row below (SH == 1) or lane ^ 8 (2x8 groups)
      const int s = msub0 + mi;
      const bool top = CFG::SH == 2 ? true : (((s / CFG::SUBX) & 1) == 0);
      if (!top) continue;
      constexpr int VSTEP = CFG::SH == 2 ? 0 : CFG::SUBX;
      if (CFG::SH == 1 && mi + VSTEP >= MT) continue;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        f32x4 m = acc[ni][mi];
        if (CFG::SH == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], __shfl_xor(m[j], 8));
        } else {
          const f32x4 o = acc[ni][mi + VSTEP < MT ? mi + VSTEP : mi];
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], o[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], __shfl_xor(m[j], 1));
        const int y = py[mi], x = px[mi];
        const bool writer = pok[mi] && !(x & 1) && !(y & 1) && (y >> 1) < Hp && (x >> 1) < Wp &&
                            (CFG::SH == 2 ? (lq < 8) : true);
        if (writer) {
          const int co = n0 + ni * 16 + 4 * lg;
          const size_t pp = ((size_t)(pb[mi] * Hp + (y >> 1)) * Wp + (x >> 1));
          if (s3)
            sfh_store_s3(reinterpret_cast<unsigned short*>(d.dst_pool), pp * 3 * pcs + co, pcs, m);
          else
            *reinterpret_cast<f32x4*>(d.dst_pool + pp * pcs + co) = m;
        }
      }
    }
  }
}
