// conv_epilogue.h - the epilogue shared by the fp32 (conv_mfma.hip) and split-bf16 (conv_s3.hip)
// convolution kernels:  y = acc*scale + shift (+ residual) (ReLU)  ->  fp32 NHWC or S3 tensor,
// plain / transposed-conv scatter, optional fused 2x2 max-pool output (MaxPool2d(2) of Down,
// unet/unet_parts.py:33, written by the PRODUCER so that consumers never re-read 4x the data).
//
// Accumulator layout (both kernels): acc[ni][mi] holds, for pixel (lane & 15) of pixel group
// msub0+mi, the 4 consecutive couts n0 + ni*16 + 4*(lane >> 4) + {0..3}.
#pragma once
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

typedef unsigned int sfh_u32x2 __attribute__((ext_vector_type(2)));

typedef float sfh_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sfh_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int sfh_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned sfh_cvt_pk(float a, float b) {  // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned, __builtin_convertvector((sfh_f32x2){a, b}, sfh_bf16x2));
}

// three-plane split of 4 fp32 values with packed conversions: out[p] = 4 bf16 (8 bytes) of plane p
__device__ __forceinline__ void sfh_split4(const f32x4& v, sfh_u32x2 (&out)[3]) {
  f32x4 r = v;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    const unsigned w0 = sfh_cvt_pk(r[0], r[1]), w1 = sfh_cvt_pk(r[2], r[3]);
    out[p][0] = w0;
    out[p][1] = w1;
    if (p < 2) {
      r[0] -= __builtin_bit_cast(float, w0 << 16);
      r[1] -= __builtin_bit_cast(float, w0 & 0xFFFF0000u);
      r[2] -= __builtin_bit_cast(float, w1 << 16);
      r[3] -= __builtin_bit_cast(float, w1 & 0xFFFF0000u);
    }
  }
}

// two-plane fp16 split ("H2", include/sfh_amd.h) of 4 fp32 values: u = clamp(v * scale), scale = 2^e the tensor's
// power-of-two factor (2^SFH_H2_ACT_EXP by default: the low plane of ordinary activations stays a NORMAL fp16
// number, 22 significand bits kept), plane0 = f16(u), plane1 = f16(u - plane0), both round-to-nearest-even
// (v_cvt_pk_f16_f32).  Values beyond the fp16 range saturate (a NaN becomes -65504); `over` collects max |u| - or a
// NaN - BEFORE saturation so that the caller can report it (sfh_h2_report): the host then picks a smaller exponent
// for the tensor or, for a non-finite value, repeats the work in a format with fp32's range, where NaN / Inf propagate.
typedef _Float16 sfh_f16x2 __attribute__((ext_vector_type(2)));
constexpr float kSfhH2Scale = (float)(1 << SFH_H2_ACT_EXP), kSfhH2InvScale = 1.f / (float)(1 << SFH_H2_ACT_EXP);
constexpr float kSfhH2Max = 65504.f;
// `over` is the running UNSIGNED-INTEGER maximum of the bit patterns of |u|: ordered like the floats, with Inf and every
// NaN above all finite values, so a NaN sticks (fmaxf would drop it)
__device__ __forceinline__ bool sfh_h2_out_of_range(unsigned over) { return over > 0x477FE000u; }   // bits of 65504.f
// 2^e for a tensor exponent -64 <= e <= 64 (exact: the bit pattern of the power of two)
__device__ __forceinline__ float sfh_h2_pow2(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }

__device__ __forceinline__ unsigned sfh_cvt_pk_h(float a, float b) {  // v_cvt_pk_f16_f32 (RNE)
  return __builtin_bit_cast(unsigned, __builtin_convertvector((sfh_f32x2){a, b}, sfh_f16x2));
}
__device__ __forceinline__ sfh_f32x2 sfh_unpack_h(unsigned w) {
  return __builtin_convertvector(__builtin_bit_cast(sfh_f16x2, w), sfh_f32x2);
}

__device__ __forceinline__ void sfh_split4_h2(const f32x4& v, float scale, sfh_u32x2 (&out)[2], unsigned& over) {
  f32x4 u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t = v[j] * scale;
    const unsigned ab = __builtin_bit_cast(unsigned, t) & 0x7FFFFFFFu;
    over = ab > over ? ab : over;
    u[j] = fminf(fmaxf(t, -kSfhH2Max), kSfhH2Max);
  }
  const unsigned w0 = sfh_cvt_pk_h(u[0], u[1]), w1 = sfh_cvt_pk_h(u[2], u[3]);
  const sfh_f32x2 b0 = sfh_unpack_h(w0), b1 = sfh_unpack_h(w1);
  out[0][0] = w0;
  out[0][1] = w1;
  out[1][0] = sfh_cvt_pk_h(u[0] - b0[0], u[1] - b0[1]);
  out[1][1] = sfh_cvt_pk_h(u[2] - b1[0], u[3] - b1[1]);
}

// End of a kernel that produced H2 values, reached by ALL 64 lanes of the wave (the butterfly reads every lane):
// `over` = the lane's running maximum (sfh_split4_h2).  overflow (optional): OR-ed with 1 when a value was saturated;
// range (optional): atomic max of the wave's maximum - only when it exceeds what the word already holds, so that a
// steady-state launch (the word carries the maximum of the earlier batches) issues no atomic at all.
__device__ __forceinline__ void sfh_h2_report(unsigned over, unsigned* overflow, unsigned* range) {
  if (!overflow && !range) return;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)over, m);
    over = o > over ? o : over;
  }
  if ((threadIdx.x & 63) == 0) {
    if (overflow && sfh_h2_out_of_range(over)) atomicOr(overflow, 1u);
    if (range && over > *reinterpret_cast<volatile unsigned*>(range)) atomicMax(range, over);
  }
}


// lane ^ 1 and lane ^ 8 exchanges of the pooling maxima as DPP moves (v_mov_b32_dpp quad_perm:[1,0,3,2] / row_ror:8): __shfl_xor
// compiles to ds_bpermute_b32, i.e. 32-64 trips per lane through the LDS queue that the co-resident waves' operand reads
// share (the pooled output was 14.6 % of a wave's life in the 64-channel layers, more than the four times larger pass 1)
__device__ __forceinline__ float sfh_dpp_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float sfh_dpp_xor8(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}

// sum over the 16 lanes of a DPP row (lane & 15 = the pixel of a pixel group), result in every lane of the row (fp64)
__device__ __forceinline__ double sfh_dpp_f64(double v, int ctrl) {   // the same DPP move on both halves
  const long long u = __builtin_bit_cast(long long, v);
  int lo = (int)u, hi = (int)(u >> 32);
  switch (ctrl) {   // the control word is an immediate of the instruction
    case 0xB1: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;
    case 0x4E: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;
    case 0x124: lo = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xF, 0xF, true); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, true); break;
  }
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double sfh_row16_sum(double v) {
  v += sfh_dpp_f64(v, 0xB1);
  v += sfh_dpp_f64(v, 0x4E);
  v += sfh_dpp_f64(v, 0x124);
  v += sfh_dpp_f64(v, 0x128);
  return v;
}

// a kernel configuration opts into the BatchNorm-statistics epilogue with `static constexpr bool STATS = true`
template <class CFG, class = void>
struct sfh_cfg_stats { static constexpr bool value = false; };
template <class CFG>
struct sfh_cfg_stats<CFG, decltype((void)CFG::STATS)> { static constexpr bool value = CFG::STATS; };

// a kernel configuration whose tile holds fewer real pixel groups than its waves cover says so with `static constexpr int NGRP`
// (conv_small.hip: 15 groups in a 16-group layout); groups >= NGRP are never stored
template <class CFG, class = void>
struct sfh_cfg_ngrp { static constexpr int value = 1 << 30; };
template <class CFG>
struct sfh_cfg_ngrp<CFG, decltype((void)CFG::NGRP)> { static constexpr int value = CFG::NGRP; };

constexpr unsigned kSfhOOB = 0xFFFFFFF0u;  // byte offset that the buffer range check rejects

// CFG supplies SUBX, SH, SW, FLATROWS; G supplies Ho, Wo, rows_total, rows_per_img, rows_magic.
// All global accesses are buffer loads/stores with 32-bit byte offsets: a pixel outside the frame
// carries kSfhOOB and its stores are dropped by the descriptor's range check (no branches, no
// 64-bit address arithmetic); the plane / cout-group advance rides in the scalar offset.
// FMTS: bit 0 - the kernel may be asked for an S3 destination, bit 1 - for an H2 destination (the other
// format's code is compiled out).
template <class CFG, int NI, int MT, int FMTS = 1, class G>
__device__ __forceinline__ void sfh_conv_epilogue(const sfh_conv_desc& d, const G& g, f32x4 (&acc)[NI][MT],
                                                  int n0, int msub0, int r0, int x0, int lq, int lg,
                                                  size_t dst_byte_off = 0,     // split-K: this workgroup's slab of dst
                                                  unsigned stats_slot = 0u) {  // STATS configurations: the wave's row of stats_partial
  const bool h2 = (FMTS & 2) && d.dst_fmt == SFH_FMT_H2;
  const bool s3 = h2 || ((FMTS & 1) && d.dst_fmt == SFH_FMT_S3);  // a split (plane) format
  const unsigned np4 = h2 ? 8u : 12u;                              // (plane, group) runs per 32-channel block
  unsigned over = 0u;
  // H2 tensors carry v * 2^e with a per-tensor exponent (include/sfh_amd.h): destination / pooled output, residual
  const float h2_dst_scale = sfh_h2_pow2(d.h2_exp_dst), h2_res_inv = sfh_h2_pow2(-d.h2_exp_res);
  // S3 layout (B, H, cs/32, 3 planes, 4 groups of 8 ch, W, 8) bf16: for one image row every
  // (channel block, plane, group) is a contiguous run of W x 16 bytes, so that 16 consecutive pixels
  // of a lane group are 256 contiguous bytes (lane groups lg and lg^1 hold the two 8-byte halves of
  // each 16-byte element) and the consumers' LDS-DMA pieces are contiguous along x.
  const unsigned cs = (unsigned)d.dst_cs;
  constexpr bool UPF = CFG::KS == 2;  // the 2x2 up-scatter conv of the fused Up block
  unsigned wdst = (unsigned)(d.out_mode == SFH_OUT_UPSCATTER2 ? 2 * g.Wo : g.Wo);
  unsigned hdst = (unsigned)(d.out_mode == SFH_OUT_UPSCATTER2 ? 2 * g.Ho : g.Ho);
  if constexpr (UPF) {
    if (d.up_dst_w) wdst = (unsigned)d.up_dst_w;
    if (d.up_dst_h) hdst = (unsigned)d.up_dst_h;
  }
  const unsigned run = wdst * 16u;                            // bytes of one (block, plane, group) run
  const unsigned rowb = s3 ? (cs >> 5) * np4 * run : wdst * cs * 4u;  // bytes per image row
  const unsigned planeb = 4u * run;                           // S3: next plane
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<float*>(reinterpret_cast<char*>(d.dst) + dst_byte_off), 0, (int)kSfhOOB, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(d.residual ? d.residual : d.dst), 0, (int)(d.residual ? kSfhOOB : 0u), 0x00020000);
  f32x4 sc[NI], sh[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int co = n0 + ni * 16 + 4 * lg;
    sc[ni] = *reinterpret_cast<const f32x4*>(d.scale + co);
    sh[ni] = *reinterpret_cast<const f32x4*>(d.shift + co);
  }
  // transposed-conv scatter: the wave's couts [n0, n0 + 16*NI) lie in one (dy,dx) quadrant
  int corel = n0, qd = 0;
  if (d.out_mode == SFH_OUT_UPSCATTER2) {
    const int cr = d.cout >> 2;
    qd = n0 / cr;
    corel = n0 - qd * cr;
  }
  // byte offset of channel (corel + 4*lg) relative to the pixel's row/x position; cout group ni
  // adds ni_off(ni) (+16 channels = +2 groups; corel is a multiple of 32)
  const unsigned c_lane = (unsigned)(corel + 4 * lg);
  const unsigned lane_co = s3 ? ((c_lane >> 5) * np4 + ((c_lane & 31u) >> 3)) * run + ((c_lane >> 2) & 1u) * 8u
                              : c_lane * 4u;
  auto ni_off = [&](int ni) -> unsigned {
    return s3 ? ((unsigned)(ni >> 1) * np4 + (unsigned)(ni & 1) * 2u) * run : (unsigned)ni * 64u;
  };
  const unsigned xb = s3 ? 16u : cs * 4u;  // bytes per pixel step along x
  int pb[MT], py[MT], px[MT];
  // the fp32 residual beside an S3 destination and the border-class shift table exist only for the
  // 2x2 up-scatter conv (fused Up block); every other instance compiles them out
  unsigned voff[MT], rvoff[UPF ? MT : 1], cls[UPF ? MT : 1];
  const bool res_s3 = s3 && !d.residual_f32;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int s = msub0 + mi;
    const int sy = s / CFG::SUBX, sx = s - sy * CFG::SUBX;
    const int oy = sy * CFG::SH + lq / CFG::SW, ox = sx * CFG::SW + lq % CFG::SW;
    const int x = x0 + ox;
    int b, y;
    bool ok = x < g.Wo && s < sfh_cfg_ngrp<CFG>::value;
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
    pb[mi] = b; py[mi] = y; px[mi] = x;
    unsigned rowi, xo;
    if (d.out_mode == SFH_OUT_UPSCATTER2) {
      rowi = (unsigned)b * hdst + (unsigned)(2 * y + (qd >> 1));
      xo = (unsigned)(2 * x + (qd & 1));
      if constexpr (UPF) ok = ok && (unsigned)(2 * y + (qd >> 1)) < hdst && xo < wdst;
    } else {
      rowi = (unsigned)(b * g.Ho + y);
      xo = (unsigned)x;
    }
    voff[mi] = (ok && !d.head_skip_dst) ? rowi * rowb + xo * xb + lane_co : kSfhOOB;
    if constexpr (UPF) {
      // fp32 NHWC residual beside an S3 destination (same pixel, channel stride cs)
      rvoff[mi] = ok ? ((rowi * wdst + xo) * cs + c_lane) * 4u : kSfhOOB;
      // class of the output pixel relative to the up-sampled tensor (2*h0 x 2*w0 inside the destination)
      const unsigned yo = rowi - (unsigned)b * hdst, uh = 2u * (unsigned)d.h0, uw = 2u * (unsigned)d.w0;
      cls[mi] = (yo == 0u ? 0u : (yo == uh - 1u ? 2u : (yo >= uh ? 3u : 1u))) * 4u +
                (xo == 0u ? 0u : (xo == uw - 1u ? 2u : (xo >= uw ? 3u : 1u)));
    }
  }
  constexpr bool STATS = sfh_cfg_stats<CFG>::value;
  // ---- pass 1: finish the values in place and store them
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const unsigned nioff = ni_off(ni);
      f32x4 v = acc[ni][mi];
      f32x4 shv = sh[ni];
      if constexpr (UPF) {
        if (d.shift_border)
          shv = *reinterpret_cast<const f32x4*>(d.shift_border + cls[mi] * (unsigned)d.cout + n0 + ni * 16 + 4 * lg);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] * sc[ni][j] + shv[j];
      if (d.residual) {
        if (res_s3 && h2) {
          if constexpr ((FMTS & 2) != 0) {
            f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 1; p >= 0; --p) {
              const sfh_u32x2 w = __builtin_bit_cast(
                  sfh_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rr_, (int)voff[mi], (int)(nioff + p * planeb), 0));
              const sfh_f32x2 a = sfh_unpack_h(w[0]), b = sfh_unpack_h(w[1]);
              q[0] += a[0]; q[1] += a[1]; q[2] += b[0]; q[3] += b[1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += q[j] * h2_res_inv;
          }
        } else if (res_s3) {
#pragma unroll
          for (int p = 2; p >= 0; --p) {
            const sfh_u32x2 w = __builtin_bit_cast(
                sfh_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rr_, (int)voff[mi], (int)(nioff + p * planeb), 0));
            v[0] += __builtin_bit_cast(float, w[0] << 16);
            v[1] += __builtin_bit_cast(float, w[0] & 0xFFFF0000u);
            v[2] += __builtin_bit_cast(float, w[1] << 16);
            v[3] += __builtin_bit_cast(float, w[1] & 0xFFFF0000u);
          }
        } else {
          // fp32 NHWC residual; beside an S3 destination (residual_f32) it is addressed on its own
          // (same pixel, channel stride cs) - computed here so that no other launch pays for it
          unsigned ro = voff[mi], rn = nioff;
          if (s3) {
            if constexpr (UPF) {
              ro = rvoff[mi];
            } else {
              ro = voff[mi] == kSfhOOB ? kSfhOOB
                                       : (((unsigned)(pb[mi] * g.Ho + py[mi]) * wdst + (unsigned)px[mi]) * cs + c_lane) * 4u;
            }
            rn = (unsigned)ni * 64u;
          }
          const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr_, (int)ro, (int)rn, 0));
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += q[j];
        }
      }
      if (d.relu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sfh_relu(v[j]);
      }
      acc[ni][mi] = v;
      if (h2) {
        if constexpr ((FMTS & 2) != 0) {
          sfh_u32x2 pl[2];
          sfh_split4_h2(v, h2_dst_scale, pl, over);
#pragma unroll
          for (int p = 0; p < 2; ++p)
            __builtin_amdgcn_raw_buffer_store_b64(pl[p], rd, (int)voff[mi], (int)(nioff + p * planeb), 0);
        }
      } else if (s3) {
        sfh_u32x2 pl[3];
        sfh_split4(v, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p)
          __builtin_amdgcn_raw_buffer_store_b64(pl[p], rd, (int)voff[mi], (int)(nioff + p * planeb), 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sfh_u32x4, v), rd, (int)voff[mi], (int)nioff, 0);
      }
    }
  }
  // ---- pass 2: fused MaxPool2d(2) output (floor): rows (y, y+1) x cols (x, x+1), y and x even
  if (d.dst_pool) {
    const int Hp = g.Ho >> 1, Wp = g.Wo >> 1;
    const unsigned pcs = (unsigned)d.pool_cs;
    const unsigned prun = (unsigned)Wp * 16u;
    const unsigned prowb = s3 ? (pcs >> 5) * np4 * prun : (unsigned)Wp * pcs * 4u;
    const unsigned pplaneb = 4u * prun;
    const unsigned pxb = s3 ? 16u : pcs * 4u;
    auto pni_off = [&](int ni) -> unsigned {
      return s3 ? ((unsigned)(ni >> 1) * np4 + (unsigned)(ni & 1) * 2u) * prun : (unsigned)ni * 64u;
    };
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(d.dst_pool, 0, (int)kSfhOOB, 0x00020000);
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      // vertical partner: the pixel group one tile row below (SH == 1) or lane ^ 8 (2x8 groups)
      const int s = msub0 + mi;
      const bool top = CFG::SH == 2 ? true : (((s / CFG::SUBX) & 1) == 0);
      if (!top) continue;
      constexpr int VSTEP = CFG::SH == 2 ? 0 : CFG::SUBX;
      if (CFG::SH == 1 && mi + VSTEP >= MT) continue;
      const int y = py[mi], x = px[mi];
      const bool writer = voff[mi] != kSfhOOB && !(x & 1) && !(y & 1) && (y >> 1) < Hp && (x >> 1) < Wp &&
                          (CFG::SH == 2 ? (lq < 8) : true);
      const unsigned pc = (unsigned)(n0 + 4 * lg);
      const unsigned pv = writer ? (unsigned)(pb[mi] * Hp + (y >> 1)) * prowb + (unsigned)(x >> 1) * pxb +
                                       (s3 ? ((pc >> 5) * np4 + ((pc & 31u) >> 3)) * prun + ((pc >> 2) & 1u) * 8u
                                           : pc * 4u)
                                 : kSfhOOB;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        f32x4 m = acc[ni][mi];
        if (CFG::SH == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] = sfh_max_nan(m[j], sfh_dpp_xor8(m[j]));
        } else {
          const f32x4 o = acc[ni][mi + VSTEP < MT ? mi + VSTEP : mi];
#pragma unroll
          for (int j = 0; j < 4; ++j) m[j] = sfh_max_nan(m[j], o[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = sfh_max_nan(m[j], sfh_dpp_xor1(m[j]));
        const unsigned nioff = pni_off(ni);
        if (h2) {
          if constexpr ((FMTS & 2) != 0) {
            sfh_u32x2 pl[2];
            unsigned dummy = 0u;   // the pooled values are a subset of the values checked above
            sfh_split4_h2(m, h2_dst_scale, pl, dummy);
#pragma unroll
            for (int p = 0; p < 2; ++p)
              __builtin_amdgcn_raw_buffer_store_b64(pl[p], rp, (int)pv, (int)(nioff + p * pplaneb), 0);
          }
        } else if (s3) {
          sfh_u32x2 pl[3];
          sfh_split4(m, pl);
#pragma unroll
          for (int p = 0; p < 3; ++p)
            __builtin_amdgcn_raw_buffer_store_b64(pl[p], rp, (int)pv, (int)(nioff + p * pplaneb), 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(sfh_u32x4, m), rp, (int)pv, (int)nioff, 0);
        }
      }
    }
  }
  // BatchNorm sums of a training-mode layer (sfh_conv_desc.stats_partial; configurations with STATS only).  Forward mode: the
  // sums of z and z * z over the in-frame pixels, one channel at a time from the finished values still in `acc`, in fp64 like
  // the separate pass (sfh_bn_stats) - two accumulators live at a time, so the instance keeps its register count
  if constexpr (STATS) {
    if (d.stats_partial) {
      double* const row = d.stats_partial + (size_t)(stats_slot & (unsigned)(d.stats_rows - 1)) * (size_t)(2 * d.cout);
      if (d.bwd_z) {
        // backward mode (this launch is the backward-data conv whose output dy is the ONLY gradient of a BatchNorm + ReLU
        // layer): with z the pre-BatchNorm tensor of that layer (same fp32 NHWC shape as dst), g = dy * (y > 0),
        // y = (z - mean) * invstd * gamma + beta as bn_apply computes it; the sums of g and of g * xhat (= dbeta, dgamma)
        // - the separate reduction pass over dy and z (sfh_bn_bwd_reduce) is not needed
        const __amdgpu_buffer_rsrc_t rz =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.bwd_z), 0, (int)kSfhOOB, 0x00020000);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int c0 = n0 + ni * 16 + 4 * lg;
          const f32x4 mean = *reinterpret_cast<const f32x4*>(d.bwd_mi + c0);
          const f32x4 inv = *reinterpret_cast<const f32x4*>(d.bwd_mi + d.cout + c0);
          f32x4 gam = {1.f, 1.f, 1.f, 1.f}, bet = {1.f, 1.f, 1.f, 1.f};   // without a ReLU: y > 0 for every element
          if (d.bwd_beta) {
            gam = *reinterpret_cast<const f32x4*>(d.bwd_gamma + c0);
            bet = *reinterpret_cast<const f32x4*>(d.bwd_beta + c0);
          }
          double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) {
            const f32x4 zq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, (int)voff[mi], (int)ni_off(ni), 0));
            const bool in_frame = voff[mi] != kSfhOOB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float xh = (zq[j] - mean[j]) * inv[j];
              const float y = d.bwd_beta ? xh * gam[j] + bet[j] : 1.f;
              const float gj = (in_frame && y > 0.f) ? acc[ni][mi][j] : 0.f;
              a[j] += (double)gj;
              b[j] += (double)gj * (double)xh;
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double sa = sfh_row16_sum(a[j]), sb = sfh_row16_sum(b[j]);
            if (lq == 0) {
              unsafeAtomicAdd(row + c0 + j, sa);
              unsafeAtomicAdd(row + d.cout + c0 + j, sb);
            }
          }
        }
      } else {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
              const double z = voff[mi] != kSfhOOB ? (double)acc[ni][mi][j] : 0.0;
              a += z;
              b += z * z;
            }
            a = sfh_row16_sum(a);
            b = sfh_row16_sum(b);
            if (lq == 0) {
              const int c = n0 + ni * 16 + 4 * lg + j;
              unsafeAtomicAdd(row + c, a);
              unsafeAtomicAdd(row + d.cout + c, b);
            }
          }
      }
    }
  }
  // H2 destination: leave the largest |u| for the host - beyond the fp16 range the value was saturated: the engine
  // lowers the tensor's exponent and re-runs from this layer (non-finite: the batch goes through the three-plane
  // bf16 format, which has fp32's exponent range)
  if constexpr ((FMTS & 2) != 0) {
    if (h2) sfh_h2_report(over, d.h2_overflow, d.h2_range);
  }
}
