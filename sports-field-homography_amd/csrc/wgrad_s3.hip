// wgrad_s3.hip - backward-filter (weight gradient) of stride-1 3x3 and 1x1 convolutions on the 16-bit matrix cores.
//
//   raw[m][tap][n_off + n] += sum_{b,y,x} dz[b][y][x][m] * xin[b][y + ky - 1][x + kx - 1][n]        (1x1: one tap, no shift)
//
// Same "bf16x6" arithmetic as the forward kernel (conv_s3.hip): both operands are exact three-plane bf16
// splits of fp32 values (S3 tensors), every product is accumulated from its six partial products >= 2^-16
// in fp32 by v_mfma_f32_16x16x32_bf16.  GEMM view: M = cout (A operand, from dz), N = cin (B operand, from
// the layer input), K = pixels.
//
// The S3 layout keeps 8 channels of one pixel in 16 bytes, i.e. it is K-strided for this product.  gfx950's
// transposing LDS read (ds_read_b64_tr_b16: a 4-row x 16-column block of 16-bit elements delivered
// column-major) turns it into MFMA fragments for free: "rows" are four pixels of a k-step, "columns" the 16
// channels of the fragment, and every lane supplies the address of its own row piece, so the tap shift of
// the input window is just another immediate offset.
//
// Workgroup (4 waves) = 64 couts x 32 cins x 9 taps, looped over a contiguous range of 64-pixel tiles
// (split-K over workgroups, fp32 atomics at the end).  Per tile the dz pixels and the input halo go
// global -> LDS by LDS-DMA in the S3 chunk order ([plane][8-channel group][pixel][16 B]; group stride = 4 mod
// 16 chunks, which makes the transposed reads of a 32-lane half conflict-free); one LDS buffer, two
// workgroups per CU, so that the DMA of one hides under the MFMAs of the other.  Wave w: cin half w&1
// (16 cins), cout half w>>1 (32 couts): 18 accumulator tiles, per 32-pixel k-step 66 transposed reads
// against 108 MFMAs.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

constexpr unsigned kOOB = 0xFFFFFFF0u;

struct WgS3Args {
  const void* dz; int M;               // dz S3 (B, H, M/32, 3, 4, W, 8)
  const void* x; int xc, xh, xw;       // input S3 (B, xh, xc/32, 3, 4, xw, 8)
  int N, pad_top, pad_left;            // N channels used (from channel 0), tensor placed at (pad_top, pad_left)
  int batch, H, W;
  float* raw; int raw_n, n_off;
  int ntx, nty, ntiles, nsplit, tps, mblk, nblk;
  unsigned bytes_dz, bytes_x;
};

__device__ __forceinline__ bf16x8 frag(s16x4 lo, s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// NP = planes per operand: 3 = S3 tensors (bf16, six products), 2 = H2 tensors (fp16 planes of v * 2^SFH_H2_ACT_EXP,
// three products; the accumulators are multiplied by 2^-(2 * SFH_H2_ACT_EXP) before they are added to raw)
// KS = 3: nine taps; KS = 1: one tap (ConvTranspose2d as a 1x1 conv over space_to_depth2(dY), Bottleneck 1x1s) - the same
// tile loop with 12 instead of 108 MFMAs per wave and tile, i.e. bound by the LDS-DMA of the two operands, which is still
// several times the rate of the fp32 kernel these layers ran on
template <int TR, int TW, int NP, int KS = 3>
__global__ __launch_bounds__(256, NP == 2 ? 3 : 2) void wgrad_s3_kernel(const WgS3Args a) {
  static_assert(TR * TW == 64 && (TW == 8 || TW == 16 || TW == 32), "64-pixel tiles, two 32-pixel k-steps");
  static_assert(KS == 3 || KS == 1, "3x3 or 1x1");
  constexpr int NTAP = KS * KS, PADK = KS / 2;
  constexpr int R4 = 4 * NP;                                         // (plane, group) runs per 32-channel block
  constexpr int HWD = TW + KS - 1, HR = TR + KS - 1, HP = HR * HWD; // input halo of a tile
  constexpr int PX = ((HP - 4 + 15) / 16) * 16 + 4;                 // chunks per (plane, group) of the halo image
  constexpr int PD = 68;                                            // ... of the dz image (64 pixels)
  constexpr int XCH = R4 * PX;
  constexpr int NXI = (HP + 63) / 64;                               // DMA instructions per (plane, group) of the halo
  constexpr int RS = 32 / TW;                                       // tile rows per k-step
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];       // [halo image 12 * PX][dz image 24 * PD] chunks

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int nb16 = wv & 1, mh = wv >> 1;

  // blocks of one pixel split run back to back on ONE XCD (workgroups go round-robin to the 8 XCDs): the
  // split's dz / input tiles are fetched into that L2 once, not once per (m, n) block
  const int mn = a.mblk * a.nblk;
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int split = (kq / mn) * 8 + xcd, mni = kq % mn;
  if (split >= a.nsplit) return;
  const int mb64 = mni / a.nblk, nb32 = mni % a.nblk;

  const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dz), 0, (int)a.bytes_dz, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.bytes_x, 0x00020000);
  const int MB = a.M >> 5, CBX = a.xc >> 5;

  // transposed-read byte bases of this lane (k index inside a step: 8g + 4h + q; h and the k-step are immediates)
  const int kk0 = 8 * g + q;
  const unsigned dzb = (unsigned)(((4 * mh + (pp >> 1)) * PD + kk0) * 16 + (pp & 1) * 8 + XCH * 16);
  const unsigned xb = (unsigned)(((2 * nb16 + (pp >> 1)) * PX + (kk0 / TW) * HWD + (kk0 % TW)) * 16 + (pp & 1) * 8);

  f32x4 acc[2][NTAP];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int t = 0; t < NTAP; ++t) acc[mb][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const lds_char_ptr ldsc = (lds_char_ptr)(lds_ptr_t)lds;
  auto tr = [&](unsigned byte_off) -> s16x4 {   // EXEC is all ones here (no divergent control flow around the reads)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ldsc + byte_off));
  };

  const int t_begin = split * a.tps;
  const int t_end = (t_begin + a.tps < a.ntiles) ? t_begin + a.tps : a.ntiles;
  for (int t = t_begin; t < t_end; ++t) {
    const int tx = t % a.ntx, r1 = t / a.ntx;
    const int ty = r1 % a.nty, b = r1 / a.nty;
    const int y0 = ty * TR, x0 = tx * TW;
    // ---- LDS-DMA: 12 (plane, group) runs of the input halo, 24 of dz; wave w takes 3 and 6 of them
    {
      unsigned voff[NXI];
      bool act[NXI];
#pragma unroll
      for (int i = 0; i < NXI; ++i) {
        const int hp = i * 64 + lane;
        const int r = hp / HWD, c = hp - r * HWD;
        const int yy = y0 - PADK + r - a.pad_top, xx = x0 - PADK + c - a.pad_left;
        const bool ok = (unsigned)yy < (unsigned)a.xh && (unsigned)xx < (unsigned)a.xw;
        voff[i] = ok ? (unsigned)((yy * CBX * R4 * a.xw + xx) * 16) : kOOB;
        act[i] = hp < HP;
      }
      const unsigned xsb = (unsigned)((b * a.xh * CBX + nb32) * R4) * (unsigned)a.xw * 16u;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int pg = wv * NP + j;                      // plane * 4 + group
        const unsigned soff = xsb + (unsigned)pg * (unsigned)a.xw * 16u;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
          if (act[i])
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(lds + pg * PX + i * 64), 16, (int)voff[i], (int)soff, 0, 0);
        }
      }
      const int py = lane / TW, px = lane - py * TW;
      const int y = y0 + py, x = x0 + px;
      const unsigned dvoff = (y < a.H && x < a.W) ? (unsigned)((y * MB * R4 * a.W + x) * 16) : kOOB;
      const unsigned dsb = (unsigned)((b * a.H * MB + mb64 * 2) * R4) * (unsigned)a.W * 16u;
#pragma unroll
      for (int j = 0; j < 2 * NP; ++j) {
        const int pg8 = wv * (2 * NP) + j;               // plane * 8 + group of the 64 couts
        const int p = pg8 >> 3, g8 = pg8 & 7;
        const unsigned soff = dsb + (unsigned)(((g8 >> 2) * R4 + p * 4 + (g8 & 3))) * (unsigned)a.W * 16u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rdz, (lds_ptr_t)(lds + XCH + pg8 * PD), 16, (int)dvoff, (int)soff, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- two k-steps of 32 pixels
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[2][NP];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const unsigned ad = dzb + (unsigned)(((p * 8 + 2 * mb) * PD + 32 * s) * 16);
          af[mb][p] = frag(tr(ad), tr(ad + 64));
        }
#pragma unroll
      for (int t9 = 0; t9 < NTAP; ++t9) {
        const int ky = t9 / KS, kx = t9 % KS;
        bf16x8 bfr[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const unsigned ad = xb + (unsigned)(((p * 4) * PX + (s * RS + ky) * HWD + kx) * 16);
          bfr[p] = frag(tr(ad), tr(ad + 64));
        }
        // the kept partial products, smallest first (as in the forward kernel)
        constexpr int NPROD = NP == 3 ? 6 : 3;
        constexpr int PA[6] = {0, 1, NP == 3 ? 2 : 0, 0, 1, 0}, PB[6] = {NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0, 0};
#pragma unroll
        for (int k6 = 0; k6 < NPROD; ++k6)
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            if constexpr (NP == 3)
              acc[mb][t9] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mb][PA[k6]], bfr[PB[k6]], acc[mb][t9], 0, 0, 0);
            else
              acc[mb][t9] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[mb][PA[k6]]),
                                                                  __builtin_bit_cast(f16x8, bfr[PB[k6]]), acc[mb][t9], 0, 0, 0);
          }
      }
    }
    __syncthreads();   // every wave is done with the buffer before the next tile's DMA lands in it
  }

  // ---- split-K: add this workgroup's partial sums
  const int m0 = mb64 * 64 + 32 * mh + 4 * g;
  const int n = a.n_off + nb32 * 32 + 16 * nb16 + (lane & 15);
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int t9 = 0; t9 < NTAP; ++t9)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        unsafeAtomicAdd(a.raw + ((long)(m0 + 16 * mb + r) * NTAP + t9) * a.raw_n + n,
                        NP == 2 ? acc[mb][t9][r] * (1.f / (float)(1 << (2 * SFH_H2_ACT_EXP))) : acc[mb][t9][r]);
}

template <int TR, int TW, int NP, int KS = 3>
int launch(WgS3Args a, hipStream_t stream) {
  constexpr int HP = (TR + KS - 1) * (TW + KS - 1);
  constexpr int PX = ((HP - 4 + 15) / 16) * 16 + 4;
  constexpr int LDS_BYTES = (4 * NP * PX + 8 * NP * 68) * 16;
  a.ntx = sfh_cdiv(a.W, TW);
  a.nty = sfh_cdiv(a.H, TR);
  a.ntiles = a.ntx * a.nty * a.batch;
  const int mn = a.mblk * a.nblk;
  // about three rounds of two workgroups per CU - and a multiple of 8: split s runs on XCD s % 8, so that
  // fewer than 8 (or 12: 8 + 4) splits leave XCDs idle (22x40 layers: 3 splits ran on 3 of the 8 XCDs, 103
  // instead of 250 TFLOP/s-equivalent)
  int nsplit = ((1536 / mn + 7) / 8) * 8;
  // Every split ends with 64 x 32 x 9 fp32 atomics per workgroup: a split of a few tiles spends longer in that tail than
  // in its MFMAs (ResNet layer1, 90x160: 768 splits of 5 tiles = 28 M atomics for 17 GFLOP, 138 us).  At least 16 tiles
  // per split, as long as one round of workgroups (512) is still launched.
  const int by_tiles = ((a.ntiles / 16 + 7) / 8) * 8, one_round = ((512 / mn + 7) / 8) * 8;
  if (nsplit > by_tiles) nsplit = by_tiles > one_round ? by_tiles : (one_round < nsplit ? one_round : nsplit);
  if (nsplit < 8) nsplit = 8;
  if (nsplit > a.ntiles) nsplit = a.ntiles;
  a.tps = sfh_cdiv(a.ntiles, nsplit);
  a.nsplit = sfh_cdiv(a.ntiles, a.tps);
  if (a.nsplit > 8 && a.nsplit % 8 != 0 && a.nsplit < 32) {   // e.g. 13 of 16: shave the tail to a multiple of 8
    const int want = (a.nsplit / 8) * 8;
    const int tps2 = sfh_cdiv(a.ntiles, want);
    if (sfh_cdiv(a.ntiles, tps2) % 8 == 0) { a.tps = tps2; a.nsplit = sfh_cdiv(a.ntiles, tps2); }
  }
  const long nblocks = (long)sfh_cdiv(a.nsplit, 8) * 8 * mn;
  SFH_REQUIRE(nblocks < (1L << 31), "conv_wgrad_s3: grid too large");
  sfh_allow_big_lds(reinterpret_cast<const void*>(&wgrad_s3_kernel<TR, TW, NP, KS>));
  hipLaunchKernelGGL((wgrad_s3_kernel<TR, TW, NP, KS>), dim3((unsigned)nblocks), dim3(256), LDS_BYTES, stream, a);
  return sfh_check_launch("wgrad_s3_kernel");
}

}  // namespace

extern "C" int sfh_conv_wgrad_s3(const void* dz_s3, int M, const void* x_s3, int x_channels, int xh, int xw, int N,
                                 int pad_top, int pad_left, int batch, int H, int W, int ksize, float* raw, int raw_n,
                                 int n_off, int fmt, void* stream) {
  SFH_REQUIRE(fmt == SFH_FMT_S3 || fmt == SFH_FMT_H2, "conv_wgrad_s3: fmt=%d (S3 or H2)", fmt);
  SFH_REQUIRE(ksize == 3 || ksize == 1, "conv_wgrad_s3: ksize=%d (3 or 1)", ksize);
  const unsigned long long bpe = fmt == SFH_FMT_H2 ? 4ULL : 6ULL;
  SFH_REQUIRE(dz_s3 && x_s3 && raw, "conv_wgrad_s3: null pointer");
  SFH_REQUIRE(batch > 0 && H > 0 && W > 0 && xh > 0 && xw > 0, "conv_wgrad_s3: bad geometry");
  SFH_REQUIRE(M > 0 && M % 64 == 0, "conv_wgrad_s3: M=%d must be a multiple of 64", M);
  SFH_REQUIRE(N > 0 && N % 32 == 0 && x_channels % 32 == 0 && x_channels >= N, "conv_wgrad_s3: N=%d of %d channels", N, x_channels);
  SFH_REQUIRE(n_off >= 0 && n_off + N <= raw_n, "conv_wgrad_s3: n_off=%d N=%d raw_n=%d", n_off, N, raw_n);
  SFH_REQUIRE(pad_top >= 0 && pad_left >= 0 && pad_top + xh <= H && pad_left + xw <= W, "conv_wgrad_s3: source does not fit the frame");
  const unsigned long long bdz = bpe * batch * H * W * M, bx = bpe * batch * xh * xw * x_channels;
  SFH_REQUIRE(bdz < kOOB && bx < kOOB, "conv_wgrad_s3: a tensor of %llu bytes exceeds the 4 GiB descriptor range", bdz > bx ? bdz : bx);
  WgS3Args a;
  a.dz = dz_s3; a.M = M; a.x = x_s3; a.xc = x_channels; a.xh = xh; a.xw = xw; a.N = N;
  a.pad_top = pad_top; a.pad_left = pad_left; a.batch = batch; a.H = H; a.W = W;
  a.raw = raw; a.raw_n = raw_n; a.n_off = n_off;
  a.mblk = M / 64; a.nblk = N / 32;
  a.bytes_dz = (unsigned)bdz; a.bytes_x = (unsigned)bx;
  a.ntx = a.nty = a.ntiles = a.nsplit = a.tps = 0;
  // 96-pixel tiles (3x32 / 6x16 / 12x8: three k-steps per LDS fill, so that the MFMAs of one workgroup outlast
  // the DMA wait of its co-resident partner) were measured no faster: 25.7 vs 24.0 ms per step for all launches.
  // tile shape with the least padded area (ties: the widest).  Weighting the area by the rate each shape
  // reaches on full tiles (2x32 250, 4x16 190, 8x8 103 TFLOP/s-equivalent) and so moving the 45x80 and 22x40
  // layers to 2x32 tiles was measured slower overall (31.0 vs 27.8 ms per step for all launches).
  const long c0 = (long)sfh_cdiv(H, 2) * sfh_cdiv(W, 32), c1 = (long)sfh_cdiv(H, 4) * sfh_cdiv(W, 16),
             c2 = (long)sfh_cdiv(H, 8) * sfh_cdiv(W, 8);
  hipStream_t st = (hipStream_t)stream;
#define SFH_WG_PICK(NP_, KS_)                                        \
  do {                                                               \
    if (c0 <= c1 && c0 <= c2) return launch<2, 32, NP_, KS_>(a, st); \
    if (c1 <= c2) return launch<4, 16, NP_, KS_>(a, st);             \
    return launch<8, 8, NP_, KS_>(a, st);                            \
  } while (0)
  if (fmt == SFH_FMT_H2) {
    if (ksize == 1) SFH_WG_PICK(2, 1);
    SFH_WG_PICK(2, 3);
  }
  if (ksize == 1) SFH_WG_PICK(3, 1);
  SFH_WG_PICK(3, 3);
#undef SFH_WG_PICK
}
