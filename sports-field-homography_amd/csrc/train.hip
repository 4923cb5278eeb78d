// train.hip - training-mode kernels of the hot path: batch-statistics
// BatchNorm forward/backward, ReLU/max-pool backward, the data movers of the backward pass and the
// weight-gradient convolution on the fp32 matrix cores.
//
// Replaces, for Reconstructor.forward in train() mode + loss.backward() (train.py:170,233):
//   nn.BatchNorm2d(training=True) fwd/bwd of DoubleConv / BasicBlock (unet/unet_parts.py:16,19,
//   models/resnet.py:67-74), nn.ReLU / nn.MaxPool2d backward, and cuDNN's backward-filter.
// Backward-data convolutions reuse sfh_conv_fwd with weights packed by sfh_pack_conv_weights
// mode 3 / 4.  All activations fp32 NHWC with channel stride == channel count.
//
// Reductions over pixels accumulate in fp64 (ATen's CPU batch-norm uses a double accumulator).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"
#include "conv_epilogue.h"   // sfh_split4_h2: the two-plane fp16 split of the H2 format

namespace {

// ------------------------------------------------------------------ per-channel reductions
// rows = pixels, C channels (multiple of 4).  A block of 256 threads = (256/cq) pixel lanes x cq
// channel quads strides over the pixels (grid-stride, at most RED_BLOCKS blocks so that the fp64
// atomics on the few accumulator addresses stay cheap) and adds its partials to fp64 accumulators;
// C > 1024 is processed in chunks of 1024 channels.
constexpr int RED_ROWS = 256;     // pixels per block below which no further blocks are launched
constexpr int RED_BLOCKS = 1024;  // 4 per CU

static inline unsigned red_grid(long npix) {
  long nb = (npix + RED_ROWS - 1) / RED_ROWS;
  return (unsigned)(nb < 1 ? 1 : (nb > RED_BLOCKS ? RED_BLOCKS : nb));
}

// acc[0][c] += sum z, acc[1][c] += sum z^2
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, long npix, int C,
                                                       double* __restrict__ acc) {
  __shared__ double sh[256];
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int cq = min(1024, C - c0) >> 2;
    const int lanes = 256 / cq;
    const int q = threadIdx.x % cq, pl = threadIdx.x / cq;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (pl < lanes) {
      // every block streams ONE contiguous range of pixels, four pixels per iteration (64 B per thread in
      // flight; one 16-byte load per thread kept the chip at 2 TB/s).  Measured: 3.4-3.5 TB/s on the 0.94 GB tensors, which is what a read-only stream reaches here
      // (grid-stride order and 2048 blocks were no faster).
      const long chunk = (npix + gridDim.x - 1) / gridDim.x;
      const long pend = min(npix, (long)(blockIdx.x + 1) * chunk);
      const long stride = lanes;
      long p = (long)blockIdx.x * chunk + pl;
      const float* zc = z + c0 + 4 * q;
      for (; p + 3 * stride < pend; p += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(zc + (p + u * stride) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s0[j] += (double)v[u][j];
            s1[j] += (double)v[u][j] * (double)v[u][j];
          }
      }
      for (; p < pend; p += stride) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(zc + p * C);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s0[j] += (double)v[j];
          s1[j] += (double)v[j] * (double)v[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __syncthreads();
      sh[threadIdx.x] = (pl < lanes) ? (j < 4 ? s0[j] : s1[j - 4]) : 0.0;
      __syncthreads();
      if (threadIdx.x < cq) {
        double t = 0.0;
        for (int l = 0; l < lanes; ++l) t += sh[l * cq + threadIdx.x];
        unsafeAtomicAdd(&acc[(long)(j >> 2) * C + c0 + 4 * threadIdx.x + (j & 3)], t);
      }
    }
  }
}

// acc[k] += sum_r partial[r][k], k < 2 * C: the per-wave fp64 sums a convolution's epilogue left (sfh_conv_desc.stats_partial).
// Block = 64 columns x 4 row lanes; blockIdx.y strides over the rows.
__global__ __launch_bounds__(256) void bn_stats_partials_kernel(const double* __restrict__ partial, int rows, int ncol,
                                                                double* __restrict__ acc) {
  __shared__ double sh[256];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  double s = 0.0;
  if (col < ncol)
    for (int r = blockIdx.y * 4 + rl; r < rows; r += gridDim.y * 4) s += partial[(long)r * ncol + col];
  sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 64 && col < ncol)
    unsafeAtomicAdd(&acc[col], sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
}

// acc[c] += sum x[:, c] over a channel slice (cs >= C)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long npix, int C, int cs,
                                                     double* __restrict__ acc) {
  // one channel quad per thread column; supports cs >= C (channel slices)
  const int cq = C >> 2;
  for (int q0 = 0; q0 < cq; q0 += 256) {
    const int qn = min(256, cq - q0);
    const int lanes = 256 / qn;
    const int q = threadIdx.x % qn, pl = threadIdx.x / qn;
    double s[4] = {0, 0, 0, 0};
    if (pl < lanes)
      for (long p = (long)blockIdx.x * lanes + pl; p < npix; p += (long)gridDim.x * lanes) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * cs + 4 * (q0 + q));
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += (double)v[j];
      }
    __shared__ double sh[256];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __syncthreads();
      sh[threadIdx.x] = (pl < lanes) ? s[j] : 0.0;
      __syncthreads();
      if (threadIdx.x < qn) {
        double t = 0.0;
        for (int l = 0; l < lanes; ++l) t += sh[l * qn + threadIdx.x];
        unsafeAtomicAdd(&acc[4 * (q0 + threadIdx.x) + j], t);
      }
    }
  }
}

__global__ void bn_finalize_kernel(const double* __restrict__ acc, long npix, int C, float eps, float momentum,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean_invstd, long* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;   // nn.BatchNorm2d's counter (one launch less per layer)
  if (c >= C) return;
  const double n = (double)npix;
  const double mean = acc[c] / n;
  double var = acc[C + c] / n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  mean_invstd[c] = (float)mean;
  mean_invstd[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = npix > 1 ? var * n / (n - 1.0) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
  }
}

// bn_stats_partials + bn_finalize in ONE launch (round 6: a training step had 54 such pairs): a workgroup of 16 row lanes x 64
// channels sums the (rows, 2, C) table of per-wave sums a conv epilogue left - in a fixed order, no atomics - and finishes
// mean / invstd / running statistics for its 64 channels.
__global__ __launch_bounds__(1024) void bn_finalize_partials_kernel(const double* __restrict__ partial, int rows, long npix,
                                                                    int C, float eps, float momentum,
                                                                    float* __restrict__ running_mean,
                                                                    float* __restrict__ running_var,
                                                                    float* __restrict__ mean_invstd,
                                                                    long* __restrict__ num_batches_tracked) {
  __shared__ double sh[2][16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C)
    for (int r = rl; r < rows; r += 16) {
      s0 += partial[(long)r * 2 * C + c];
      s1 += partial[(long)r * 2 * C + C + c];
    }
  sh[0][rl][cl] = s0;
  sh[1][rl][cl] = s1;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
  if (rl != 0 || c >= C) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    s0 += sh[0][k][cl];
    s1 += sh[1][k][cl];
  }
  const double n = (double)npix;
  const double mean = s0 / n;
  double var = s1 / n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  mean_invstd[c] = (float)mean;
  mean_invstd[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = npix > 1 ? var * n / (n - 1.0) : var;
    running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
    running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
  }
}

// y = [relu]( (z - mean) * invstd * gamma + beta [+ residual] )
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ mi,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ residual, int relu, long total4,
                                                       int C, float* __restrict__ y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c = (int)((i * 4) % C);
  const f32x4 v = reinterpret_cast<const f32x4*>(z)[i];
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xh = (v[j] - mi[c + j]) * mi[C + c + j];
    o[j] = xh * gamma[c + j] + beta[c + j];
  }
  if (residual) {
    const f32x4 r = reinterpret_cast<const f32x4*>(residual)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += r[j];
  }
  if (relu) {
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = sfh_relu(o[j]);
  }
  reinterpret_cast<f32x4*>(y)[i] = o;
}

// g = dy * (y > 0 if relu); sums: [0] = sum g, [1] = sum g * xhat
// relu with y == nullptr: the layer has no residual, so the ReLU decision y > 0 is recomputed from z with the
// arithmetic of bn_apply ((z - mean) * invstd * gamma + beta, same operation order => same bits) instead of
// reading y: 8 instead of 12 bytes per element.
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ z, const float* __restrict__ mi,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int relu, long npix, int C, double* __restrict__ acc) {
  const bool sign_from_z = relu && !y;
  __shared__ double sh[256];
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int cq = min(1024, C - c0) >> 2;
    const int lanes = 256 / cq;
    const int q = threadIdx.x % cq, pl = threadIdx.x / cq;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (pl < lanes) {
      const long stride = (long)gridDim.x * lanes;
      long p = (long)blockIdx.x * lanes + pl;
      float mean[4], invstd[4], gam[4] = {1.f, 1.f, 1.f, 1.f}, bet[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mean[j] = mi[c0 + 4 * q + j];
        invstd[j] = mi[C + c0 + 4 * q + j];
        if (sign_from_z) {
          gam[j] = gamma[c0 + 4 * q + j];
          bet[j] = beta[c0 + 4 * q + j];
        }
      }
      auto add = [&](const f32x4& g, const f32x4& zz, const f32x4& yy) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float yv = sign_from_z ? (zz[j] - mean[j]) * invstd[j] * gam[j] + bet[j] : yy[j];
          const float gj = yv > 0.f ? g[j] : 0.f;
          const float xh = (zz[j] - mean[j]) * invstd[j];
          s0[j] += (double)gj;
          s1[j] += (double)gj * (double)xh;
        }
      };
      // two pixels per iteration: six 16-byte loads per thread in flight
      for (; p + stride < npix; p += 2 * stride) {
        const long o0 = p * C + c0 + 4 * q, o1 = (p + stride) * C + c0 + 4 * q;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(dy + o0), g1 = *reinterpret_cast<const f32x4*>(dy + o1);
        const f32x4 z0 = *reinterpret_cast<const f32x4*>(z + o0), z1 = *reinterpret_cast<const f32x4*>(z + o1);
        f32x4 y0 = {1.f, 1.f, 1.f, 1.f}, y1 = {1.f, 1.f, 1.f, 1.f};
        if (relu && y) {
          y0 = *reinterpret_cast<const f32x4*>(y + o0);
          y1 = *reinterpret_cast<const f32x4*>(y + o1);
        }
        add(g0, z0, y0);
        add(g1, z1, y1);
      }
      for (; p < npix; p += stride) {
        const long o = p * C + c0 + 4 * q;
        f32x4 yy = {1.f, 1.f, 1.f, 1.f};
        if (relu && y) yy = *reinterpret_cast<const f32x4*>(y + o);
        add(*reinterpret_cast<const f32x4*>(dy + o), *reinterpret_cast<const f32x4*>(z + o), yy);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __syncthreads();
      sh[threadIdx.x] = (pl < lanes) ? (j < 4 ? s0[j] : s1[j - 4]) : 0.0;
      __syncthreads();
      if (threadIdx.x < cq) {
        double t = 0.0;
        for (int l = 0; l < lanes; ++l) t += sh[l * cq + threadIdx.x];
        unsafeAtomicAdd(&acc[(long)(j >> 2) * C + c0 + 4 * threadIdx.x + (j & 3)], t);
      }
    }
  }
}

// dz = gamma * invstd * (g - sum_g / N - xhat * sum_gx / N); optionally dres = g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ z, const float* __restrict__ mi,
                                                           const float* __restrict__ gamma, const double* __restrict__ acc,
                                                           int relu, long npix, long total4, int C,
                                                           float* __restrict__ dz, float* __restrict__ dres,
                                                           const float* __restrict__ beta, float* __restrict__ acc_f32) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (acc_f32 && blockIdx.x == 0)   // dbeta | dgamma as float32 for the caller (one conversion launch less per layer)
    for (int k = threadIdx.x; k < 2 * C; k += 256) acc_f32[k] = (float)acc[k];
  if (i >= total4) return;
  const int c = (int)((i * 4) % C);
  const f32x4 g4 = reinterpret_cast<const f32x4*>(dy)[i];
  const f32x4 zz = reinterpret_cast<const f32x4*>(z)[i];
  f32x4 yy = {1.f, 1.f, 1.f, 1.f};
  if (relu && y) yy = reinterpret_cast<const f32x4*>(y)[i];
  if (relu && !y) {   // sign recomputed from z (see bn_bwd_reduce_kernel)
#pragma unroll
    for (int j = 0; j < 4; ++j) yy[j] = (zz[j] - mi[c + j]) * mi[C + c + j] * gamma[c + j] + beta[c + j];
  }
  const float inv_n = 1.0f / (float)npix;
  f32x4 o, gg;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float g = yy[j] > 0.f ? g4[j] : 0.f;
    const float invstd = mi[C + c + j];
    const float xh = (zz[j] - mi[c + j]) * invstd;
    const float mg = (float)acc[c + j] * inv_n, mgx = (float)acc[C + c + j] * inv_n;
    o[j] = gamma[c + j] * invstd * (g - mg - xh * mgx);
    gg[j] = g;
  }
  reinterpret_cast<f32x4*>(dz)[i] = o;
  if (dres) reinterpret_cast<f32x4*>(dres)[i] = gg;
}

// ------------------------------------------------------------------ BatchNorm kernels with an S3 copy
// Same arithmetic as bn_apply / bn_bwd_apply, with the thread mapping of f32_to_s3 (one thread = 8
// channels of one pixel, a wave = 16 pixels x the 4 groups of a 32-channel block) so that the result is
// written twice in one pass: fp32 NHWC (for BatchNorm backward and the backward-filter kernel) and the
// split-bf16 S3 layout (B*H, C/32, 3, 4, W, 8) that the next convolution reads.  C % 32 == 0.
typedef unsigned short tr_u16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void tr_split8(const float (&v)[8], unsigned short* __restrict__ dst, long e, long ps) {
  tr_u16x8 p0, p1, p2;
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
  *reinterpret_cast<tr_u16x8*>(dst + e) = p0;
  *reinterpret_cast<tr_u16x8*>(dst + e + ps) = p1;
  *reinterpret_cast<tr_u16x8*>(dst + e + 2 * ps) = p2;
}

__device__ __forceinline__ long tr_s3_elem(long row, int x, int c, int W, int C, int np = 3) {
  return ((((row * (C >> 5) + (c >> 5)) * np) * 4 + ((c & 31) >> 3)) * W + x) * 8;
}

// the same 8 values as two fp16 planes (H2 format, include/sfh_amd.h); `over`: see sfh_split4_h2
__device__ __forceinline__ void tr_split8_h2(const float (&v)[8], unsigned short* __restrict__ dst, long e, long ps,
                                             unsigned& over) {
  sfh_u32x2 pa[2], pb[2];
  sfh_split4_h2((f32x4){v[0], v[1], v[2], v[3]}, kSfhH2Scale, pa, over);
  sfh_split4_h2((f32x4){v[4], v[5], v[6], v[7]}, kSfhH2Scale, pb, over);
  typedef unsigned int tr_u32x4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<tr_u32x4*>(dst + e) = (tr_u32x4){pa[0][0], pa[0][1], pb[0][0], pb[0][1]};
  *reinterpret_cast<tr_u32x4*>(dst + e + ps) = (tr_u32x4){pa[1][0], pa[1][1], pb[1][0], pb[1][1]};
}

// Thread mapping of the two kernels below: a workgroup = 64 consecutive pixels (of the flattened (rows, W) tensor) x one
// 32-channel block, wave g of it = channel group g (8 channels): the per-channel terms are wave-uniform (scalar loads, no
// vector-memory traffic beside the tensors themselves) and a wave's split-layout store is one 1 KB run per plane.  The
// fp32 copy (y / dz) is optional: a layer whose only consumers read the split copy skips a third of its traffic.
__device__ __forceinline__ bool tr_block_thread(long npix, int W, int C, long& p, long& row, int& x, int& c0) {
  const int nb = C >> 5;
  const int cb = (int)(blockIdx.x % (unsigned)nb);
  const long chunk = (long)(blockIdx.x / (unsigned)nb);
  const int g = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  c0 = cb * 32 + g * 8;
  p = chunk * 64 + (threadIdx.x & 63);
  const unsigned r32 = (unsigned)p / (unsigned)W;   // the launchers require npix < 2^31
  row = (long)r32;
  x = (int)((unsigned)p - r32 * (unsigned)W);
  return p < npix;
}

template <int NP>   // planes of the split copy: 3 = S3 (bf16), 2 = H2 (fp16)
__global__ __launch_bounds__(256) void bn_apply_s3_kernel(const float* __restrict__ z, const float* __restrict__ mi,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ residual, int relu, long npix, int W,
                                                          int C, float* __restrict__ y,
                                                          unsigned short* __restrict__ y_s3, unsigned* __restrict__ overflow) {
  long p, row; int x, c0;
  const bool live = tr_block_thread(npix, W, C, p, row, x, c0);
  float mean[8], invstd[8], gam[8], bet[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mean[j] = mi[c0 + j];
    invstd[j] = mi[C + c0 + j];
    gam[j] = gamma[c0 + j];
    bet[j] = beta[c0 + j];
  }
  if (!live) return;
  const long o = p * C + c0;
  float v[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 zz = *reinterpret_cast<const f32x4*>(z + o + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * h + j] = (zz[j] - mean[4 * h + j]) * invstd[4 * h + j] * gam[4 * h + j] + bet[4 * h + j];
    if (residual) {
      const f32x4 r = *reinterpret_cast<const f32x4*>(residual + o + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[4 * h + j] += r[j];
    }
  }
  if (relu) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = sfh_relu(v[j]);
  }
  if (y) {
    *reinterpret_cast<f32x4*>(y + o) = (f32x4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(y + o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
  }
  if constexpr (NP == 3) {
    tr_split8(v, y_s3, tr_s3_elem(row, x, c0, W, C), 4L * W * 8);
  } else {
    unsigned over = 0u;
    tr_split8_h2(v, y_s3, tr_s3_elem(row, x, c0, W, C, 2), 4L * W * 8, over);
    if (overflow && sfh_h2_out_of_range(over)) atomicOr(overflow, 1u);
  }
}

template <int NP>
__global__ __launch_bounds__(256) void bn_bwd_apply_s3_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              const float* __restrict__ z, const float* __restrict__ mi,
                                                              const float* __restrict__ gamma, const double* __restrict__ acc,
                                                              int relu, long npix, int W, int C,
                                                              float* __restrict__ dz, float* __restrict__ dres,
                                                              unsigned short* __restrict__ dz_s3, unsigned* __restrict__ overflow,
                                                              const float* __restrict__ beta, float* __restrict__ acc_f32) {
  if (acc_f32 && blockIdx.x == 0)   // dbeta | dgamma as float32 for the caller (one conversion launch less per layer)
    for (int k = threadIdx.x; k < 2 * C; k += 256) acc_f32[k] = (float)acc[k];
  long p, row; int x, c0;
  const bool live = tr_block_thread(npix, W, C, p, row, x, c0);
  const float inv_n = 1.0f / (float)npix;
  const bool sign_from_z = relu && !y;
  float mean[8], invstd[8], gam[8], bet[8], mg[8], mgx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mean[j] = mi[c0 + j];
    invstd[j] = mi[C + c0 + j];
    gam[j] = gamma[c0 + j];
    bet[j] = sign_from_z ? beta[c0 + j] : 0.f;
    mg[j] = (float)acc[c0 + j] * inv_n;
    mgx[j] = (float)acc[C + c0 + j] * inv_n;
  }
  if (!live) return;
  const long o = p * C + c0;
  float v[8], gg[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 g4 = *reinterpret_cast<const f32x4*>(dy + o + 4 * h);
    const f32x4 zz = *reinterpret_cast<const f32x4*>(z + o + 4 * h);
    f32x4 yy = {1.f, 1.f, 1.f, 1.f};
    if (relu && y) yy = *reinterpret_cast<const f32x4*>(y + o + 4 * h);
    if (sign_from_z) {   // sign recomputed from z (see bn_bwd_reduce_kernel)
#pragma unroll
      for (int j = 0; j < 4; ++j) yy[j] = (zz[j] - mean[4 * h + j]) * invstd[4 * h + j] * gam[4 * h + j] + bet[4 * h + j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = 4 * h + j;
      const float g = yy[j] > 0.f ? g4[j] : 0.f;
      const float xh = (zz[j] - mean[k]) * invstd[k];
      v[k] = gam[k] * invstd[k] * (g - mg[k] - xh * mgx[k]);
      gg[k] = g;
    }
  }
  if (dz) {
    *reinterpret_cast<f32x4*>(dz + o) = (f32x4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(dz + o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
  }
  if (dres) {
    *reinterpret_cast<f32x4*>(dres + o) = (f32x4){gg[0], gg[1], gg[2], gg[3]};
    *reinterpret_cast<f32x4*>(dres + o + 4) = (f32x4){gg[4], gg[5], gg[6], gg[7]};
  }
  if constexpr (NP == 3) {
    tr_split8(v, dz_s3, tr_s3_elem(row, x, c0, W, C), 4L * W * 8);
  } else {
    unsigned over = 0u;
    tr_split8_h2(v, dz_s3, tr_s3_elem(row, x, c0, W, C, 2), 4L * W * 8, over);
    if (overflow && sfh_h2_out_of_range(over)) atomicOr(overflow, 1u);
  }
}

// ------------------------------------------------------------------ ConvTranspose2d(2, stride 2) backward, first pass
// du (B, 2h, 2w, cout) fp32 -> s (B, h, w, 4 * cout) in the split format ONLY, s[(py * 2 + px) * cout + co] at (y, x) =
// du[2y + py][2x + px][co] (the 1x1 backward-data conv and the backward-filter kernel of the transposed conv read nothing
// else), and acc[co] += sum over pixels of du[..., co] - the bias gradient - from the same read.  Replaces colsum +
// space_to_depth2 + f32_to_h2 (three passes, 20 bytes per element) by one of 8.  Mapping as tr_block_thread, but a block
// strides over many 64-pixel chunks so that its column sums end in one fp64 atomic per channel and wave.
template <int NP>
__global__ __launch_bounds__(256) void s2d_split_colsum_kernel(const float* __restrict__ du, int h, int w, int cout,
                                                               long npix, long nchunks, int chunk_stride,
                                                               unsigned short* __restrict__ s_split,
                                                               double* __restrict__ acc, int acc_rows,
                                                               unsigned* __restrict__ overflow) {
  const int C4 = 4 * cout, nb = C4 >> 5;
  const int cb = (int)(blockIdx.x % (unsigned)nb);
  const int g = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int c0 = cb * 32 + g * 8;
  const int par = c0 / cout, co = c0 - par * cout, py = par >> 1, px = par & 1;
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned over = 0u;
  // four chunks per iteration, all eight 16-byte loads of a lane issued before the first value is used
  for (long chunk = (long)(blockIdx.x / (unsigned)nb); chunk < nchunks; chunk += 4L * chunk_stride) {
    f32x4 a[4], bq[4];
    unsigned rowu[4];
    int xu[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long p = (chunk + (long)u * chunk_stride) * 64 + (threadIdx.x & 63);
      ok[u] = chunk + (long)u * chunk_stride < nchunks && p < npix;
      const unsigned pu = ok[u] ? (unsigned)p : 0u;             // (npix < 2^31: checked by the launcher)
      rowu[u] = pu / (unsigned)w;                               // b * h + y
      xu[u] = (int)(pu - rowu[u] * (unsigned)w);
      const unsigned b = rowu[u] / (unsigned)h, y = rowu[u] - b * (unsigned)h;
      const float* sp = du + (((long)b * 2 * h + 2 * y + py) * (2L * w) + 2 * xu[u] + px) * cout + co;
      a[u] = bq[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ok[u]) {
        a[u] = *reinterpret_cast<const f32x4*>(sp);
        bq[u] = *reinterpret_cast<const f32x4*>(sp + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      const float v[8] = {a[u][0], a[u][1], a[u][2], a[u][3], bq[u][0], bq[u][1], bq[u][2], bq[u][3]};
#ifndef SFH_S2D_NOSUM   // (experiment build: without the column sums)
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += (double)v[j];
#endif
      if constexpr (NP == 3) tr_split8(v, s_split, tr_s3_elem((long)rowu[u], xu[u], c0, w, C4), 4L * w * 8);
      else tr_split8_h2(v, s_split, tr_s3_elem((long)rowu[u], xu[u], c0, w, C4, 2), 4L * w * 8, over);
    }
  }
  if constexpr (NP == 2)
    if (overflow && sfh_h2_out_of_range(over)) atomicOr(overflow, 1u);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double t = s[j];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
    // (same-address fp64 atomics are what this kernel would otherwise wait for: 4096 blocks on ONE row of 64 addresses ran
    // at 3.5 TB/s, on 32 rows at the 5 TB/s of the plain BatchNorm pass - profiles/r05_s2d_atomics.txt)
    if ((threadIdx.x & 63) == 0)
      unsafeAtomicAdd(&acc[(size_t)((blockIdx.x / (unsigned)nb) % (unsigned)acc_rows) * (size_t)cout + co + j], t);
  }
}

// ------------------------------------------------------------------ max-pool 2x2 (floor)
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, int H, int W, int C,
                                                           int Ho, int Wo, long total4, float* __restrict__ y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int cq = C >> 2;
  const int q = (int)(i % cq);
  long r = i / cq;
  const int xo = (int)(r % Wo); r /= Wo;
  const int yo = (int)(r % Ho);
  const long b = r / Ho;
  const float* p = x + ((b * H + 2 * yo) * (long)W + 2 * xo) * C + 4 * q;
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), bq = *reinterpret_cast<const f32x4*>(p + C);
  const f32x4 c = *reinterpret_cast<const f32x4*>(p + (long)W * C), d = *reinterpret_cast<const f32x4*>(p + (long)W * C + C);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = sfh_max_nan(sfh_max_nan(a[j], bq[j]), sfh_max_nan(c[j], d[j]));
  reinterpret_cast<f32x4*>(y)[i] = o;
}

// dx[b,y,x,c] (+)= dy[b,y/2,x/2,c] where (y,x) is the FIRST maximum of its window in scan order
// (ATen's max_pool2d keeps the first index), 0 elsewhere (incl. the floor-cropped border).
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           int H, int W, int C, int Ho, int Wo, long total4,
                                                           int accumulate, float* __restrict__ dx) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per pooled output quad
  if (i >= total4) return;
  const int cq = C >> 2;
  const int q = (int)(i % cq);
  long r = i / cq;
  const int xo = (int)(r % Wo); r /= Wo;
  const int yo = (int)(r % Ho);
  const long b = r / Ho;
  const long base = ((b * H + 2 * yo) * (long)W + 2 * xo) * C + 4 * q;
  const long off[4] = {0, C, (long)W * C, (long)W * C + C};
  f32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(x + base + off[k]);
  const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
  f32x4 o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int best = 0;
    float bv = v[0][j];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (v[k][j] > bv) { bv = v[k][j]; best = k; }
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k][j] = (k == best) ? g[j] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f32x4* d = reinterpret_cast<f32x4*>(dx + base + off[k]);
    if (accumulate) {
      f32x4 t = *d;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] += o[k][j];
      *d = t;
    } else {
      *d = o[k];
    }
  }
}

// ------------------------------------------------------------------ BatchNorm + ReLU + max-pool 2x2 of a skip tensor
// The encoder's skip tensors (DoubleConv outputs that feed both MaxPool2d(2) and an Up block, unet/unet_model.py) in
// training mode, split formats: one pass over the conv output z writes y = relu(bn(z)) AND maxpool2(y), both in the split
// format only.  The split is monotone and the maximum is taken on the fp32 values, so the pooled planes are bit for bit
// those of bn_apply -> maxpool2_fwd -> f32_to_h2, without the fp32 copy of y (4 B per element written, 4 read), the pooled
// fp32 tensor and its conversion pass.  One thread = a 2x2 window (cropped at odd borders: those pixels still get their y)
// x 8 channels; a wave = 64 consecutive windows of the flattened (B * ceil(H/2), ceil(W/2)) grid x one channel group.
template <int NP>
__global__ __launch_bounds__(256) void bn_apply_pool_s3_kernel(const float* __restrict__ z, const float* __restrict__ mi,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               int H, int W, int C, long ncols,
                                                               unsigned short* __restrict__ y_s3, unsigned short* __restrict__ p_s3,
                                                               unsigned* __restrict__ overflow) {
  // one thread = the two rows of a window row x ONE column x 8 channels; lanes run along x (every store of a wave is one
  // contiguous run), the horizontal half of the maximum comes from the neighbouring lane.  Columns are counted over a width
  // rounded up to even, so that lanes 2i / 2i + 1 always hold the two columns of one window.
  const int nb = C >> 5;
  const int cb = (int)(blockIdx.x % (unsigned)nb);
  const long chunk = (long)(blockIdx.x / (unsigned)nb);
  const int g = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int c0 = cb * 32 + g * 8;
  float mean[8], invstd[8], gam[8], bet[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mean[j] = mi[c0 + j];
    invstd[j] = mi[C + c0 + j];
    gam[j] = gamma[c0 + j];
    bet[j] = beta[c0 + j];
  }
  const int qh = (H + 1) >> 1, Wp = (W + 1) & ~1, Ho = H >> 1, Wo = W >> 1;
  const long q = chunk * 64 + (threadIdx.x & 63);
  const unsigned qu = q < ncols ? (unsigned)q : 0u;
  const unsigned prow = qu / (unsigned)Wp;                 // b * qh + qy
  const int x = (int)(qu - prow * (unsigned)Wp);
  const unsigned b = prow / (unsigned)qh;
  const int qy = (int)(prow - b * (unsigned)qh);
  const bool live = q < ncols && x < W;
  float vr[2][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
  unsigned over = 0u;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int yy = 2 * qy + k;
    if (!live || yy >= H) continue;
    const long row = (long)b * H + yy;
    const long o = (row * W + x) * C + c0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 zz = *reinterpret_cast<const f32x4*>(z + o + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        vr[k][4 * h + j] = sfh_relu((zz[j] - mean[4 * h + j]) * invstd[4 * h + j] * gam[4 * h + j] + bet[4 * h + j]);
    }
    if constexpr (NP == 3) tr_split8(vr[k], y_s3, tr_s3_elem(row, x, c0, W, C), 4L * W * 8);
    else tr_split8_h2(vr[k], y_s3, tr_s3_elem(row, x, c0, W, C, 2), 4L * W * 8, over);
  }
  // the window's maximum with maxpool2_fwd's nesting, max(max(a, b), max(c, d)) over (row 0: a b, row 1: c d) - sfh_max_nan
  // picks its second argument on ties, so the nesting decides the sign of a zero; used on the even lane (a, c are its own)
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float r0 = __shfl_xor(vr[0][j], 1, 64), r1 = __shfl_xor(vr[1][j], 1, 64);
    m[j] = sfh_max_nan(sfh_max_nan(vr[0][j], r0), sfh_max_nan(vr[1][j], r1));
  }
  if (live && !(x & 1) && qy < Ho && (x >> 1) < Wo) {
    const long pr = (long)b * Ho + qy;
    if constexpr (NP == 3) tr_split8(m, p_s3, tr_s3_elem(pr, x >> 1, c0, Wo, C), 4L * Wo * 8);
    else tr_split8_h2(m, p_s3, tr_s3_elem(pr, x >> 1, c0, Wo, C, 2), 4L * Wo * 8, over);
  }
  if constexpr (NP == 2)
    if (overflow && sfh_h2_out_of_range(over)) atomicOr(overflow, 1u);
}

// Backward of the same pair, up to the BatchNorm sums: dx (B,H,W,C; the gradient y already has from its other consumer
// when accumulate != 0) += the max-pool routing of dp (first maximum of each window in scan order, as maxpool2_bwd; the
// window values y = relu(bn(z)) are recomputed from z with bn_apply's arithmetic instead of read), and, since dx is then
// the layer's total gradient, acc[0][c] += sum g, acc[1][c] += sum g * xhat with g = dx * (y > 0): what bn_bwd_reduce would
// find in a second pass over dx and z.  Mapping of the reductions above; one thread-iteration = a 2x2 window x 4 channels.
__global__ __launch_bounds__(256) void pool2_bwd_bn_reduce_kernel(const float* __restrict__ z, const float* __restrict__ mi,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  const float* __restrict__ dp, int H, int W, int C,
                                                                  long nquads, int accumulate, float* __restrict__ dx,
                                                                  double* __restrict__ acc) {
  __shared__ double sh[256];
  const int qh = (H + 1) >> 1, qw = (W + 1) >> 1, Ho = H >> 1, Wo = W >> 1;
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int cq = min(1024, C - c0) >> 2;
    const int lanes = 256 / cq;
    const int cqi = threadIdx.x % cq, pl = threadIdx.x / cq;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (pl < lanes) {
      const int cc = c0 + 4 * cqi;
      float mean[4], invstd[4], gam[4], bet[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        mean[j] = mi[cc + j];
        invstd[j] = mi[C + cc + j];
        gam[j] = gamma[cc + j];
        bet[j] = beta[cc + j];
      }
      for (long q = (long)blockIdx.x * lanes + pl; q < nquads; q += (long)gridDim.x * lanes) {
        const unsigned qrow = (unsigned)q / (unsigned)qw;
        const int qx = (int)((unsigned)q - qrow * (unsigned)qw);
        const unsigned b = qrow / (unsigned)qh;
        const int qy = (int)(qrow - b * (unsigned)qh);
        const bool window = qy < Ho && qx < Wo;
        const long base = (((long)b * H + 2 * qy) * W + 2 * qx) * C + cc;
        const long off[4] = {0, C, (long)W * C, (long)W * C + C};
        const bool valid[4] = {true, 2 * qx + 1 < W, 2 * qy + 1 < H, 2 * qx + 1 < W && 2 * qy + 1 < H};
        f32x4 yv[4], xh[4], t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f32x4 zz = {0.f, 0.f, 0.f, 0.f};
          t[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (valid[k]) {
            zz = *reinterpret_cast<const f32x4*>(z + base + off[k]);
            if (accumulate) t[k] = *reinterpret_cast<const f32x4*>(dx + base + off[k]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            yv[k][j] = (zz[j] - mean[j]) * invstd[j] * gam[j] + bet[j];
            xh[k][j] = (zz[j] - mean[j]) * invstd[j];
          }
        }
        if (window) {
          const f32x4 gq = *reinterpret_cast<const f32x4*>(dp + (((long)b * Ho + qy) * Wo + qx) * C + cc);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            int best = 0;
            float bv = sfh_relu(yv[0][j]);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
              const float v = sfh_relu(yv[k][j]);
              if (v > bv) { bv = v; best = k; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k][j] += (k == best) ? gq[j] : 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!valid[k]) continue;
          if (window || !accumulate) *reinterpret_cast<f32x4*>(dx + base + off[k]) = t[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float gj = yv[k][j] > 0.f ? t[k][j] : 0.f;
            s0[j] += (double)gj;
            s1[j] += (double)gj * (double)xh[k][j];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __syncthreads();
      sh[threadIdx.x] = (pl < lanes) ? (j < 4 ? s0[j] : s1[j - 4]) : 0.0;
      __syncthreads();
      if (threadIdx.x < cq) {
        double tt = 0.0;
        for (int l = 0; l < lanes; ++l) tt += sh[l * cq + threadIdx.x];
        unsafeAtomicAdd(&acc[(long)(j >> 2) * C + c0 + 4 * threadIdx.x + (j & 3)], tt);
      }
    }
  }
}

// ------------------------------------------------------------------ data movers
// dst (B,h,w,C) (+)= src[b, y+oy, x+ox, c_off : c_off+C] with src (B,Hs,Ws,cs); out-of-range -> 0
__global__ __launch_bounds__(256) void slice_add_kernel(const float* __restrict__ src, int Hs, int Ws, int cs, int c_off,
                                                        int oy, int ox, int h, int w, int C, long total4,
                                                        int accumulate, float* __restrict__ dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int cq = C >> 2;
  const int q = (int)(i % cq);
  long r = i / cq;
  const int x = (int)(r % w); r /= w;
  const int y = (int)(r % h);
  const long b = r / h;
  const int sy = y + oy, sx = x + ox;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (sy >= 0 && sy < Hs && sx >= 0 && sx < Ws)
    v = *reinterpret_cast<const f32x4*>(src + ((b * Hs + sy) * (long)Ws + sx) * cs + c_off + 4 * q);
  f32x4* d = reinterpret_cast<f32x4*>(dst) + i;
  if (accumulate) {
    f32x4 t = *d;
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] += v[j];
    *d = t;
  } else {
    *d = v;
  }
}

// dst (B,H,W,C): dst[2j,2i] = src[j,i] (src (B,ho,wo,C)), zero elsewhere
__global__ __launch_bounds__(256) void zero_stuff2_kernel(const float* __restrict__ src, int ho, int wo, int H, int W,
                                                          int C, long total4, float* __restrict__ dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int cq = C >> 2;
  const int q = (int)(i % cq);
  long r = i / cq;
  const int x = (int)(r % W); r /= W;
  const int y = (int)(r % H);
  const long b = r / H;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!(y & 1) && !(x & 1) && (y >> 1) < ho && (x >> 1) < wo)
    v = *reinterpret_cast<const f32x4*>(src + ((b * ho + (y >> 1)) * (long)wo + (x >> 1)) * C + 4 * q);
  reinterpret_cast<f32x4*>(dst)[i] = v;
}

// ------------------------------------------------------------------ OutConv backward
// logits = w x + b (1x1, cin -> NC): dx[p][ci] = sum_k dl[k][p] w[k][ci];
// acc_w[k][ci] += sum_p dl[k][p] x[p][ci]; acc_b[k] += sum_p dl[k][p].   dl is NCHW, x / dx NHWC.
// BN (sfh_outconv_bwd_bn): x is not read - it is the BatchNorm + ReLU output of the layer in front, recomputed from that
// layer's conv output z with bn_apply's arithmetic (same bits) - and, dx being that layer's whole gradient, the pass also leaves
// its backward sums [sum g | sum g * xhat], g = dx * (x > 0): sfh_bn_bwd_reduce's second pass over dx and z is not needed.
template <int NC, bool BN>
__global__ __launch_bounds__(256) void outconv_bwd_kernel(const float* __restrict__ x, int cin,
                                                          const float* __restrict__ w, const float* __restrict__ dl,
                                                          long npix, int HW, float* __restrict__ dx,
                                                          double* __restrict__ acc_w, double* __restrict__ acc_b,
                                                          int pix_per_block, const float* __restrict__ bn_mi,
                                                          const float* __restrict__ bn_gamma, const float* __restrict__ bn_beta,
                                                          double* __restrict__ acc_bn) {
  __shared__ double sh[256];
  const int cq = cin >> 2;  // <= 64
  const int lanes = 256 / cq;
  const int q = threadIdx.x % cq, pl = threadIdx.x / cq;
  f32x4 wk[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) wk[k] = *reinterpret_cast<const f32x4*>(w + k * cin + 4 * q);
  float bmean[4], binv[4], bgam[4], bbet[4];
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  if constexpr (BN) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bmean[j] = bn_mi[4 * q + j];
      binv[j] = bn_mi[cin + 4 * q + j];
      bgam[j] = bn_gamma[4 * q + j];
      bbet[j] = bn_beta[4 * q + j];
    }
  }
  // dW / db partial sums: fp32 over at most 64 pixels of a thread, then promoted into fp64 - a workgroup covers npix / 1024
  // pixels, so a thread's chain grows with the image (about 900 terms at 1280x720 x 16) and must not stay in fp32
  float sw[NC][4], sb[NC];
  double dsw[NC][4], dsb[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    sb[k] = 0.f;
    dsb[k] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sw[k][j] = 0.f;
      dsw[k][j] = 0.0;
    }
  }
  const long p0 = (long)blockIdx.x * pix_per_block, p1 = min(p0 + (long)pix_per_block, npix);
  int since_flush = 0;
  if (pl < lanes)
    for (long p = p0 + pl; p < p1; p += lanes) {
      if (++since_flush > 64) {
        since_flush = 1;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          dsb[k] += (double)sb[k];
          sb[k] = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            dsw[k][j] += (double)sw[k][j];
            sw[k][j] = 0.f;
          }
        }
      }
      const long b = p / HW, i = p - b * HW;
      f32x4 xv = *reinterpret_cast<const f32x4*>(x + p * cin + 4 * q);
      f32x4 xh = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[j] = (xv[j] - bmean[j]) * binv[j];
          xv[j] = sfh_relu((xv[j] - bmean[j]) * binv[j] * bgam[j] + bbet[j]);
        }
      }
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const float g = dl[(b * NC + k) * HW + i];
        sb[k] += g;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] += g * wk[k][j];
          sw[k][j] += g * xv[j];
        }
      }
      if (dx) *reinterpret_cast<f32x4*>(dx + p * cin + 4 * q) = o;
      if constexpr (BN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float gj = xv[j] > 0.f ? o[j] : 0.f;     // (x > 0 exactly where bn_apply's pre-ReLU value is > 0)
          s0[j] += (double)gj;
          s1[j] += (double)gj * (double)xh[j];
        }
      }
    }
  if constexpr (BN) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __syncthreads();
      sh[threadIdx.x] = (pl < lanes) ? (j < 4 ? s0[j] : s1[j - 4]) : 0.0;
      __syncthreads();
      if (threadIdx.x < cq) {
        double t = 0.0;
        for (int l = 0; l < lanes; ++l) t += sh[l * cq + threadIdx.x];
        unsafeAtomicAdd(&acc_bn[(long)(j >> 2) * cin + 4 * threadIdx.x + (j & 3)], t);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      __syncthreads();
      sh[threadIdx.x] = (pl < lanes) ? (j < 4 ? dsw[k][j] + (double)sw[k][j] : dsb[k] + (double)sb[k]) : 0.0;
      __syncthreads();
      if (j < 4) {
        if (threadIdx.x < cq) {
          double t = 0.0;
          for (int l = 0; l < lanes; ++l) t += sh[l * cq + threadIdx.x];
          unsafeAtomicAdd(&acc_w[k * cin + 4 * threadIdx.x + j], t);
        }
      } else if (threadIdx.x == 0) {
        double t = 0.0;
        for (int l = 0; l < lanes; ++l) t += sh[l * cq];  // q == 0 lanes
        unsafeAtomicAdd(&acc_b[k], t);
      }
    }
  }
}

// ------------------------------------------------------------------ ResNetSTN backward pieces
// MaxPool2d(3, stride 2, padding 1) backward: every input pixel looks at the (up to 4) windows that
// contain it and takes their gradient when it is the window's first maximum in scan order.
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               int H, int W, int C, int Ho, int Wo, long total4,
                                                               float* __restrict__ dx) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int cq = C >> 2;
  const int q = (int)(idx % cq);
  long r = idx / cq;
  const int xi = (int)(r % W); r /= W;
  const int yi = (int)(r % H);
  const long b = r / H;
  const float* xb = x + b * (long)H * W * C + 4 * q;
  f32x4 g = {0.f, 0.f, 0.f, 0.f};
  for (int yo = yi / 2; yo <= (yi + 1) / 2; ++yo) {   // windows with 2*yo-1 <= yi <= 2*yo+1
    if (yo >= Ho) continue;
    for (int xo = xi / 2; xo <= (xi + 1) / 2; ++xo) {
      if (xo >= Wo) continue;
      f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int by[4] = {-1, -1, -1, -1}, bx[4] = {-1, -1, -1, -1};
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = 2 * yo - 1 + ky;
        if (yy < 0 || yy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
          const int xx = 2 * xo - 1 + kx;
          if (xx < 0 || xx >= W) continue;
          const f32x4 v = *reinterpret_cast<const f32x4*>(xb + ((long)yy * W + xx) * C);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (v[j] > best[j] || by[j] < 0) { best[j] = v[j]; by[j] = yy; bx[j] = xx; }
        }
      }
      const f32x4 d = *reinterpret_cast<const f32x4*>(dy + ((b * Ho + yo) * (long)Wo + xo) * C + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (by[j] == yi && bx[j] == xi) g[j] += d[j];
    }
  }
  reinterpret_cast<f32x4*>(dx)[idx] = g;
}

// AdaptiveAvgPool2d(1) + Linear backward; one block per frame.
// dx[b,p,c] = (sum_j dout[b][j] w[j][c]) / HW;  acc_w[j][c] += dout[b][j] * mean_p x[b,p,c];  acc_b[j] += dout[b][j]
__global__ __launch_bounds__(256) void avgpool_linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ dout, int HW, int C, int nout,
                                                                 float* __restrict__ dx, double* __restrict__ acc_w,
                                                                 double* __restrict__ acc_b) {
  const int b = blockIdx.x;
  const float* xb = x + (long)b * HW * C;
  float* dxb = dx + (long)b * HW * C;
  const float inv = 1.0f / (float)HW;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int i = 0; i < HW; ++i) s += xb[(long)i * C + c];
    const float mean = s * inv;
    float df = 0.f;
    for (int j = 0; j < nout; ++j) {
      const float d = dout[b * nout + j];
      df += d * w[j * C + c];
      unsafeAtomicAdd(&acc_w[(long)j * C + c], (double)d * (double)mean);
    }
    df *= inv;
    for (int i = 0; i < HW; ++i) dxb[(long)i * C + c] = df;
  }
  if (threadIdx.x < nout) unsafeAtomicAdd(&acc_b[threadIdx.x], (double)dout[b * nout + threadIdx.x]);
}

// Backward-data of the 7x7 stride-2 pad-3 stem (models/resnet.py:172) for the first `nc` input channels
// (the logits inside cat((logits, x), 1)): dlogits[b][c][y][x] += sum_{co,ky,kx} dz[b][Y][X][co] *
// w[co][c][ky][kx] with 2Y + ky - 3 = y, 2X + kx - 3 = x.  One thread per input pixel; the weights of
// the nc channels sit in LDS as [ky][kx][co][4].
template <int NC4>
__global__ __launch_bounds__(256) void stem_bwd_data_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                            int cin, int c_off, int nc, int H, int W, int Ho, int Wo,
                                                            float* __restrict__ dlogits, int batch) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [49][64][4*NC4]
  for (int i = threadIdx.x; i < 49 * 64 * 4 * NC4; i += 256) {
    const int c = i % (4 * NC4), co = (i / (4 * NC4)) % 64, t = i / (4 * NC4 * 64);
    wl[i] = c < nc ? w[((long)co * cin + c_off + c) * 49 + t] : 0.f;
  }
  __syncthreads();
  // a thread owns the pixels x and x + 64 of a row: same column parity, i.e. the same taps (ky, kx) and the same weights -
  // every 16-byte LDS read of a weight quad feeds two pixels (the kernel was bound by those reads: one per four FMAs).
  // (Round 4: FOUR pixels per thread measured slower, 1.28 -> 1.36 ms; waves of ONE column parity - wave-uniform tap loops,
  // broadcast weight reads - slower still, 1.23 -> 1.88 ms: with lane -> every second column each dz load instruction touches
  // 64 cache lines instead of 32; the kernel is bound by that gather, not by the LDS weight reads.)
  // Workgroups reach the 8 XCDs round-robin in launch order.  Round 6 (profiles/r06_tcc_train_per_launch.txt): with (x tile, 4-row
  // band, frame) = blockIdx the neighbouring bands - which gather the same dz rows - sat on different XCDs: 1.78 GB of L2 misses for
  // 0.3 GB of tensors.  XCD x now owns the contiguous range [x * chunk, (x + 1) * chunk) of the (frame, band, x tile) order.
  // (1-D launch of 8 * chunk workgroups; gx x gy x batch tiles)
  const int gx = (W + 127) >> 7, gy = (H + 3) >> 2;
  const long total = (long)gx * gy * batch;
  const long chunk = (total + 7) >> 3;
  const long own = (long)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  if (own >= total) return;                                // (uniform per workgroup; the weights' barrier is behind us)
  const int bx = (int)(own % gx), by = (int)((own / gx) % gy);
  const int x = bx * 128 + (threadIdx.x & 63);
  const int y = by * 4 + (threadIdx.x >> 6);
  const int b = (int)(own / ((long)gx * gy));
  if (x >= W || y >= H) return;
  const bool two = x + 64 < W;
  float acc[2][4 * NC4];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < 4 * NC4; ++c) acc[p][c] = 0.f;
  for (int ky = (y + 3) & 1; ky < 7; ky += 2) {
    const int Y = (y + 3 - ky) >> 1;
    if (Y < 0 || Y >= Ho) continue;
    for (int kx = (x + 3) & 1; kx < 7; kx += 2) {
      const int X0 = (x + 3 - kx) >> 1, X1 = X0 + 32;
      const bool ok0 = X0 >= 0 && X0 < Wo, ok1 = two && X1 >= 0 && X1 < Wo;
      if (!ok0 && !ok1) continue;
      const float* dp0 = dz + (((long)b * Ho + Y) * Wo + (ok0 ? X0 : 0)) * 64;
      const float* dp1 = dz + (((long)b * Ho + Y) * Wo + (ok1 ? X1 : 0)) * 64;
      const float* wp = wl + (ky * 7 + kx) * 64 * 4 * NC4;
#pragma unroll 4
      for (int co4 = 0; co4 < 16; ++co4) {
        f32x4 d0 = *reinterpret_cast<const f32x4*>(dp0 + 4 * co4), d1 = *reinterpret_cast<const f32x4*>(dp1 + 4 * co4);
        if (!ok0) d0 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (!ok1) d1 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int q = 0; q < NC4; ++q) {
            const f32x4 ww = *reinterpret_cast<const f32x4*>(wp + ((4 * co4 + j) * NC4 + q) * 4);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              acc[0][4 * q + c] += d0[j] * ww[c];
              acc[1][4 * q + c] += d1[j] * ww[c];
            }
          }
        }
      }
    }
  }
  for (int c = 0; c < nc; ++c) {
    dlogits[(((long)b * nc + c) * H + y) * W + x] += acc[0][c];
    if (two) dlogits[(((long)b * nc + c) * H + y) * W + x + 64] += acc[1][c];
  }
}

// ------------------------------------------------------------------ losses of the training step
// train.py:181-224 + models/losses.py in one pass over the pixels: segmentation CE (per-sample
// weighted), SmoothL1 between the bilinear warp and gt/nc (per-sample weighted), consistency CE against
// trunc(warp * nc).  Writes d loss / d logits (NCHW) and d loss / d warp, accumulates the three loss
// values in fp64.  loss[0] = seg, [1] = rec, [2] = consistency (already scaled by their lambdas).
struct LossArgs {
  const float* logits; const long* gt; const float* weight; const float* warp;
  int nc, HW; long npix; int batch;
  float l_seg, l_rec, l_cons;   // lambda (0 = loss disabled)
  int rec_mse;                  // 1: nn.MSELoss, 0: nn.SmoothL1Loss (train.py:113-118)
  int seg_focal, cons_focal;    // 1: kornia FocalLoss(alpha=1, gamma=2) instead of CE (train.py:101,126)
  float* dlogits; float* dwarp; double* loss;
};

// one classification loss term at a pixel: CE (lse - l[tgt]) or Kornia's focal loss with alpha 1, gamma 2
// (train.py:101,126): sum_k (onehot_k + 1e-6) * -(1 - q_k)^2 log q_k, q = softmax + 1e-8; adds coef * dL/dl to dl
template <int NC>
__device__ __forceinline__ float cls_loss(const float (&l)[NC], const float (&pr)[NC], float lse, int tgt, bool focal,
                                          float coef, float (&dl)[NC]) {
  if (!focal) {
    float lt = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (k == tgt) lt = l[k];
      dl[k] += coef * (pr[k] - (k == tgt ? 1.f : 0.f));
    }
    return lse - lt;
  }
  float dLdp[NC], dot = 0.f, loss = 0.f;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const float q = pr[k] + 1e-8f, t = (k == tgt ? 1.f : 0.f) + 1e-6f;
    const float om = 1.f - q, lg = logf(q);
    loss += t * (-om * om * lg);
    dLdp[k] = t * (2.f * om * lg - om * om / q);
    dot += dLdp[k] * pr[k];
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) dl[k] += coef * pr[k] * (dLdp[k] - dot);
  return loss;
}

template <int NC>
__global__ __launch_bounds__(256) void train_losses_kernel(const LossArgs a) {
  __shared__ double sh[256];
  double s_seg = 0.0, s_rec = 0.0, s_cons = 0.0;
  const float inv = 1.0f / ((float)a.HW * (float)a.batch);
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < a.npix; p += (long)gridDim.x * 256) {
    const long b = p / a.HW, i = p - b * a.HW;
    float l[NC], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NC; ++k) { l[k] = a.logits[(b * NC + k) * a.HW + i]; mx = fmaxf(mx, l[k]); }
    float se = 0.f, pr[NC], dl[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { pr[k] = expf(l[k] - mx); se += pr[k]; dl[k] = 0.f; }
    const float lse = logf(se) + mx, rs = 1.0f / se;
#pragma unroll
    for (int k = 0; k < NC; ++k) pr[k] *= rs;
    const int gt = (int)a.gt[p];
    const float wb = a.weight[b];
    const float wv = a.warp ? a.warp[p] : 0.f;
    const int tc = min(max((int)(wv * (float)NC), 0), NC - 1);   // (warp_mask * nc).to(long)
    if (a.l_seg != 0.f) s_seg += (double)(wb * cls_loss<NC>(l, pr, lse, gt, a.seg_focal != 0, a.l_seg * wb * inv, dl));
    if (a.l_cons != 0.f) s_cons += (double)cls_loss<NC>(l, pr, lse, tc, a.cons_focal != 0, a.l_cons * inv, dl);
#pragma unroll
    for (int k = 0; k < NC; ++k) a.dlogits[(b * NC + k) * a.HW + i] = dl[k];
    if (a.warp) {
      const float d = wv - (float)gt / (float)NC;
      const float ad = fabsf(d);
      if (a.rec_mse) {
        s_rec += (double)(wb * d * d);
        if (a.dwarp) a.dwarp[p] = a.l_rec * wb * inv * 2.f * d;
      } else {
        s_rec += (double)(wb * (ad < 1.f ? 0.5f * d * d : ad - 0.5f));
        if (a.dwarp) a.dwarp[p] = a.l_rec * wb * inv * (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f));
      }
    }
  }
  double v[3] = {s_seg * a.l_seg, s_rec * a.l_rec, s_cons * a.l_cons};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    __syncthreads();
    sh[threadIdx.x] = v[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) unsafeAtomicAdd(&a.loss[k], sh[0] * (double)inv);
  }
}

// models/losses.py:6-19 (reduction 'mean') and its gradient wrt the projected points; one thread per frame.
__global__ void reproj_loss_kernel(const float* __restrict__ poi, const float* __restrict__ gt,
                                   const float* __restrict__ nonzeros, const float* __restrict__ num_nonzero,
                                   int batch, int npts, float lambda, float* __restrict__ dpoi, double* __restrict__ loss) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  double s = 0.0;
  const float sc = lambda / (num_nonzero[b] * (float)batch);
  for (int n = 0; n < npts; ++n) {
    const long o = ((long)b * npts + n) * 2;
    const float dx = gt[o] - poi[o], dy = gt[o + 1] - poi[o + 1];
    const float dist = sqrtf(dx * dx + dy * dy);
    const float nz = nonzeros[(long)b * npts + n];
    s += (double)(dist * nz);
    const float g = dist > 0.f ? nz * sc / dist : 0.f;   // torch propagates NaN at dist == 0; 0 here
    dpoi[o] = -g * dx;
    dpoi[o + 1] = -g * dy;
  }
  unsafeAtomicAdd(loss, s * (double)sc);
}

// ------------------------------------------------------------------ clip_grad_value_ + RMSprop
// train.py:88,234-237: g = clamp(grad, -clip, clip); g += wd * p; sq = alpha*sq + (1-alpha)*g*g;
// buf = mu*buf + g / (sqrt(sq) + eps); p -= lr * buf   (torch.optim.RMSprop, centered=False).
// Multi-tensor: `table` holds per tensor {param, grad, square_avg, momentum_buf} pointers and `chunks`
// {tensor index, element offset, count}; one block per chunk.
struct OptTensor { float* p; const float* g; float* sq; float* buf; };
struct OptChunk { int tensor; int count; long offset; };

__global__ __launch_bounds__(256) void rmsprop_kernel(const OptTensor* __restrict__ table, const OptChunk* __restrict__ chunks,
                                                      float lr, float alpha, float eps, float wd, float mu, float clip,
                                                      float gscale) {
  const OptChunk c = chunks[blockIdx.x];
  const OptTensor t = table[c.tensor];
  for (int i = threadIdx.x; i < c.count; i += 256) {
    const long o = c.offset + i;
    float g = t.g[o] * gscale;   // 1/world: data-parallel mean of the all-reduced sum
    if (clip > 0.f) g = fminf(fmaxf(g, -clip), clip);
    const float p = t.p[o];
    g = g + wd * p;
    const float sq = alpha * t.sq[o] + (1.0f - alpha) * g * g;
    t.sq[o] = sq;
    const float avg = sqrtf(sq) + eps;
    float step = g / avg;
    if (mu > 0.f) {
      step = mu * t.buf[o] + step;
      t.buf[o] = step;
    }
    t.p[o] = p - lr * step;
  }
}

// train.py:90: torch.optim.SGD(lr, momentum, weight_decay) (dampening 0, no Nesterov) behind clip_grad_value_:
// g = clamp(grad); g += wd * p; buf = mu * buf + g (a zero-filled buffer gives torch's first step, buf = g); p -= lr * buf.
__global__ __launch_bounds__(256) void sgd_kernel(const OptTensor* __restrict__ table, const OptChunk* __restrict__ chunks,
                                                  float lr, float wd, float mu, float clip, float gscale) {
  const OptChunk c = chunks[blockIdx.x];
  const OptTensor t = table[c.tensor];
  for (int i = threadIdx.x; i < c.count; i += 256) {
    const long o = c.offset + i;
    float g = t.g[o] * gscale;
    if (clip > 0.f) g = fminf(fmaxf(g, -clip), clip);
    const float p = t.p[o];
    g = g + wd * p;
    if (mu > 0.f) {
      g = mu * t.buf[o] + g;
      t.buf[o] = g;
    }
    t.p[o] = p - lr * g;
  }
}

// train.py:92: torch.optim.Adam(lr, betas, weight_decay) (L2 form, no amsgrad) behind clip_grad_value_, in torch's own
// operation order: m = lerp(m, g, 1 - b1); v = b2 * v + (1 - b2) * g * g; denom = sqrt(v) / sqrt(1 - b2^t) + eps;
// p -= (lr / (1 - b1^t)) * m / denom.  `buf` holds exp_avg, `sq` exp_avg_sq; the two bias corrections come from the host.
__global__ __launch_bounds__(256) void adam_kernel(const OptTensor* __restrict__ table, const OptChunk* __restrict__ chunks,
                                                   float step_size, float b1, float b2, float eps, float wd, float clip,
                                                   float gscale, float bias2_sqrt) {
  const OptChunk c = chunks[blockIdx.x];
  const OptTensor t = table[c.tensor];
  for (int i = threadIdx.x; i < c.count; i += 256) {
    const long o = c.offset + i;
    float g = t.g[o] * gscale;
    if (clip > 0.f) g = fminf(fmaxf(g, -clip), clip);
    const float p = t.p[o];
    g = g + wd * p;
    float m = t.buf[o];
    m = m + (1.0f - b1) * (g - m);
    t.buf[o] = m;
    const float v = b2 * t.sq[o] + (1.0f - b2) * g * g;
    t.sq[o] = v;
    const float denom = sqrtf(v) / bias2_sqrt + eps;
    t.p[o] = p - step_size * (m / denom);
  }
}

// train.py:136-144,203-208: per_sample_weighted_criterion(MSELoss / SmoothL1Loss(reduction='none'), uv, gt_uv, weights) *
// lambda on (B,2,H,W) tensors.  models/losses.py:33-41 takes torch.mean(loss, dim=(1, 2)) - on a 4-D map that is the mean
// over CHANNEL and ROW, leaving (B, W) - and multiplies by the (B,) weights, which broadcasts along the LAST axis: the
// weight of column w is weights[w] (legal only for B == W) or weights[0] (B == 1).  Restated as written: `wcol` = 1 picks
// weights[w], 0 weights[0].  loss += lambda * mean_{b,w}(weight * mean_{c,h} l); duv = its gradient.
__global__ __launch_bounds__(256) void uv_loss_kernel(const float* __restrict__ uv, const float* __restrict__ gt,
                                                      const float* __restrict__ weight, int wcol, int C, int H, int W,
                                                      long total, float lambda, int mse, float* __restrict__ duv,
                                                      double* __restrict__ loss, float inv) {
  __shared__ double sh[256];
  double s = 0.0;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const int w = (int)(p % W);
    const float wb = weight[wcol ? w : 0];
    const float d = uv[p] - gt[p];
    const float ad = fabsf(d);
    if (mse) {
      s += (double)(wb * d * d);
      duv[p] = lambda * wb * inv * 2.f * d;
    } else {
      s += (double)(wb * (ad < 1.f ? 0.5f * d * d : ad - 0.5f));
      duv[p] = lambda * wb * inv * (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f));
    }
  }
  __syncthreads();
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, sh[0] * (double)lambda * (double)inv);
}

// ------------------------------------------------------------------ multi-tensor copy
// dst_t (contiguous) = src_t (up to 4-D, any strides) [* scale] for many tensors in ONE launch: the gradient assembly of a
// training step (182 tensors, the weight gradients as permuted views of the backward-filter buffers, times 1 / the
// power-of-two gradient scale) and the BatchNorm statistics snapshot (162 buffers) were one hipMemcpy each.
// 4-byte elements; SCALE = false moves the words untouched (integer buffers).
struct CopyTensor { void* dst; const void* src; int d1, d2, d3, pad; long s0, s1, s2, s3; };

template <bool SCALE>
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyTensor* __restrict__ table, const OptChunk* __restrict__ chunks,
                                                         float scale, const float* __restrict__ scale_dev = nullptr) {
  if (SCALE && scale_dev) scale = *scale_dev;     // (the factor lives on the device: sfh_multi_copy_dscale)
  const OptChunk c = chunks[blockIdx.x];
  const CopyTensor t = table[c.tensor];
  const bool contiguous = t.s3 == 1 && t.s2 == t.d3 && t.s1 == (long)t.d2 * t.d3 && t.s0 == (long)t.d1 * t.d2 * t.d3;
  for (int i = threadIdx.x; i < c.count; i += 256) {
    const long o = c.offset + i;
    long so = o;
    if (!contiguous) {
      const long i3 = o % t.d3;
      long r = o / t.d3;
      const long i2 = r % t.d2;
      r /= t.d2;
      const long i1 = r % t.d1, i0 = r / t.d1;
      so = i0 * t.s0 + i1 * t.s1 + i2 * t.s2 + i3 * t.s3;
    }
    if (SCALE) reinterpret_cast<float*>(t.dst)[o] = reinterpret_cast<const float*>(t.src)[so] * scale;
    else reinterpret_cast<unsigned*>(t.dst)[o] = reinterpret_cast<const unsigned*>(t.src)[so];
  }
}

// ------------------------------------------------------------------ weight gradient (fp32 MFMA)
// raw[m][tap][n] += sum over pixels p of dz[p][m] * xin[p + tap][n]
//   GEMM view: M = output channels, N = input channels of one source, K = B*H*W pixels.
// Workgroup = 64 m x 16*NSUB n x all taps, over a strided subset of 2x32-pixel tiles (blockIdx.z =
// split); each wave owns 16 m.  v_mfma_f32_16x16x4_f32: A[i=m][k=pixel], B[k=pixel][j=n] - in NHWC the
// 16 channels of a pixel are contiguous, so both fragments are plain ds_read_b32 with k = lane/16.
// LDS rows are 64 floats per pixel; bit 4 of the channel index is XORed with the pixel parity so the
// two pixels a 32-lane group reads land in different bank halves.
struct WgradArgs {
  const float* dz; int dz_cs; int M;
  const float* x; int x_cs, xh, xw, N, pad_top, pad_left;
  int batch, H, W;
  float* raw; int raw_n, n_off;
  int ntx, nty, ntiles, nsplit, mt, mn;
  // wgrad_c4_kernel only (sfh_conv_wgrad_c4_bn): dz points at dy, the gradient of the BatchNorm + ReLU OUTPUT, and the
  // BatchNorm backward is applied while a tile is loaded (bn_z != nullptr): z, mean | invstd, gamma, beta, the finished
  // backward sums [sum g | sum g * xhat] and 1 / (pixels per channel)
  const float* bn_z; const float* bn_mi; const float* bn_gamma; const float* bn_beta; const double* bn_acc; float bn_inv_n;
};

template <int KS, int NSUB, int TH, int TW>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
  static_assert(TH * TW == 64 && TW % 4 == 0, "64-pixel tiles");
  constexpr int PADB = KS == 3 ? 1 : (KS == 4 ? 2 : 0);
  constexpr int HR = TH + KS - 1, HW = TW + KS - 1, HPX = HR * HW;
  constexpr int KK = KS * KS;
  constexpr int NHV = (HPX * 4 * NSUB + 255) / 256;  // halo float4 loads per thread
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xL = lds;               // [HPX][64]
  float* zL = lds + HPX * 64;    // [64][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  // XCD-aware order (workgroups go round-robin to the 8 XCDs): the mn = (m tiles) x (n tiles) workgroups
  // of one pixel split run back to back on ONE XCD, so the split's dz / x tiles are fetched into that
  // L2 once instead of once per (m, n) block
  const int xcd = blockIdx.x & 7, kk_ = blockIdx.x >> 3;
  const int split = (kk_ / a.mn) * 8 + xcd, mni = kk_ % a.mn;
  if (split >= a.nsplit) return;
  const int m0 = (mni % a.mt) * 64, n0 = (mni / a.mt) * (16 * NSUB);
  f32x4 acc[KK][NSUB];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int s = 0; s < NSUB; ++s) acc[t][s] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // global loads of one tile into registers (zero outside the frame / the channel range)
  f32x4 hv[NHV], zv[4];
  auto load_tile = [&](int tile) {
    const int tx = tile % a.ntx;
    const int ty = (tile / a.ntx) % a.nty;
    const int b = tile / (a.ntx * a.nty);
    const int y0 = ty * TH, x0 = tx * TW;
#pragma unroll
    for (int k = 0; k < NHV; ++k) {
      const int i = tid + k * 256;
      const int q = i % (4 * NSUB), hp = i / (4 * NSUB);
      const int hr = hp / HW, hc = hp - hr * HW;
      const int sy = y0 + hr - PADB - a.pad_top, sx = x0 + hc - PADB - a.pad_left;
      const int n = n0 + 4 * q;
      hv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (hp < HPX && sy >= 0 && sy < a.xh && sx >= 0 && sx < a.xw && n < a.N)
        hv[k] = *reinterpret_cast<const f32x4*>(a.x + (((long)b * a.xh + sy) * a.xw + sx) * a.x_cs + n);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256;
      const int q = i & 15, p = i >> 4;
      const int r = p / TW, c = p - r * TW;
      const int y = y0 + r, x = x0 + c, m = m0 + 4 * q;
      zv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (y < a.H && x < a.W && m < a.M)
        zv[k] = *reinterpret_cast<const f32x4*>(a.dz + (((long)b * a.H + y) * a.W + x) * a.dz_cs + m);
    }
  };
  // Software pipeline over the workgroup's tiles (round 4): the loads of tile t + 1 are requested right behind the
  // barrier that publishes tile t in LDS and stay in flight during its contraction; before, every tile paid its own
  // load latency between two barriers (the 3-channel first layer: 1.03 ms for 0.94 GB = 0.9 TB/s).
  // Only for the small instances (the 3-channel first layer: 36 accumulator registers): with 128+ accumulator registers
  // the prefetched tile costs occupancy and the 7x7 stem's instance measured 0.63 -> 0.71 ms.
  constexpr bool PIPE = KK * NSUB <= 9;
  if (PIPE && split < a.ntiles) load_tile(split);
  for (int tile = split; tile < a.ntiles; tile += a.nsplit) {
    if (!PIPE) load_tile(tile);   // every global load of the tile issued first (one latency, not one per iteration)
    __syncthreads();  // previous tile's fragments consumed
#pragma unroll
    for (int k = 0; k < NHV; ++k) {
      const int i = tid + k * 256;
      const int q = i % (4 * NSUB), hp = i / (4 * NSUB);
      if (hp < HPX) *reinterpret_cast<f32x4*>(xL + hp * 64 + 4 * (q ^ ((hp & 1) << 2))) = hv[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256;
      const int q = i & 15, p = i >> 4;
      *reinterpret_cast<f32x4*>(zL + p * 64 + 4 * (q ^ ((p & 1) << 2))) = zv[k];
    }
    __syncthreads();
    if (PIPE && tile + a.nsplit < a.ntiles) load_tile(tile + a.nsplit);
    // ---- contraction over the 64 pixels of the tile, 4 per MFMA
#pragma unroll 1
    for (int r = 0; r < TH; ++r) {
#pragma unroll 2
      for (int c4 = 0; c4 < TW / 4; ++c4) {
        const int pz = r * TW + c4 * 4 + kq;
        const float av = zL[pz * 64 + ((wave * 16 + l16) ^ ((pz & 1) << 4))];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky)
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const int ph = (r + ky) * HW + c4 * 4 + kq + kx;
            const float* row = xL + ph * 64;
            const int sw = (ph & 1) << 4;
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
              const float bv = row[(s * 16 + l16) ^ sw];
              acc[ky * KS + kx][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[ky * KS + kx][s], 0, 0, 0);
            }
          }
      }
    }
  }
  // ---- D[i = 4*(lane/16) + r][j = lane%16]
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      const int n = n0 + s * 16 + l16;
      if (n >= a.N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wave * 16 + 4 * kq + r;
        if (m < a.M) unsafeAtomicAdd(a.raw + ((long)m * KK + t) * a.raw_n + a.n_off + n, acc[t][s][r]);
      }
    }
}

template <int KS, int NSUB, int TH, int TW>
int launch_wgrad(WgradArgs a, hipStream_t stream) {
  constexpr int HPX = (TH + KS - 1) * (TW + KS - 1);
  const size_t lds = (size_t)(HPX + 64) * 64 * sizeof(float);
  a.ntx = sfh_cdiv(a.W, TW);
  a.nty = sfh_cdiv(a.H, TH);
  a.ntiles = a.batch * a.ntx * a.nty;
  const int mn = sfh_cdiv(a.M, 64) * sfh_cdiv(a.N, 16 * NSUB);
  int ns = sfh_cdiv(1024, mn);
  if (ns > a.ntiles) ns = a.ntiles;
  if (ns < 1) ns = 1;
  if (ns > 65535) ns = 65535;
  if (ns < 8 && a.ntiles >= 8) ns = 8;  // one split per XCD at least
  a.nsplit = ns;
  a.mt = sfh_cdiv(a.M, 64);
  a.mn = mn;
  const dim3 grid((unsigned)(sfh_cdiv(ns, 8) * 8 * mn));
  hipLaunchKernelGGL((wgrad_kernel<KS, NSUB, TH, TW>), grid, dim3(256), lds, stream, a);
  return sfh_check_launch("wgrad_kernel");
}

// Backward-filter of a 3x3 conv over at most FOUR input channels (the UNet's first layer: N = 3 stored as 4; round 4).
// The generic kernel above spends one MFMA per tap on a 16-wide n block of which 4 columns are channels (75 % padding:
// 0.75 ms, bound by the fp32 matrix pipe).  Here the n axis is (tap, channel): n = 4 * tap + c, 36 real columns in three
// blocks of 16 - three MFMAs per four pixels instead of nine - and the halo is 16 bytes per pixel (2 KB instead of 35 KB of
// LDS: four workgroups per CU).  B fragment: lane (j = n in the block, k = pixel) reads x[pixel + tap(n)][c(n)], one
// ds_read_b32 at a per-lane constant offset; columns 36 .. 47 read a slot that stays zero.
// BN: the layer's BatchNorm backward rides in the tile load (bn_bwd_apply_kernel's arithmetic, same operation order): the
// kernel reads dy and z (8 B per element) instead of a dz tensor that a separate pass would first write and this one read
// back (12 + 4 B per element moved by the pair).
template <int TH, int TW, bool BN>
__global__ __launch_bounds__(256, 4) void wgrad_c4_kernel(const WgradArgs a) {
  static_assert(TH * TW == 64 && TW % 4 == 0, "64-pixel tiles");
  constexpr int HR = TH + 2, HW = TW + 2, HPX = HR * HW;
  __shared__ __attribute__((aligned(16))) float zL[64 * 64];     // dz tile [64 pixels][64 m]
  __shared__ __attribute__((aligned(16))) f32x4 xL[HPX + 1];     // halo, 4 channels per pixel; slot HPX stays zero
  const float* xf = reinterpret_cast<const float*>(xL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  const int xcd = blockIdx.x & 7, kk_ = blockIdx.x >> 3;
  const int split = (kk_ / a.mn) * 8 + xcd, mni = kk_ % a.mn;
  if (split >= a.nsplit) return;
  const int m0 = mni * 64;
  int offn[3];
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) {
    const int n = nb * 16 + l16, tap = n >> 2, c = n & 3;
    offn[nb] = n < 36 ? ((tap / 3) * HW + tap % 3) * 4 + c : -1;
  }
  f32x4 acc[3];
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (tid == 0) xL[HPX] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 hv, zv[4];
  // (BN) the four channels m0 + 4 * (tid & 15) .. + 3 of every value this thread loads: 256 % 16 == 0
  float bmean[4], binv[4], bgam[4], bbet[4], bmg[4], bmgx[4];
  if constexpr (BN) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = min(m0 + 4 * (tid & 15) + j, a.M - 1);
      bmean[j] = a.bn_mi[m];
      binv[j] = a.bn_mi[a.M + m];
      bgam[j] = a.bn_gamma[m];
      bbet[j] = a.bn_beta[m];
      bmg[j] = (float)a.bn_acc[m] * a.bn_inv_n;
      bmgx[j] = (float)a.bn_acc[a.M + m] * a.bn_inv_n;
    }
  }
  auto load_tile = [&](int tile) {
    const int tx = tile % a.ntx;
    const int ty = (tile / a.ntx) % a.nty;
    const int b = tile / (a.ntx * a.nty);
    const int y0 = ty * TH, x0 = tx * TW;
    hv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tid < HPX) {
      const int hr = tid / HW, hc = tid - hr * HW;
      const int sy = y0 + hr - 1, sx = x0 + hc - 1;
      if (sy >= 0 && sy < a.H && sx >= 0 && sx < a.W)
        hv = *reinterpret_cast<const f32x4*>(a.x + (((long)b * a.H + sy) * a.W + sx) * 4);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256;
      const int q = i & 15, p = i >> 4;
      const int r = p / TW, c = p - r * TW;
      const int y = y0 + r, x = x0 + c, m = m0 + 4 * q;
      zv[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (y < a.H && x < a.W && m < a.M) {
        const long o = (((long)b * a.H + y) * a.W + x) * a.dz_cs + m;
        zv[k] = *reinterpret_cast<const f32x4*>(a.dz + o);
        if constexpr (BN) {
          const f32x4 zz = *reinterpret_cast<const f32x4*>(a.bn_z + o);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float yv = (zz[j] - bmean[j]) * binv[j] * bgam[j] + bbet[j];
            const float g = yv > 0.f ? zv[k][j] : 0.f;
            const float xh = (zz[j] - bmean[j]) * binv[j];
            zv[k][j] = bgam[j] * binv[j] * (g - bmg[j] - xh * bmgx[j]);
          }
        }
      }
    }
  };
  if (split < a.ntiles) load_tile(split);
  for (int tile = split; tile < a.ntiles; tile += a.nsplit) {
    __syncthreads();  // previous tile's fragments consumed
    if (tid < HPX) xL[tid] = hv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256;
      const int q = i & 15, p = i >> 4;
      *reinterpret_cast<f32x4*>(zL + p * 64 + 4 * (q ^ ((p & 1) << 2))) = zv[k];
    }
    __syncthreads();
    if (tile + a.nsplit < a.ntiles) load_tile(tile + a.nsplit);   // in flight during the contraction
#pragma unroll 2
    for (int r = 0; r < TH; ++r) {
#pragma unroll
      for (int c4 = 0; c4 < TW / 4; ++c4) {
        const int pz = r * TW + c4 * 4 + kq;
        const float av = zL[pz * 64 + ((wave * 16 + l16) ^ ((pz & 1) << 4))];
        const int ph4 = (r * HW + c4 * 4 + kq) * 4;     // float index of the window's top-left pixel
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
          const float bv = xf[offn[nb] >= 0 ? ph4 + offn[nb] : HPX * 4];
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[nb], 0, 0, 0);
        }
      }
    }
  }
  // D[i = 4 * (lane / 16) + r][j = lane % 16]: m = m0 + 16 * wave + i, n = 16 * nb + j = 4 * tap + c
#pragma unroll
  for (int nb = 0; nb < 3; ++nb) {
    const int n = nb * 16 + l16, tap = n >> 2, c = n & 3;
    if (n >= 36 || c >= a.N) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wave * 16 + 4 * kq + r;
      if (m < a.M) unsafeAtomicAdd(a.raw + ((long)m * 9 + tap) * a.raw_n + a.n_off + c, acc[nb][r]);
    }
  }
}

template <int TH, int TW, bool BN = false>
int launch_wgrad_c4(WgradArgs a, hipStream_t stream) {
  a.ntx = sfh_cdiv(a.W, TW);
  a.nty = sfh_cdiv(a.H, TH);
  a.ntiles = a.batch * a.ntx * a.nty;
  const int mn = sfh_cdiv(a.M, 64);
  int ns = sfh_cdiv(2048, mn);          // four workgroups per CU: two rounds of 1024
  if (ns > a.ntiles) ns = a.ntiles;
  if (ns < 1) ns = 1;
  if (ns < 8 && a.ntiles >= 8) ns = 8;
  a.nsplit = ns;
  a.mt = mn;
  a.mn = mn;
  const dim3 grid((unsigned)(sfh_cdiv(ns, 8) * 8 * mn));
  hipLaunchKernelGGL((wgrad_c4_kernel<TH, TW, BN>), grid, dim3(256), 0, stream, a);
  return sfh_check_launch("wgrad_c4_kernel");
}

// tile shape (rows x cols, 64 pixels) with the fewest padded pixels; ties go to the widest
static int wgrad_tile(int H, int W) {
  const int th[3] = {2, 4, 8}, tw[3] = {32, 16, 8};
  int best = 0;
  long bc = -1;
  for (int i = 0; i < 3; ++i) {
    const long c = (long)sfh_cdiv(H, th[i]) * sfh_cdiv(W, tw[i]);
    if (bc < 0 || c < bc) { bc = c; best = i; }
  }
  return best;
}

}  // namespace

// =============================================================================== C ABI
extern "C" int sfh_bn_stats(const float* z, int64_t npix, int C, double* acc, void* stream) {
  SFH_REQUIRE(z && acc && npix > 0 && C > 0 && C % 4 == 0, "bn_stats: bad argument");
  const unsigned nb = red_grid((long)npix);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, z, (long)npix, C, acc);
  return sfh_check_launch("bn_stats_kernel");
}

extern "C" int sfh_bn_stats_partials(const double* partial, int rows, int C, double* acc, void* stream) {
  SFH_REQUIRE(partial && acc && rows > 0 && C > 0, "bn_stats_partials: bad argument");
  const int ncol = 2 * C;
  const dim3 grid((unsigned)((ncol + 63) / 64), (unsigned)(rows >= 128 ? 32 : (rows + 3) / 4));
  hipLaunchKernelGGL(bn_stats_partials_kernel, grid, dim3(256), 0, (hipStream_t)stream, partial, rows, ncol, acc);
  return sfh_check_launch("bn_stats_partials_kernel");
}

extern "C" int sfh_bn_finalize(const double* acc, int64_t npix, int C, float eps, float momentum,
                               float* running_mean, float* running_var, float* mean_invstd, int64_t* num_batches_tracked,
                               void* stream) {
  SFH_REQUIRE(acc && mean_invstd && npix > 0 && C > 0, "bn_finalize: bad argument");
  SFH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must come in pairs");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, acc,
                     (long)npix, C, eps, momentum, running_mean, running_var, mean_invstd, (long*)num_batches_tracked);
  return sfh_check_launch("bn_finalize_kernel");
}

extern "C" int sfh_bn_finalize_partials(const double* partial, int rows, int64_t npix, int C, float eps, float momentum,
                                        float* running_mean, float* running_var, float* mean_invstd,
                                        int64_t* num_batches_tracked, void* stream) {
  SFH_REQUIRE(partial && mean_invstd && rows > 0 && npix > 0 && C > 0, "bn_finalize_partials: bad argument");
  SFH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_partials: running stats must come in pairs");
  hipLaunchKernelGGL(bn_finalize_partials_kernel, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, partial,
                     rows, (long)npix, C, eps, momentum, running_mean, running_var, mean_invstd, (long*)num_batches_tracked);
  return sfh_check_launch("bn_finalize_partials_kernel");
}

extern "C" int sfh_bn_apply(const float* z, const float* mean_invstd, const float* gamma, const float* beta,
                            const float* residual, int relu, int64_t npix, int C, float* y, void* y_s3, int W,
                            int split_fmt, uint32_t* overflow, void* stream) {
  SFH_REQUIRE(z && mean_invstd && gamma && beta && (y || y_s3) && npix > 0 && C > 0 && C % 4 == 0, "bn_apply: bad argument");
  if (y_s3) {
    SFH_REQUIRE(C % 32 == 0 && W > 0 && npix % W == 0, "bn_apply: the split copy needs C %% 32 == 0 and npix = rows * W");
    SFH_REQUIRE(split_fmt == SFH_FMT_S3 || split_fmt == SFH_FMT_H2, "bn_apply: split_fmt=%d (S3 or H2)", split_fmt);
    const long nblk = ((long)npix + 63) / 64 * (C / 32);
    SFH_REQUIRE(nblk < (1L << 31) && npix < (1L << 31) - 64, "bn_apply: tensor too large for one launch");
    const dim3 grid((unsigned)nblk);
    if (split_fmt == SFH_FMT_H2)
      hipLaunchKernelGGL(bn_apply_s3_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, z, mean_invstd, gamma, beta,
                         residual, relu, (long)npix, W, C, y, (unsigned short*)y_s3, overflow);
    else
      hipLaunchKernelGGL(bn_apply_s3_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, z, mean_invstd, gamma, beta,
                         residual, relu, (long)npix, W, C, y, (unsigned short*)y_s3, overflow);
    return sfh_check_launch("bn_apply_s3_kernel");
  }
  const long total4 = (long)npix * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z,
                     mean_invstd, gamma, beta, residual, relu, total4, C, y);
  return sfh_check_launch("bn_apply_kernel");
}

extern "C" int sfh_bn_bwd_reduce(const float* dy, const float* y, const float* z, const float* mean_invstd,
                                 const float* gamma, const float* beta, int relu, int64_t npix, int C, double* acc,
                                 void* stream) {
  SFH_REQUIRE(dy && z && mean_invstd && acc && (y || !relu || (gamma && beta)) && npix > 0 && C > 0 && C % 4 == 0,
              "bn_bwd_reduce: bad argument (relu needs y, or gamma and beta to recompute its sign from z)");
  const unsigned nb = red_grid((long)npix);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, y, z, mean_invstd, gamma,
                     beta, relu, (long)npix, C, acc);
  return sfh_check_launch("bn_bwd_reduce_kernel");
}

extern "C" int sfh_bn_bwd_apply(const float* dy, const float* y, const float* z, const float* mean_invstd,
                                const float* gamma, const float* beta, const double* acc, int relu, int64_t npix, int C,
                                float* dz, float* dres, void* dz_s3, int W, int split_fmt, uint32_t* overflow,
                                float* acc_f32, void* stream) {
  SFH_REQUIRE(dy && z && mean_invstd && gamma && acc && (dz || dz_s3) && (y || !relu || beta) && npix > 0 && C > 0 &&
                  C % 4 == 0,
              "bn_bwd_apply: bad argument (relu needs y, or beta to recompute its sign from z)");
  if (dz_s3) {
    SFH_REQUIRE(C % 32 == 0 && W > 0 && npix % W == 0, "bn_bwd_apply: the split copy needs C %% 32 == 0 and npix = rows * W");
    SFH_REQUIRE(split_fmt == SFH_FMT_S3 || split_fmt == SFH_FMT_H2, "bn_bwd_apply: split_fmt=%d (S3 or H2)", split_fmt);
    const long nblk = ((long)npix + 63) / 64 * (C / 32);
    SFH_REQUIRE(nblk < (1L << 31) && npix < (1L << 31) - 64, "bn_bwd_apply: tensor too large for one launch");
    const dim3 grid((unsigned)nblk);
    if (split_fmt == SFH_FMT_H2)
      hipLaunchKernelGGL(bn_bwd_apply_s3_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, dy, y, z, mean_invstd, gamma,
                         acc, relu, (long)npix, W, C, dz, dres, (unsigned short*)dz_s3, overflow, beta, acc_f32);
    else
      hipLaunchKernelGGL(bn_bwd_apply_s3_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, dy, y, z, mean_invstd, gamma,
                         acc, relu, (long)npix, W, C, dz, dres, (unsigned short*)dz_s3, overflow, beta, acc_f32);
    return sfh_check_launch("bn_bwd_apply_s3_kernel");
  }
  const long total4 = (long)npix * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dy, y, z, mean_invstd, gamma, acc, relu, (long)npix, total4, C, dz, dres, beta, acc_f32);
  return sfh_check_launch("bn_bwd_apply_kernel");
}

extern "C" int sfh_colsum(const float* x, int64_t npix, int C, int cs, double* acc, void* stream) {
  SFH_REQUIRE(x && acc && npix > 0 && C > 0 && C % 4 == 0 && cs >= C && cs % 4 == 0, "colsum: bad argument");
  const unsigned nb = red_grid((long)npix);
  hipLaunchKernelGGL(colsum_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, (long)npix, C, cs, acc);
  return sfh_check_launch("colsum_kernel");
}

extern "C" int sfh_s2d_split_colsum(const float* du, int batch, int h, int w, int cout, void* s_split, int split_fmt,
                                    double* acc, int acc_rows, uint32_t* overflow, void* stream) {
  SFH_REQUIRE(du && s_split && acc && batch > 0 && h > 0 && w > 0 && cout > 0 && cout % 8 == 0 && acc_rows >= 1 && acc_rows <= 4096,
              "s2d_split_colsum: bad argument (cout %% 8 == 0, 1 <= acc_rows <= 4096)");
  SFH_REQUIRE(split_fmt == SFH_FMT_S3 || split_fmt == SFH_FMT_H2, "s2d_split_colsum: split_fmt=%d (S3 or H2)", split_fmt);
  const long npix = (long)batch * h * w;
  SFH_REQUIRE(npix < (1L << 31) - 64 && (long)batch * 2 * h < (1L << 31), "s2d_split_colsum: tensor too large for one launch");
  const int nb = 4 * cout / 32;
  const long nchunks = (npix + 63) / 64;
#ifdef SFH_S2D_BLOCKS                          // (experiment build: another number of blocks)
  long per = SFH_S2D_BLOCKS / nb;
#else
  long per = 4096 / nb;                       // about 4096 blocks: 16 per CU, each striding over its share of the chunks
#endif
  if (per < 1) per = 1;
  if (per > nchunks) per = nchunks;
  const dim3 grid((unsigned)(per * nb));
  if (split_fmt == SFH_FMT_H2)
    hipLaunchKernelGGL(s2d_split_colsum_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, du, h, w, cout, npix, nchunks,
                       (int)per, (unsigned short*)s_split, acc, acc_rows, overflow);
  else
    hipLaunchKernelGGL(s2d_split_colsum_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, du, h, w, cout, npix, nchunks,
                       (int)per, (unsigned short*)s_split, acc, acc_rows, overflow);
  return sfh_check_launch("s2d_split_colsum_kernel");
}

extern "C" int sfh_bn_apply_pool(const float* z, const float* mean_invstd, const float* gamma, const float* beta, int batch,
                                 int H, int W, int C, void* y_s3, void* pool_s3, int split_fmt, uint32_t* overflow,
                                 void* stream) {
  SFH_REQUIRE(z && mean_invstd && gamma && beta && y_s3 && pool_s3 && batch > 0 && H >= 2 && W >= 2 && C > 0 && C % 32 == 0,
              "bn_apply_pool: bad argument (C %% 32 == 0, H and W >= 2)");
  SFH_REQUIRE(split_fmt == SFH_FMT_S3 || split_fmt == SFH_FMT_H2, "bn_apply_pool: split_fmt=%d (S3 or H2)", split_fmt);
  const long nquads = (long)batch * ((H + 1) / 2) * ((W + 1) & ~1);   // (row pair, column) slots, width rounded up to even
  const long nblk = (nquads + 63) / 64 * (C / 32);
  SFH_REQUIRE(nblk < (1L << 31) && nquads < (1L << 31) - 64 && (long)batch * H * W < (1L << 31),
              "bn_apply_pool: tensor too large for one launch");
  if (split_fmt == SFH_FMT_H2)
    hipLaunchKernelGGL(bn_apply_pool_s3_kernel<2>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, z, mean_invstd,
                       gamma, beta, H, W, C, nquads, (unsigned short*)y_s3, (unsigned short*)pool_s3, overflow);
  else
    hipLaunchKernelGGL(bn_apply_pool_s3_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, z, mean_invstd,
                       gamma, beta, H, W, C, nquads, (unsigned short*)y_s3, (unsigned short*)pool_s3, overflow);
  return sfh_check_launch("bn_apply_pool_s3_kernel");
}

extern "C" int sfh_pool2_bwd_bn_reduce(const float* z, const float* mean_invstd, const float* gamma, const float* beta,
                                       const float* dpool, int batch, int H, int W, int C, int accumulate, float* dx,
                                       double* acc, void* stream) {
  SFH_REQUIRE(z && mean_invstd && gamma && beta && dpool && dx && acc && batch > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0,
              "pool2_bwd_bn_reduce: bad argument");
  const long nquads = (long)batch * ((H + 1) / 2) * ((W + 1) / 2);
  SFH_REQUIRE(nquads < (1L << 31) && (long)batch * H * W < (1L << 31), "pool2_bwd_bn_reduce: tensor too large for one launch");
  const unsigned nb = red_grid(nquads);
  hipLaunchKernelGGL(pool2_bwd_bn_reduce_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, z, mean_invstd, gamma, beta,
                     dpool, H, W, C, nquads, accumulate, dx, acc);
  return sfh_check_launch("pool2_bwd_bn_reduce_kernel");
}

extern "C" int sfh_maxpool2_fwd(const float* x, float* y, int batch, int H, int W, int C, void* stream) {
  SFH_REQUIRE(x && y && batch > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "maxpool2_fwd: bad argument");
  const int Ho = H / 2, Wo = W / 2;
  const long total4 = (long)batch * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     H, W, C, Ho, Wo, total4, y);
  return sfh_check_launch("maxpool2_fwd_kernel");
}

extern "C" int sfh_maxpool2_bwd(const float* x, const float* dy, float* dx, int batch, int H, int W, int C,
                                int accumulate, void* stream) {
  SFH_REQUIRE(x && dy && dx && batch > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "maxpool2_bwd: bad argument");
  SFH_REQUIRE(accumulate || (H % 2 == 0 && W % 2 == 0),
              "maxpool2_bwd: odd sizes leave a border the kernel does not write; zero dx and pass accumulate=1");
  const int Ho = H / 2, Wo = W / 2;
  const long total4 = (long)batch * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     dy, H, W, C, Ho, Wo, total4, accumulate, dx);
  return sfh_check_launch("maxpool2_bwd_kernel");
}

extern "C" int sfh_slice_add(const float* src, int Hs, int Ws, int cs, int c_off, int oy, int ox, float* dst,
                             int batch, int h, int w, int C, int accumulate, void* stream) {
  SFH_REQUIRE(src && dst && batch > 0 && h > 0 && w > 0 && C > 0 && C % 4 == 0 && cs % 4 == 0 && c_off % 4 == 0 &&
                  c_off >= 0 && c_off + C <= cs && Hs > 0 && Ws > 0,
              "slice_add: bad argument");
  const long total4 = (long)batch * h * w * (C / 4);
  hipLaunchKernelGGL(slice_add_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     Hs, Ws, cs, c_off, oy, ox, h, w, C, total4, accumulate, dst);
  return sfh_check_launch("slice_add_kernel");
}

extern "C" int sfh_zero_stuff2(const float* src, float* dst, int batch, int ho, int wo, int H, int W, int C,
                               void* stream) {
  SFH_REQUIRE(src && dst && batch > 0 && ho > 0 && wo > 0 && C > 0 && C % 4 == 0, "zero_stuff2: bad argument");
  SFH_REQUIRE(H >= 2 * ho - 1 && W >= 2 * wo - 1, "zero_stuff2: destination %dx%d too small for %dx%d", H, W, ho, wo);
  const long total4 = (long)batch * H * W * (C / 4);
  hipLaunchKernelGGL(zero_stuff2_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     ho, wo, H, W, C, total4, dst);
  return sfh_check_launch("zero_stuff2_kernel");
}

extern "C" int sfh_conv_wgrad(const float* dz, int dz_cs, int M, const float* x, int x_cs, int xh, int xw, int N,
                              int pad_top, int pad_left, int batch, int H, int W, int ksize, float* raw, int raw_n,
                              int n_off, void* stream) {
  SFH_REQUIRE(dz && x && raw, "conv_wgrad: null pointer");
  SFH_REQUIRE(batch > 0 && H > 0 && W > 0 && xh > 0 && xw > 0, "conv_wgrad: bad geometry");
  SFH_REQUIRE(M > 0 && M % 4 == 0 && dz_cs % 4 == 0 && dz_cs >= M, "conv_wgrad: M=%d dz_cs=%d", M, dz_cs);
  SFH_REQUIRE(N > 0 && N % 4 == 0 && x_cs % 4 == 0 && x_cs >= N, "conv_wgrad: N=%d x_cs=%d", N, x_cs);
  SFH_REQUIRE(n_off >= 0 && n_off + N <= raw_n, "conv_wgrad: n_off=%d N=%d raw_n=%d", n_off, N, raw_n);
  SFH_REQUIRE(ksize == 1 || ksize == 3 || ksize == 4, "conv_wgrad: ksize %d", ksize);
  SFH_REQUIRE(ksize != 4 || N <= 32, "conv_wgrad: the 4x4 (stem) case supports N <= 32");
  WgradArgs a;
  a.dz = dz; a.dz_cs = dz_cs; a.M = M;
  a.x = x; a.x_cs = x_cs; a.xh = xh; a.xw = xw; a.N = N; a.pad_top = pad_top; a.pad_left = pad_left;
  a.batch = batch; a.H = H; a.W = W;
  a.raw = raw; a.raw_n = raw_n; a.n_off = n_off;
  a.ntx = a.nty = a.ntiles = a.nsplit = a.mt = a.mn = 0;
  a.bn_z = a.bn_mi = a.bn_gamma = a.bn_beta = nullptr; a.bn_acc = nullptr; a.bn_inv_n = 0.f;
  hipStream_t st = (hipStream_t)stream;
  const int t = wgrad_tile(H, W);
#define SFH_WG(KS_, NS_)                                         \
  (t == 0 ? launch_wgrad<KS_, NS_, 2, 32>(a, st)                  \
          : (t == 1 ? launch_wgrad<KS_, NS_, 4, 16>(a, st) : launch_wgrad<KS_, NS_, 8, 8>(a, st)))
  // the network's first layer (3 input channels stored as 4, one source of the frame's own size): n = (tap, channel)
  if (ksize == 3 && N <= 4 && x_cs == 4 && xh == H && xw == W && pad_top == 0 && pad_left == 0)
    return t == 0 ? launch_wgrad_c4<2, 32>(a, st) : (t == 1 ? launch_wgrad_c4<4, 16>(a, st) : launch_wgrad_c4<8, 8>(a, st));
  if (ksize == 3 && N <= 16) return SFH_WG(3, 1);
  if (ksize == 3) return SFH_WG(3, 4);
  if (ksize == 1) return SFH_WG(1, 4);
  return SFH_WG(4, 2);
#undef SFH_WG
}

extern "C" int sfh_conv_wgrad_c4_bn(const float* dy, const float* z, const float* mean_invstd, const float* gamma,
                                    const float* beta, const double* acc, int M, const float* x, int N, int batch, int H,
                                    int W, float* raw, int raw_n, void* stream) {
  SFH_REQUIRE(dy && z && mean_invstd && gamma && beta && acc && x && raw, "conv_wgrad_c4_bn: null pointer");
  SFH_REQUIRE(batch > 0 && H > 0 && W > 0 && M > 0 && M % 4 == 0 && N > 0 && N <= 4 && raw_n >= N,
              "conv_wgrad_c4_bn: M=%d N=%d raw_n=%d", M, N, raw_n);
  WgradArgs a;
  a.dz = dy; a.dz_cs = M; a.M = M;
  a.x = x; a.x_cs = 4; a.xh = H; a.xw = W; a.N = N; a.pad_top = 0; a.pad_left = 0;
  a.batch = batch; a.H = H; a.W = W;
  a.raw = raw; a.raw_n = raw_n; a.n_off = 0;
  a.ntx = a.nty = a.ntiles = a.nsplit = a.mt = a.mn = 0;
  a.bn_z = z; a.bn_mi = mean_invstd; a.bn_gamma = gamma; a.bn_beta = beta; a.bn_acc = acc;
  a.bn_inv_n = 1.0f / (float)((long)batch * H * W);
  hipStream_t st = (hipStream_t)stream;
  const int t = wgrad_tile(H, W);
  return t == 0 ? launch_wgrad_c4<2, 32, true>(a, st)
                : (t == 1 ? launch_wgrad_c4<4, 16, true>(a, st) : launch_wgrad_c4<8, 8, true>(a, st));
}

static int launch_outconv_bwd(const float* x, int cin, const float* w, const float* dlogits_nchw, int nc, int batch, int H,
                              int W, float* dx, double* acc_w, double* acc_b, const float* bn_mi, const float* bn_gamma,
                              const float* bn_beta, double* acc_bn, void* stream) {
  SFH_REQUIRE(x && w && dlogits_nchw && acc_w && acc_b && batch > 0 && H > 0 && W > 0, "outconv_bwd: bad argument");
  SFH_REQUIRE(cin % 4 == 0 && cin >= 4 && cin <= 256, "outconv_bwd: cin=%d (multiple of 4, <= 256)", cin);
  SFH_REQUIRE(nc >= 1 && nc <= 8, "outconv_bwd: nc=%d unsupported (1..8)", nc);
  const long npix = (long)batch * H * W;
  // every workgroup ends with nc * (cin + 1) fp64 atomics on the same addresses: about 1024 workgroups (at 640x360 x 16:
  // 3840 pixels each, 0.49 ms; 1024 pixels each = 3600 workgroups: 0.56 ms; 8192: 0.63 - profiles/r05_s2d_atomics.txt)
  long ppb = (npix / 1024 + 255) / 256 * 256;
  if (ppb < 1024) ppb = 1024;
  const unsigned grid = (unsigned)((npix + ppb - 1) / ppb);
#define SFH_OB(N)                                                                                                  \
  case N:                                                                                                          \
    if (acc_bn)                                                                                                    \
      hipLaunchKernelGGL((outconv_bwd_kernel<N, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, cin, w,   \
                         dlogits_nchw, npix, H * W, dx, acc_w, acc_b, (int)ppb, bn_mi, bn_gamma, bn_beta, acc_bn); \
    else                                                                                                           \
      hipLaunchKernelGGL((outconv_bwd_kernel<N, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, cin, w,  \
                         dlogits_nchw, npix, H * W, dx, acc_w, acc_b, (int)ppb, bn_mi, bn_gamma, bn_beta, acc_bn); \
    break;
  switch (nc) { SFH_OB(1) SFH_OB(2) SFH_OB(3) SFH_OB(4) SFH_OB(5) SFH_OB(6) SFH_OB(7) SFH_OB(8) }
#undef SFH_OB
  return sfh_check_launch("outconv_bwd_kernel");
}

extern "C" int sfh_outconv_bwd(const float* x, int cin, const float* w, const float* dlogits_nchw, int nc,
                               int batch, int H, int W, float* dx, double* acc_w, double* acc_b, void* stream) {
  return launch_outconv_bwd(x, cin, w, dlogits_nchw, nc, batch, H, W, dx, acc_w, acc_b, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int sfh_outconv_bwd_bn(const float* z, const float* mean_invstd, const float* gamma, const float* beta, int cin,
                                  const float* w, const float* dlogits_nchw, int nc, int batch, int H, int W, float* dx,
                                  double* acc_w, double* acc_b, double* acc_bn, void* stream) {
  SFH_REQUIRE(mean_invstd && gamma && beta && acc_bn && dx, "outconv_bwd_bn: null pointer");
  return launch_outconv_bwd(z, cin, w, dlogits_nchw, nc, batch, H, W, dx, acc_w, acc_b, mean_invstd, gamma, beta, acc_bn, stream);
}

extern "C" int sfh_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int batch, int H, int W, int C,
                                    void* stream) {
  SFH_REQUIRE(x && dy && dx && batch > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool3x3s2_bwd: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total4 = (long)batch * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, dy, H, W, C, Ho, Wo, total4, dx);
  return sfh_check_launch("maxpool3x3s2_bwd_kernel");
}

extern "C" int sfh_avgpool_linear_bwd(const float* x, const float* w, const float* dout, int batch, int H, int W,
                                      int C, int nout, float* dx, double* acc_w, double* acc_b, void* stream) {
  SFH_REQUIRE(x && w && dout && dx && acc_w && acc_b && batch > 0 && H > 0 && W > 0 && C > 0 && nout > 0 && nout <= 256,
              "avgpool_linear_bwd: bad argument");
  hipLaunchKernelGGL(avgpool_linear_bwd_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, x, w, dout,
                     H * W, C, nout, dx, acc_w, acc_b);
  return sfh_check_launch("avgpool_linear_bwd_kernel");
}

extern "C" int sfh_stem_bwd_data(const float* dz, const float* w, int cin, int c_off, int nc, int batch, int H,
                                 int W, float* dlogits_nchw, void* stream) {
  SFH_REQUIRE(dz && w && dlogits_nchw && batch > 0 && batch <= 65535 && H > 0 && W > 0, "stem_bwd_data: bad argument");
  SFH_REQUIRE(nc >= 1 && nc <= 8 && c_off >= 0 && c_off + nc <= cin, "stem_bwd_data: c_off=%d nc=%d cin=%d", c_off, nc, cin);
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  // 1-D launch: the kernel maps workgroup -> (frame, 4-row band, 128-column tile) so that an XCD owns a contiguous range
  const long total = (long)sfh_cdiv(W, 128) * sfh_cdiv(H, 4) * batch;
  SFH_REQUIRE(total < (1L << 28), "stem_bwd_data: grid too large");
  const dim3 grid((unsigned)(((total + 7) >> 3) << 3));
  if (nc <= 4) {
    hipLaunchKernelGGL(stem_bwd_data_kernel<1>, grid, dim3(256), 49 * 64 * 4 * sizeof(float), (hipStream_t)stream, dz, w,
                       cin, c_off, nc, H, W, Ho, Wo, dlogits_nchw, batch);
  } else {
    sfh_allow_big_lds(reinterpret_cast<const void*>(&stem_bwd_data_kernel<2>));
    hipLaunchKernelGGL(stem_bwd_data_kernel<2>, grid, dim3(256), 49 * 64 * 8 * sizeof(float), (hipStream_t)stream, dz, w,
                       cin, c_off, nc, H, W, Ho, Wo, dlogits_nchw, batch);
  }
  return sfh_check_launch("stem_bwd_data_kernel");
}

extern "C" int sfh_train_losses(const float* logits_nchw, const int64_t* gt_mask, const float* weight,
                                const float* warp_mask, int nc, int batch, int H, int W, float lambda_seg,
                                float lambda_rec, int rec_mse, float lambda_cons, int focal_flags,
                                float* dlogits_nchw, float* dwarp, double* loss3, void* stream) {
  SFH_REQUIRE(logits_nchw && gt_mask && weight && dlogits_nchw && loss3 && batch > 0 && H > 0 && W > 0,
              "train_losses: bad argument");
  SFH_REQUIRE(nc >= 2 && nc <= 8, "train_losses: nc=%d unsupported (2..8)", nc);
  SFH_REQUIRE(warp_mask || (lambda_rec == 0.f && lambda_cons == 0.f), "train_losses: rec/consistency need the warp mask");
  LossArgs a;
  a.logits = logits_nchw; a.gt = (const long*)gt_mask; a.weight = weight; a.warp = warp_mask;
  a.nc = nc; a.HW = H * W; a.npix = (long)batch * H * W; a.batch = batch;
  a.l_seg = lambda_seg; a.l_rec = lambda_rec; a.l_cons = lambda_cons; a.rec_mse = rec_mse;
  a.seg_focal = focal_flags & 1; a.cons_focal = (focal_flags >> 1) & 1;
  a.dlogits = dlogits_nchw; a.dwarp = dwarp; a.loss = loss3;
  long nb = (a.npix + 255) / 256;
  if (nb > 2048) nb = 2048;
#define SFH_TL(N)                                                                                         \
  case N:                                                                                                 \
    hipLaunchKernelGGL(train_losses_kernel<N>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a); \
    break;
  switch (nc) { SFH_TL(2) SFH_TL(3) SFH_TL(4) SFH_TL(5) SFH_TL(6) SFH_TL(7) SFH_TL(8) }
#undef SFH_TL
  return sfh_check_launch("train_losses_kernel");
}

extern "C" int sfh_reproj_loss(const float* poi, const float* gt_poi, const float* nonzeros,
                               const float* num_nonzero, int batch, int npts, float lambda, float* dpoi,
                               double* loss, void* stream) {
  SFH_REQUIRE(poi && gt_poi && nonzeros && num_nonzero && dpoi && loss && batch > 0 && npts > 0,
              "reproj_loss: bad argument");
  hipLaunchKernelGGL(reproj_loss_kernel, dim3((unsigned)sfh_cdiv(batch, 64)), dim3(64), 0, (hipStream_t)stream, poi,
                     gt_poi, nonzeros, num_nonzero, batch, npts, lambda, dpoi, loss);
  return sfh_check_launch("reproj_loss_kernel");
}

extern "C" int sfh_multi_copy(const void* tensor_table, const void* chunk_table, int nchunks, float scale, void* stream) {
  SFH_REQUIRE(tensor_table && chunk_table && nchunks > 0, "multi_copy: bad argument");
  if (scale == 1.f)
    hipLaunchKernelGGL(multi_copy_kernel<false>, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream,
                       (const CopyTensor*)tensor_table, (const OptChunk*)chunk_table, 1.f);
  else
    hipLaunchKernelGGL(multi_copy_kernel<true>, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream,
                       (const CopyTensor*)tensor_table, (const OptChunk*)chunk_table, scale);
  return sfh_check_launch("multi_copy_kernel");
}

extern "C" int sfh_sgd_step(const void* tensor_table, const void* chunk_table, int nchunks, float lr, float weight_decay,
                            float momentum, float clip_value, float grad_scale, void* stream) {
  SFH_REQUIRE(tensor_table && chunk_table && nchunks > 0, "sgd_step: bad argument");
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)tensor_table,
                     (const OptChunk*)chunk_table, lr, weight_decay, momentum, clip_value, grad_scale);
  return sfh_check_launch("sgd_kernel");
}

extern "C" int sfh_adam_step(const void* tensor_table, const void* chunk_table, int nchunks, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float clip_value, float grad_scale, int step, void* stream) {
  SFH_REQUIRE(tensor_table && chunk_table && nchunks > 0 && step >= 1, "adam_step: bad argument (step counts from 1)");
  SFH_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "adam_step: betas must lie in [0, 1)");
  const double bias1 = 1.0 - pow((double)beta1, (double)step), bias2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, (const OptTensor*)tensor_table,
                     (const OptChunk*)chunk_table, (float)((double)lr / bias1), beta1, beta2, eps, weight_decay, clip_value,
                     grad_scale, (float)sqrt(bias2));
  return sfh_check_launch("adam_kernel");
}

extern "C" int sfh_uv_loss(const float* uv, const float* gt_uv, const float* weight, int nweights, int batch, int C, int H,
                           int W, float lambda, int mse, float* duv, double* loss, void* stream) {
  SFH_REQUIRE(uv && gt_uv && weight && duv && loss && batch > 0 && C > 0 && H > 0 && W > 0, "uv_loss: bad argument");
  SFH_REQUIRE(nweights == 1 || nweights == W,
              "uv_loss: %d per-sample weights cannot be broadcast along the last axis of the (B, W) = (%d, %d) loss map "
              "(models/losses.py:38-39 multiplies torch.mean(loss, dim=(1, 2)) of a 4-D map by the weights)", nweights, batch, W);
  const long total = (long)batch * C * H * W;
  long nb = (total + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(uv_loss_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, uv, gt_uv, weight,
                     nweights == W && nweights != 1 ? 1 : 0, C, H, W, total, lambda, mse, duv, loss,
                     1.0f / ((float)batch * (float)C * (float)H * (float)W));
  return sfh_check_launch("uv_loss_kernel");
}

extern "C" int sfh_multi_copy_dscale(const void* tensor_table, const void* chunk_table, int nchunks, const float* scale_dev,
                                     void* stream) {
  SFH_REQUIRE(tensor_table && chunk_table && nchunks > 0 && scale_dev, "multi_copy_dscale: bad argument");
  hipLaunchKernelGGL(multi_copy_kernel<true>, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream,
                     (const CopyTensor*)tensor_table, (const OptChunk*)chunk_table, 1.f, scale_dev);
  return sfh_check_launch("multi_copy_kernel");
}

// The power-of-two scale a training step's backward pass is carried with in the two-plane fp16 format, chosen ON THE DEVICE from
// the largest head / theta gradient (words of sfh_multi_absminmax: nheads head tensors, then theta if has_theta) - what
// training.run_backward computed on the host behind a read-back until round 6.  Same rule: the largest head gradient goes to
// [2^(1+shift), 2^(2+shift)) (without heads: theta's to [2^(12+shift), 2^(13+shift))); if theta's gradient would then sit below
// 2^-16 the scale is raised until it does not, provided the head gradients stay below 2^6.  scale2 = {S, 1 / S}.  Raises bit 1
// of *overflow for a non-finite seed and bit 2 if no scale fits both (the caller reads the word at the end of the step).
__global__ void grad_scale_kernel(const uint32_t* __restrict__ words, int nheads, int has_theta, int shift,
                                  float* __restrict__ scale2, uint32_t* __restrict__ overflow) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float mh = 0.f, mt = 0.f;
  bool finite = true;
  for (int i = 0; i < nheads; ++i) {
    const uint32_t b = words[2 * i];
    finite &= b < 0x7F800000u;
    mh = fmaxf(mh, __uint_as_float(b));
  }
  if (has_theta) {
    const uint32_t b = words[2 * nheads];
    finite &= b < 0x7F800000u;
    mt = __uint_as_float(b);
  }
  float S = 1.f;
  if (!finite) {
    atomicOr(overflow, 2u);
  } else {
    const float m = nheads > 0 ? mh : mt;
    const int target = (nheads > 0 ? 2 : 13) + shift;
    int e = 0;
    if (m > 0.f) {
      (void)frexpf(m, &e);                    // m = f * 2^e, 0.5 <= f < 1  ->  m * S in [2^(target-1), 2^target)
      S = ldexpf(1.f, target - e);
    }
    if (nheads > 0 && mt > 0.f && mt * S < 1.52587890625e-05f) {
      (void)frexpf(mt, &e);
      const float S2 = ldexpf(1.f, -16 - e + 1);
      if (mh * S2 >= 64.f) atomicOr(overflow, 4u);
      else S = S2;
    }
  }
  scale2[0] = S;
  scale2[1] = 1.f / S;
}

extern "C" int sfh_grad_scale(const uint32_t* words, int nheads, int has_theta, int shift, float* scale2, uint32_t* overflow,
                              void* stream) {
  SFH_REQUIRE(words && scale2 && overflow && nheads >= 0 && (nheads > 0 || has_theta), "grad_scale: bad argument");
  hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, words, nheads, has_theta, shift, scale2,
                     overflow);
  return sfh_check_launch("grad_scale_kernel");
}

extern "C" int sfh_rmsprop_step(const void* tensor_table, const void* chunk_table, int nchunks, float lr,
                                float alpha, float eps, float weight_decay, float momentum, float clip_value,
                                float grad_scale, void* stream) {
  SFH_REQUIRE(tensor_table && chunk_table && nchunks > 0, "rmsprop_step: bad argument");
  hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream,
                     (const OptTensor*)tensor_table, (const OptChunk*)chunk_table, lr, alpha, eps, weight_decay,
                     momentum, clip_value, grad_scale);
  return sfh_check_launch("rmsprop_kernel");
}
