// pointwise.hip - the bandwidth-bound kernels around the conv stack (all HBM-roofline work):
// layout conversion, BatchNorm folding, OutConv (+argmax, +STN input assembly),
// consistency cross-entropy, ResNet max-pool and avg-pool+Linear.
#include <math.h>

#include "common.h"

namespace {

// ----------------------------------------------------------------- layout conversion
// (B,C,H,W) -> (B,H,W,cs): one thread per pixel, coalesced reads per channel plane,
// one 16-byte store per 4 channels.
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                    int C, int HW, int cs, long npix) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const long b = p / HW, i = p - b * HW;
  const float* s = src + b * (long)C * HW + i;
  float* d = dst + p * cs;
  for (int c4 = 0; c4 < cs; c4 += 4) {
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (c4 + j < C) ? s[(long)(c4 + j) * HW] : 0.f;
    *reinterpret_cast<f32x4*>(d + c4) = v;
  }
}

// uint8 HWC frames -> float32 CHW in [0,1]: the dataset's `img.transpose(2,0,1) / 255` then
// FloatTensor (utils/dataset.py:154-159, :323-330) on the GPU; IEEE division, so identical bits.
__global__ void u8hwc_to_f32nchw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int C, int HW,
                                        long npix) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const long b = p / HW, i = p - b * HW;
  const uint8_t* s = src + p * C;
  float* d = dst + b * (long)C * HW + i;
  for (int c = 0; c < C; ++c) d[(long)c * HW] = (float)s[c] / 255.0f;
}

// cv2.resize(frame, (W, H), interpolation=cv2.INTER_AREA) for an exact 2x downscale of uint8 frames
// (utils/dataset.py:312-316: frames wider than the target use INTER_AREA) followed by the same /255:
// OpenCV's 2x2 area fast path is (a + b + c + d + 2) >> 2 per channel.
__global__ void u8hwc_area2_to_f32nchw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int C, int H,
                                              int W, long npix) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one output pixel
  if (p >= npix) return;
  const int HW = H * W;
  const long b = p / HW;
  const int i = (int)(p - b * HW), y = i / W, x = i - y * W;
  const long rs = 2L * W * C;  // source row stride in bytes
  const uint8_t* s = src + (b * 2 * H + 2 * y) * rs + 2L * x * C;
  float* d = dst + b * (long)C * HW + i;
  for (int c = 0; c < C; ++c) {
    const int v = ((int)s[c] + (int)s[C + c] + (int)s[rs + c] + (int)s[rs + C + c] + 2) >> 2;
    d[(long)c * HW] = (float)v / 255.0f;
  }
}

// The same for any other integer factor k (1920x1080 -> 640x360 is k = 3): OpenCV's resizeAreaFast_ sums the k x k block in
// int and stores saturate_cast<uchar>(sum * scale) with scale = 1.f / (k * k) evaluated in float, i.e. the float product
// rounded half to even (lrint).  The 2x2 case above is OpenCV's own special case ((a + b + c + d + 2) >> 2), not this rule.
__global__ void u8hwc_areak_to_f32nchw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int C, int H,
                                              int W, int k, int ky, float scale, long npix) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one output pixel
  if (p >= npix) return;
  const int HW = H * W;
  const long b = p / HW;
  const int i = (int)(p - b * HW), y = i / W, x = i - y * W;
  const long rs = (long)k * W * C;  // source row stride in bytes (k = the horizontal factor, ky the vertical one)
  const uint8_t* s = src + (b * ky * H + (long)ky * y) * rs + (long)k * x * C;
  float* d = dst + b * (long)C * HW + i;
  int sum[4] = {0, 0, 0, 0};
  for (int dy = 0; dy < ky; ++dy)
    for (int dx = 0; dx < k; ++dx) {
      const uint8_t* q = s + dy * rs + (long)dx * C;
      for (int c = 0; c < C; ++c) sum[c] += (int)q[c];
    }
  for (int c = 0; c < C; ++c) {
    int v = __float2int_rn(__fmul_rn((float)sum[c], scale));
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    d[(long)c * HW] = (float)v / 255.0f;
  }
}

// Generic INTER_AREA downscale (non-integer factors): OpenCV's resizeArea_ with the per-axis tables of computeResizeAreaTab.
// One thread per destination pixel; per source row j of its y entries: buf = ((0 + S*a0) + S*a1) + ... over its x entries,
// sum = beta0 * buf0, then sum += beta_j * buf_j - products and sums individually rounded, in table order, as the row-buffer
// loop of resizeArea_ does (no FMA contraction: the file is built with -ffp-contract=off and the intrinsics pin it).
__global__ void u8hwc_area_to_f32nchw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int C, int Hs, int Ws,
                                             int Hd, int Wd, const int* __restrict__ xofs, const int* __restrict__ xsi,
                                             const float* __restrict__ xalpha, const int* __restrict__ yofs,
                                             const int* __restrict__ ysi, const float* __restrict__ ybeta, long npix) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const int HW = Hd * Wd;
  const long b = p / HW;
  const int i = (int)(p - b * HW), y = i / Wd, x = i - y * Wd;
  const uint8_t* s = src + b * (long)Hs * Ws * C;
  float sum[4] = {0.f, 0.f, 0.f, 0.f};
  const int x0 = xofs[x], x1 = xofs[x + 1];
  for (int j = yofs[y]; j < yofs[y + 1]; ++j) {
    const uint8_t* row = s + (long)ysi[j] * Ws * C;
    float buf[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = x0; k < x1; ++k) {
      const uint8_t* q = row + (long)xsi[k] * C;
      const float a = xalpha[k];
      for (int c = 0; c < C; ++c) buf[c] = __fadd_rn(buf[c], __fmul_rn((float)q[c], a));
    }
    const float beta = ybeta[j];
    if (j == yofs[y]) {
      for (int c = 0; c < C; ++c) sum[c] = __fmul_rn(beta, buf[c]);
    } else {
      for (int c = 0; c < C; ++c) sum[c] = __fadd_rn(sum[c], __fmul_rn(beta, buf[c]));
    }
  }
  float* d = dst + b * (long)C * HW + i;
  for (int c = 0; c < C; ++c) {
    int v = __float2int_rn(sum[c]);
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    d[(long)c * HW] = (float)v / 255.0f;
  }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                    int C, int HW, int cs, long npix) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const long b = p / HW, i = p - b * HW;
  const float* s = src + p * cs;
  float* d = dst + b * (long)C * HW + i;
  for (int c4 = 0; c4 < C; c4 += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(s + c4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c4 + j < C) d[(long)(c4 + j) * HW] = v[j];
  }
}

__global__ void space_to_depth2_kernel(const float* __restrict__ src, float* __restrict__ dst, int H,
                                       int W, int cs, int H2, int W2, long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the output
  if (idx >= total4) return;
  const int c4n = cs;  // float4 per output pixel = 4*cs/4
  const int q = idx % c4n;
  long r = idx / c4n;
  const int X = r % W2; r /= W2;
  const int Y = r % H2;
  const long b = r / H2;
  const int par = (q * 4) / cs, c = q * 4 - par * cs;
  const int y = 2 * Y + (par >> 1), x = 2 * X + (par & 1);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (y < H && x < W) v = *reinterpret_cast<const f32x4*>(src + ((b * H + y) * W + x) * cs + c);
  reinterpret_cast<f32x4*>(dst)[idx] = v;
}

// ----------------------------------------------------------------- resizing (K12, A3b)
// Source index arithmetic of ATen's upsample kernels (area_pixel_compute_source_index):
//   align_corners: src = dst * (in-1)/(out-1);  else: src = max(0, (dst+0.5)*in/out - 0.5)
__device__ __forceinline__ void bilinear_taps(int dst, int in, int out, int align, int& i0, int& i1, float& l1) {
  float src;
  if (align) {
    const float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    src = sc * (float)dst;
  } else {
    const float sc = (float)in / (float)out;
    src = sc * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

// planes = B*C images of (hs, ws) -> (hd, wd): F.interpolate(x, size, mode='bilinear') on NCHW
__global__ void resize_bilinear_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int hs,
                                            int ws, int hd, int wd, int align, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = i % wd;
  const long r = i / wd;
  const int y = r % hd;
  const long pl = r / hd;
  int y0, y1, x0, x1;
  float ly, lx;
  bilinear_taps(y, hs, hd, align, y0, y1, ly);
  bilinear_taps(x, ws, wd, align, x0, x1, lx);
  const float* p = src + pl * (long)hs * ws;
  const float top = (1.f - lx) * p[y0 * ws + x0] + lx * p[y0 * ws + x1];
  const float bot = (1.f - lx) * p[y1 * ws + x0] + lx * p[y1 * ws + x1];
  dst[i] = (1.f - ly) * top + ly * bot;
}

// F.interpolate(x, size, mode='nearest') on NCHW: src = min(floor(dst * in/out), in-1)
__global__ void resize_nearest_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int hs,
                                           int ws, int hd, int wd, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int x = i % wd;
  const long r = i / wd;
  const int y = r % hd;
  const long pl = r / hd;
  const int ys = min((int)floorf(y * ((float)hs / hd)), hs - 1);
  const int xs = min((int)floorf(x * ((float)ws / wd)), ws - 1);
  dst[i] = src[pl * (long)hs * ws + ys * ws + xs];
}

// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) on fp32 NHWC (unet_parts.py:49)
__global__ void upsample2x_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int C,
                                       long total4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the output
  if (i >= total4) return;
  const int c4n = C >> 2;
  const int c4 = i % c4n;
  long r = i / c4n;
  const int x = r % (2 * W); r /= (2 * W);
  const int y = r % (2 * H);
  const long b = r / (2 * H);
  int y0, y1, x0, x1;
  float ly, lx;
  bilinear_taps(y, H, 2 * H, 1, y0, y1, ly);
  bilinear_taps(x, W, 2 * W, 1, x0, x1, lx);
  const float* p = src + b * (long)H * W * C + 4 * c4;
  const f32x4 v00 = *reinterpret_cast<const f32x4*>(p + ((long)y0 * W + x0) * C);
  const f32x4 v01 = *reinterpret_cast<const f32x4*>(p + ((long)y0 * W + x1) * C);
  const f32x4 v10 = *reinterpret_cast<const f32x4*>(p + ((long)y1 * W + x0) * C);
  const f32x4 v11 = *reinterpret_cast<const f32x4*>(p + ((long)y1 * W + x1) * C);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float top = (1.f - lx) * v00[j] + lx * v01[j];
    const float bot = (1.f - lx) * v10[j] + lx * v11[j];
    o[j] = (1.f - ly) * top + ly * bot;
  }
  reinterpret_cast<f32x4*>(dst)[i] = o;
}

// Backward of upsample2x_nhwc_kernel: every input pixel gathers from the output pixels whose two
// taps include it (same index arithmetic as the forward, so the weights are identical).
__device__ __forceinline__ void up2_range(int i, int in, int& lo, int& hi) {
  // outputs o with tap index i have src = o*(in-1)/(2in-1) in (i-1, i+1)
  if (in <= 1) { lo = 0; hi = 2 * in - 1; return; }
  lo = max(0, (int)(((long)(i - 1) * (2 * in - 1)) / (in - 1)) - 1);
  hi = min(2 * in - 1, (int)(((long)(i + 1) * (2 * in - 1) + in - 2) / (in - 1)) + 1);
}

__global__ void upsample2x_nhwc_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int C,
                                           long total4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of the input gradient
  if (i >= total4) return;
  const int c4n = C >> 2;
  const int c4 = i % c4n;
  long r = i / c4n;
  const int x = r % W; r /= W;
  const int y = r % H;
  const long b = r / H;
  int ylo, yhi, xlo, xhi;
  up2_range(y, H, ylo, yhi);
  up2_range(x, W, xlo, xhi);
  const float* p = dy + b * (long)(2 * H) * (2 * W) * C + 4 * c4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int oy = ylo; oy <= yhi; ++oy) {
    int y0, y1;
    float ly;
    bilinear_taps(oy, H, 2 * H, 1, y0, y1, ly);
    const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int ox = xlo; ox <= xhi; ++ox) {
      int x0, x1;
      float lx;
      bilinear_taps(ox, W, 2 * W, 1, x0, x1, lx);
      const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
      if (wx == 0.f) continue;
      const f32x4 g = *reinterpret_cast<const f32x4*>(p + ((long)oy * (2 * W) + ox) * C);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += wy * wx * g[j];
    }
  }
  reinterpret_cast<f32x4*>(dx)[i] = acc;
}

// Backward of resize_nearest_nchw_kernel: dx[src] = sum of dy over the destination pixels that read src.
__global__ void resize_nearest_nchw_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int hs, int ws,
                                               int hd, int wd, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one source pixel
  if (i >= total) return;
  const int xs = i % ws;
  const long r = i / ws;
  const int ys = r % hs;
  const long pl = r / hs;
  const float fy = (float)hs / hd, fx = (float)ws / wd;
  // candidates: dst in [floor(src/f) - 1, ceil((src+1)/f) + 1]; keep those whose forward index is src
  const int ylo = max(0, (int)floorf(ys / fy) - 1), yhi = min(hd - 1, (int)ceilf((ys + 1) / fy) + 1);
  const int xlo = max(0, (int)floorf(xs / fx) - 1), xhi = min(wd - 1, (int)ceilf((xs + 1) / fx) + 1);
  float acc = 0.f;
  for (int y = ylo; y <= yhi; ++y) {
    if (min((int)floorf(y * fy), hs - 1) != ys) continue;
    for (int x = xlo; x <= xhi; ++x)
      if (min((int)floorf(x * fx), ws - 1) == xs) acc += dy[(pl * hd + y) * (long)wd + x];
  }
  dx[i] = acc;
}

// ----------------------------------------------------------------- Up block weight composition
// conv3x3 tap k (0..2) at output parity p (0/1) reads the up-sampled row 2Y + p + k - 1 = 2(Y + s) + d:
//   s = floor((p + k - 1) / 2), d = (p + k - 1) mod 2;   window slot a = s + (1 - p) in {0, 1}
__device__ __forceinline__ void up_tap(int p, int k, int& a, int& d) {
  const int t = p + k - 1;             // -1 .. 2
  const int s = t < 0 ? -1 : t >> 1;
  d = t & 1;
  a = s + 1 - p;
}

// w2[(q*cout + co)][cx][a][b] = sum over the conv taps that land in window slot (a,b) and over the
// intermediate channels cu of wconv[co][c0+cu][ky][kx] * wt[cx][cu][dy][dx]
// One workgroup = 16 output channels x 16 low-resolution channels of one output parity q; the two weight slabs
// of 32 intermediate channels are staged through LDS with coalesced loads (the tensors' own layouts have the
// channel at stride 9 / 4 floats).  Every thread keeps the nine per-tap sums in cu order and adds them into the
// four window slots in (ky, kx) order.
constexpr int kUpCu = 32;

template <int PY, int PX>
__device__ __forceinline__ void compose_up_tile(const float (*swc)[kUpCu * 9 + 1], const float (*swt)[kUpCu * 4 + 4],
                                                int col, int xl, int n, float (&s)[9]) {
  for (int cu = 0; cu < n; ++cu) {
    const float4 t4 = *reinterpret_cast<const float4*>(&swt[xl][cu * 4]);
    const float t[4] = {t4.x, t4.y, t4.z, t4.w};
    const float* w = &swc[col][cu * 9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int dy = (PY + ky - 1) & 1, dx = (PX + kx - 1) & 1;
        s[ky * 3 + kx] += w[ky * 3 + kx] * t[dy * 2 + dx];
      }
  }
}

__global__ __launch_bounds__(256) void compose_up_weights_kernel(const float* __restrict__ wc, int cout, int c0, int c1,
                                                                 const float* __restrict__ wt, int cx,
                                                                 float* __restrict__ w2) {
  __shared__ float swc[16][kUpCu * 9 + 1];
  __shared__ __attribute__((aligned(16))) float swt[16][kUpCu * 4 + 4];
  const int tid = threadIdx.x, xl = tid & 15, col = tid >> 4;
  const int x0 = blockIdx.x * 16, co0 = blockIdx.y * 16, q = blockIdx.z;
  const int py = q >> 1, px = q & 1, cin = c0 + c1;
  float s[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int cu0 = 0; cu0 < c1; cu0 += kUpCu) {
    const int n = min(kUpCu, c1 - cu0);
    __syncthreads();
    for (int i = tid; i < 16 * n * 9; i += 256) {
      const int r = i / (n * 9), k = i - r * (n * 9);
      swc[r][k] = wc[((long)(co0 + r) * cin + c0 + cu0) * 9 + k];
    }
    for (int i = tid; i < 16 * n * 4; i += 256) {
      const int r = i / (n * 4), k = i - r * (n * 4);
      swt[r][k] = wt[((long)(x0 + r) * c1 + cu0) * 4 + k];
    }
    __syncthreads();
    switch (q) {
      case 0: compose_up_tile<0, 0>(swc, swt, col, xl, n, s); break;
      case 1: compose_up_tile<0, 1>(swc, swt, col, xl, n, s); break;
      case 2: compose_up_tile<1, 0>(swc, swt, col, xl, n, s); break;
      default: compose_up_tile<1, 1>(swc, swt, col, xl, n, s); break;
    }
  }
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    int ay, dy;
    up_tap(py, ky, ay, dy);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      int ax, dx;
      up_tap(px, kx, ax, dx);
      const float v = s[ky * 3 + kx];
      if (ay == 0 && ax == 0) acc[0][0] += v;
      else if (ay == 0) acc[0][1] += v;
      else if (ax == 0) acc[1][0] += v;
      else acc[1][1] += v;
    }
  }
  float4* out = reinterpret_cast<float4*>(w2 + (((long)q * cout + co0 + col) * cx + x0 + xl) * 4);
  *out = make_float4(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);
}

// shift_border[cls][cv] = shift[cv] + scale[cv] * (sum over the conv taps that fall inside the up-sampled tensor
// of sum_cu wconv[co][c0+cu][ky][kx] * bt[cu]),  cv = q*cout + co, cls = 4*(row class) + (col class):
// class 0: first row/col of the up-sampled tensor (tap 0 outside), 2: its last (tap 2 outside), 3: the padded
// row/col after it (only tap 0 inside), 1: interior.  The transposed conv's bias seen through the zero-padded
// 3x3 conv, folded behind the BatchNorm scale.
__device__ __forceinline__ bool up_tap_inside(int cls, int k) {
  return cls == 1 || (cls == 0 && k != 0) || (cls == 2 && k != 2) || (cls == 3 && k == 0);
}

__global__ void compose_up_bias_kernel(const float* __restrict__ wc, int cout, int c0, int c1,
                                       const float* __restrict__ bt, const float* __restrict__ scale,
                                       const float* __restrict__ shift, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 16 * 4 * cout) return;
  const int cv = idx % (4 * cout), cls = idx / (4 * cout);
  const int co = cv % cout;
  const int cy = cls >> 2, cxx = cls & 3;
  const int cin = c0 + c1;
  float acc = 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    if (!up_tap_inside(cy, ky)) continue;
    for (int kx = 0; kx < 3; ++kx) {
      if (!up_tap_inside(cxx, kx)) continue;
      const float* wcp = wc + (((long)co * cin + c0) * 3 + ky) * 3 + kx;
      float s = 0.f;
      for (int cu = 0; cu < c1; ++cu) s += wcp[(long)cu * 9] * bt[cu];
      acc += s;
    }
  }
  out[idx] = shift[cv] + scale[cv] * acc;
}

// ----------------------------------------------------------------- BatchNorm folding
__global__ void fold_bn_kernel(const float* conv_bias, const float* gamma, const float* beta,
                               const float* mean, const float* var, float eps, int n, int repeat,
                               float* scale, float* shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * repeat) return;
  const int c = i % n;
  const float cb = conv_bias ? conv_bias[c] : 0.f;
  if (gamma) {
    const float invstd = 1.0f / sqrtf(var[c] + eps);
    const float a = gamma[c] * invstd;
    scale[i] = a;
    shift[i] = (cb - mean[c]) * a + beta[c];
  } else {
    scale[i] = 1.0f;
    shift[i] = cb;
  }
}

// ----------------------------------------------------------------- OutConv 1x1 (cin -> nc<=8)
// 128 pixels per block; two threads per pixel, each reducing half of the input channels
// straight from the NHWC row (16-byte loads), combined with one DPP shuffle.
template <int NC>
__global__ __launch_bounds__(256) void outconv_kernel(
    const float* __restrict__ x, int cin, const float* __restrict__ w, const float* __restrict__ bias,
    long npix, int HW, float* __restrict__ logits, uint8_t* __restrict__ amax,
    float* __restrict__ stn_in, int stn_cs, const float* __restrict__ frame, int frame_cs) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [NC][cin]
  for (int i = threadIdx.x; i < NC * cin; i += 256) wl[i] = w[i];
  __syncthreads();
  const long p = (long)blockIdx.x * 128 + (threadIdx.x >> 1);
  const int half = threadIdx.x & 1;
  const bool live = p < npix;
  float acc[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) acc[k] = 0.f;
  if (live) {
    const int ch = cin >> 1;  // channels per half (multiple of 4)
    const float* xp = x + p * cin + half * ch;
    const float* wp = wl + half * ch;
    for (int c = 0; c < ch; c += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + c);
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const f32x4 ww = *reinterpret_cast<const f32x4*>(wp + k * cin + c);
        acc[k] += v[0] * ww[0] + v[1] * ww[1] + v[2] * ww[2] + v[3] * ww[3];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) acc[k] += __shfl_xor(acc[k], 1);
  if (!live || half) return;
#pragma unroll
  for (int k = 0; k < NC; ++k) acc[k] += bias[k];
  const long b = p / HW, i = p - b * HW;
  if (logits) {
#pragma unroll
    for (int k = 0; k < NC; ++k) logits[(b * NC + k) * HW + i] = acc[k];
  }
  if (amax) {
    int best = 0;
    float bv = acc[0];
#pragma unroll
    for (int k = 1; k < NC; ++k)
      if (acc[k] > bv) { bv = acc[k]; best = k; }
    amax[p] = (uint8_t)best;
  }
  if (stn_in) {
    float* o = stn_in + p * stn_cs;
    const float* f = frame + p * frame_cs;
    float vals[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) vals[k] = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) vals[k] = acc[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) vals[NC + k] = f[k];
    for (int c = 0; c < stn_cs && c < 16; c += 4)
      *reinterpret_cast<f32x4*>(o + c) = (f32x4){vals[c], vals[c + 1], vals[c + 2], vals[c + 3]};
  }
}

// ----------------------------------------------------------------- consistency CE
// per pixel: logsumexp(logits[:, p]) - logits[target, p]; block partial sums, then a
// second deterministic pass sums each frame's partials and divides by H*W.
constexpr int CE_BLOCK = 256;
constexpr int CE_PIX_PER_BLOCK = 2048;

__global__ __launch_bounds__(CE_BLOCK) void ce_partial_kernel(
    const float* __restrict__ logits, const int32_t* __restrict__ mask, int nc, int H, int W,
    int hm, int wm, int blocks_per_img, float* __restrict__ partial) {
  const int b = blockIdx.x / blocks_per_img, blk = blockIdx.x - b * blocks_per_img;
  const int HW = H * W;
  const float* lg = logits + (long)b * nc * HW;
  const int32_t* mk = mask + (long)b * hm * wm;
  float sum = 0.f;
  const int start = blk * CE_PIX_PER_BLOCK;
  for (int i = start + threadIdx.x; i < min(start + CE_PIX_PER_BLOCK, HW); i += CE_BLOCK) {
    int t;
    if (hm == H && wm == W) {
      t = mk[i];
    } else {  // F.interpolate(mode='nearest'): src = floor(dst * in/out)
      const int y = i / W, x = i - y * W;
      const int ys = min((int)floorf(y * ((float)hm / H)), hm - 1);
      const int xs = min((int)floorf(x * ((float)wm / W)), wm - 1);
      t = mk[ys * wm + xs];
    }
    float m = lg[i];
    for (int k = 1; k < nc; ++k) m = fmaxf(m, lg[(long)k * HW + i]);
    float se = 0.f, xt = 0.f;
    for (int k = 0; k < nc; ++k) {
      const float v = lg[(long)k * HW + i];
      se += expf(v - m);
      if (k == t) xt = v;
    }
    sum += (m + logf(se)) - xt;
  }
  __shared__ float red[CE_BLOCK / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < CE_BLOCK / 64; ++i) s += red[i];
    partial[blockIdx.x] = s;
  }
}

__global__ void ce_final_kernel(const float* __restrict__ partial, int blocks_per_img, float inv_hw,
                                float* __restrict__ score) {
  const int b = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < blocks_per_img; i += 64) s += partial[b * blocks_per_img + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
  if (threadIdx.x == 0) score[b] = s * inv_hw;
}

// ----------------------------------------------------------------- ResNet helpers
__global__ void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                    int C, int Ho, int Wo, long total4) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  const int c4n = C >> 2;
  const int c4 = idx % c4n;
  long r = idx / c4n;
  const int xo = r % Wo; r /= Wo;
  const int yo = r % Ho;
  const long b = r / Ho;
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = 2 * yo - 1 + dy;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int xx = 2 * xo - 1 + dx;
      if (xx < 0 || xx >= W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((b * H + yy) * W + xx) * C + 4 * c4);
#pragma unroll
      for (int j = 0; j < 4; ++j) m[j] = sfh_max_nan(m[j], v[j]);
    }
  }
  *reinterpret_cast<f32x4*>(y + idx * 4) = m;
}

// AdaptiveAvgPool2d(1) + Linear in two small launches: channel means with one thread per (frame, channel)
// (coalesced over channels, 64 channels per block so that a 16-frame batch fills 128 CUs instead of 16),
// then nout dot products per frame.
__global__ __launch_bounds__(256) void avgpool_mean_kernel(const float* __restrict__ x, int HW, int C,
                                                           float* __restrict__ mean) {
  // 64 channels x 4 pixel slices per block; four independent partial sums per thread keep loads in flight
  // (one dependent chain over the 240 pixels of the last ResNet stage took 58 us)
  __shared__ float part[4][64];
  const int b = blockIdx.y, cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float* xb = x + (long)b * HW * C + c;
    int i = sl;
    for (; i + 12 < HW; i += 16) {
      s0 += xb[(long)i * C];
      s1 += xb[(long)(i + 4) * C];
      s2 += xb[(long)(i + 8) * C];
      s3 += xb[(long)(i + 12) * C];
    }
    for (; i < HW; i += 4) s0 += xb[(long)i * C];
  }
  part[sl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && c < C) mean[(long)b * C + c] = ((part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl])) / (float)HW;
}

__global__ __launch_bounds__(256) void avgpool_linear_kernel(
    const float* __restrict__ mean, const float* __restrict__ w, const float* __restrict__ bias, int C, int nout,
    float* __restrict__ out) {
  const int b = blockIdx.x;
  const float* mb = mean + (long)b * C;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int o = wv; o < nout; o += 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += mb[c] * w[o * C + c];
#pragma unroll
    for (int k = 32; k > 0; k >>= 1) s += __shfl_down(s, k);
    if (lane == 0) out[b * nout + o] = s + bias[o];
  }
}

// ----------------------------------------------------------------- mask output formatting
// Class-id masks -> the uint8 images predict.py writes (predict.py:286-315): source is either
// logits NCHW (argmax = utils/postprocess.py:7-18), an int32 warp_mask or a uint8 id mask;
// nearest resize to (hd, wd) with OpenCV's INTER_NEAREST index rule (sx = min(floor(dx * (1/fx)),
// ws-1), fx = wd/ws in double), then mask_type gray (ids) / bin ((id>0)*255) / rgb (the colour
// table of utils/postprocess.py:21-61, 3 bytes per pixel).  One thread = 4 output pixels of a row.
struct MaskPalette {
  uint8_t c[8][3];
};

template <int KIND>  // 0: int32 ids, 1: uint8 ids, 2: fp32 logits NCHW
__device__ __forceinline__ int mask_id_at(const void* __restrict__ src, long b, int nc, int hs, int ws, int sy, int sx) {
  if (KIND == 0) return ((const int32_t*)src)[(b * hs + sy) * ws + sx];
  if (KIND == 1) return ((const uint8_t*)src)[(b * hs + sy) * ws + sx];
  const float* lg = (const float*)src + (b * nc * hs + sy) * (long)ws + sx;
  const long plane = (long)hs * ws;
  int best = 0;
  float bv = lg[0];
  for (int k = 1; k < nc; ++k) {
    const float v = lg[k * plane];
    if (v > bv) { bv = v; best = k; }
  }
  return best;
}

template <int KIND>
__global__ __launch_bounds__(256) void mask_format_kernel(const void* __restrict__ src, int nc, int hs, int ws,
                                                          int hd, int wd, double ify, double ifx, int mode,
                                                          MaskPalette pal, uint8_t* __restrict__ out, long total4) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total4) return;
  const int wq = (wd + 3) >> 2;
  const int xq = (int)(t % wq) * 4;
  const long r = t / wq;
  const int dy = (int)(r % hd);
  const long b = r / hd;
  const int sy = min((int)floor((double)dy * ify), hs - 1);
  const int nb = mode == 2 ? 3 : 1;
  uint8_t* o = out + ((b * hd + dy) * (long)wd + xq) * nb;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int dx = xq + j;
    if (dx >= wd) break;
    const int sx = min((int)floor((double)dx * ifx), ws - 1);
    const int id = mask_id_at<KIND>(src, b, nc, hs, ws, sy, sx);
    if (mode == 0) {
      o[j] = (uint8_t)id;
    } else if (mode == 1) {
      o[j] = id > 0 ? 255 : 0;
    } else {
      const int k = (id >= 0 && id < 8) ? id : 0;
      o[3 * j + 0] = pal.c[k][0];
      o[3 * j + 1] = pal.c[k][1];
      o[3 * j + 2] = pal.c[k][2];
    }
  }
}

}  // namespace

extern "C" int sfh_nchw_to_nhwc(const float* src, float* dst, int batch, int C, int H, int W, int cs,
                                void* stream) {
  SFH_REQUIRE(src && dst && batch > 0 && C > 0 && H > 0 && W > 0, "nchw_to_nhwc: bad argument");
  SFH_REQUIRE(cs >= C && cs % 4 == 0, "nchw_to_nhwc: cs=%d must be >= C=%d and a multiple of 4", cs, C);
  const long npix = (long)batch * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, C, H * W, cs, npix);
  return sfh_check_launch("nchw_to_nhwc_kernel");
}

extern "C" int sfh_u8hwc_to_f32nchw(const uint8_t* src, float* dst, int batch, int C, int H, int W, void* stream) {
  SFH_REQUIRE(src && dst && batch > 0 && C > 0 && H > 0 && W > 0, "u8hwc_to_f32nchw: bad argument");
  const long npix = (long)batch * H * W;
  hipLaunchKernelGGL(u8hwc_to_f32nchw_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, C, H * W, npix);
  return sfh_check_launch("u8hwc_to_f32nchw_kernel");
}

extern "C" int sfh_u8hwc_area2_to_f32nchw(const uint8_t* src, float* dst, int batch, int C, int H, int W,
                                          void* stream) {
  SFH_REQUIRE(src && dst && batch > 0 && C > 0 && C <= 4 && H > 0 && W > 0, "u8hwc_area2_to_f32nchw: bad argument");
  const long npix = (long)batch * H * W;
  hipLaunchKernelGGL(u8hwc_area2_to_f32nchw_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, C, H, W, npix);
  return sfh_check_launch("u8hwc_area2_to_f32nchw_kernel");
}

extern "C" int sfh_u8hwc_areak_to_f32nchw(const uint8_t* src, float* dst, int batch, int C, int H, int W, int k,
                                          void* stream) {
  SFH_REQUIRE(src && dst && batch > 0 && C > 0 && C <= 4 && H > 0 && W > 0 && k >= 2 && k <= 16,
              "u8hwc_areak_to_f32nchw: bad argument (k=%d must be 2 .. 16)", k);
  if (k == 2) return sfh_u8hwc_area2_to_f32nchw(src, dst, batch, C, H, W, stream);
  const long npix = (long)batch * H * W;
  hipLaunchKernelGGL(u8hwc_areak_to_f32nchw_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, C, H, W, k, k, 1.f / (float)(k * k), npix);
  return sfh_check_launch("u8hwc_areak_to_f32nchw_kernel");
}

extern "C" int sfh_u8hwc_areaxy_to_f32nchw(const uint8_t* src, float* dst, int batch, int C, int H, int W, int kx, int ky,
                                           void* stream) {
  SFH_REQUIRE(src && dst && batch > 0 && C > 0 && C <= 4 && H > 0 && W > 0 && kx >= 1 && kx <= 64 && ky >= 1 && ky <= 64 &&
                  kx * ky >= 2, "u8hwc_areaxy_to_f32nchw: bad argument (kx=%d, ky=%d: integer factors 1 .. 64, not both 1)", kx, ky);
  if (kx == ky && kx <= 16) return sfh_u8hwc_areak_to_f32nchw(src, dst, batch, C, H, W, kx, stream);
  const long npix = (long)batch * H * W;
  hipLaunchKernelGGL(u8hwc_areak_to_f32nchw_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, C, H, W, kx, ky, 1.f / (float)(kx * ky), npix);
  return sfh_check_launch("u8hwc_areak_to_f32nchw_kernel");
}

// computeResizeAreaTab of OpenCV's imgproc/resize.cpp, restated (host code): scale = 1 / (dsize / ssize) in double
extern "C" int sfh_resize_area_tab(int ssize, int dsize, int32_t* ofs, int32_t* si, float* alpha, int cap) {
  if (!ofs || !si || !alpha || ssize <= 0 || dsize <= 0 || dsize > ssize) return -1;
  const double scale = 1.0 / ((double)dsize / (double)ssize);
  int k = 0;
  for (int dx = 0; dx < dsize; ++dx) {
    ofs[dx] = k;
    const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    const double cell = scale < ssize - fsx1 ? scale : ssize - fsx1;
    int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
    if (sx2 > ssize - 1) sx2 = ssize - 1;
    if (sx1 > sx2) sx1 = sx2;
    if (sx1 - fsx1 > 1e-3) {
      if (k >= cap) return -1;
      si[k] = sx1 - 1;
      alpha[k++] = (float)((sx1 - fsx1) / cell);
    }
    for (int sx = sx1; sx < sx2; ++sx) {
      if (k >= cap) return -1;
      si[k] = sx;
      alpha[k++] = (float)(1.0 / cell);
    }
    if (fsx2 - sx2 > 1e-3) {
      if (k >= cap) return -1;
      double a = fsx2 - sx2;
      if (a > 1.0) a = 1.0;
      if (a > cell) a = cell;
      si[k] = sx2;
      alpha[k++] = (float)(a / cell);
    }
  }
  ofs[dsize] = k;
  return k;
}

extern "C" int sfh_u8hwc_area_to_f32nchw(const uint8_t* src, float* dst, int batch, int C, int Hs, int Ws, int Hd, int Wd,
                                         const int32_t* xofs, const int32_t* xsi, const float* xalpha, const int32_t* yofs,
                                         const int32_t* ysi, const float* ybeta, void* stream) {
  SFH_REQUIRE(src && dst && xofs && xsi && xalpha && yofs && ysi && ybeta, "u8hwc_area_to_f32nchw: null pointer");
  SFH_REQUIRE(batch > 0 && C > 0 && C <= 4 && Hd > 0 && Wd > 0 && Hs >= Hd && Ws >= Wd,
              "u8hwc_area_to_f32nchw: bad geometry %dx%d -> %dx%d (a downscale, at most 4 channels)", Ws, Hs, Wd, Hd);
  const long npix = (long)batch * Hd * Wd;
  hipLaunchKernelGGL(u8hwc_area_to_f32nchw_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     src, dst, C, Hs, Ws, Hd, Wd, xofs, xsi, xalpha, yofs, ysi, ybeta, npix);
  return sfh_check_launch("u8hwc_area_to_f32nchw_kernel");
}

extern "C" int sfh_nhwc_to_nchw(const float* src, float* dst, int batch, int C, int H, int W, int cs,
                                void* stream) {
  SFH_REQUIRE(src && dst && batch > 0 && C > 0 && H > 0 && W > 0, "nhwc_to_nchw: bad argument");
  SFH_REQUIRE(cs >= C && cs % 4 == 0, "nhwc_to_nchw: cs=%d must be >= C=%d and a multiple of 4", cs, C);
  const long npix = (long)batch * H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, C, H * W, cs, npix);
  return sfh_check_launch("nhwc_to_nchw_kernel");
}

extern "C" int sfh_space_to_depth2(const float* src, float* dst, int batch, int H, int W, int cs,
                                   void* stream) {
  SFH_REQUIRE(src && dst && batch > 0 && H > 0 && W > 0 && cs > 0 && cs % 4 == 0, "space_to_depth2: bad argument");
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const long total4 = (long)batch * H2 * W2 * cs;
  hipLaunchKernelGGL(space_to_depth2_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, H, W, cs, H2, W2, total4);
  return sfh_check_launch("space_to_depth2_kernel");
}

extern "C" int sfh_resize_nchw(const float* src, float* dst, int64_t planes, int hs, int ws, int hd, int wd,
                               int mode, int align_corners, void* stream) {
  SFH_REQUIRE(src && dst && planes > 0 && hs > 0 && ws > 0 && hd > 0 && wd > 0, "resize_nchw: bad argument");
  SFH_REQUIRE(mode == 0 || mode == 1, "resize_nchw: mode %d (0 nearest, 1 bilinear)", mode);
  const long total = planes * hd * wd;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (mode == 1)
    hipLaunchKernelGGL(resize_bilinear_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, hs, ws,
                       hd, wd, align_corners, total);
  else
    hipLaunchKernelGGL(resize_nearest_nchw_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, hs, ws, hd,
                       wd, total);
  return sfh_check_launch("resize_nchw_kernel");
}

extern "C" int sfh_upsample2x_bilinear_nhwc(const float* src, float* dst, int batch, int H, int W, int C,
                                            void* stream) {
  SFH_REQUIRE(src && dst && batch > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x: bad argument");
  const long total4 = (long)batch * 2 * H * 2 * W * (C / 4);
  hipLaunchKernelGGL(upsample2x_nhwc_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, H, W, C, total4);
  return sfh_check_launch("upsample2x_nhwc_kernel");
}

extern "C" int sfh_fold_bn(const float* conv_bias, const float* gamma, const float* beta,
                           const float* mean, const float* var, float eps, int n, int repeat,
                           float* scale, float* shift, void* stream) {
  SFH_REQUIRE(scale && shift && n > 0 && repeat > 0, "fold_bn: bad argument");
  SFH_REQUIRE(!gamma || (beta && mean && var), "fold_bn: gamma given without beta/mean/var");
  hipLaunchKernelGGL(fold_bn_kernel, dim3((unsigned)((n * repeat + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, conv_bias, gamma, beta, mean, var, eps, n, repeat, scale, shift);
  return sfh_check_launch("fold_bn_kernel");
}

extern "C" int sfh_outconv_fwd(const float* x, int cin, const float* w, const float* bias, int nc,
                               int batch, int H, int W, float* logits_nchw, uint8_t* argmax_u8,
                               float* stn_in, int stn_cs, const float* frame_nhwc, int frame_cs,
                               void* stream) {
  SFH_REQUIRE(x && w && bias && batch > 0 && H > 0 && W > 0, "outconv: bad argument");
  SFH_REQUIRE(cin % 8 == 0 && cin <= 1024, "outconv: cin=%d must be a multiple of 8", cin);
  SFH_REQUIRE(nc >= 1 && nc <= 8, "outconv: nc=%d unsupported (1..8)", nc);
  SFH_REQUIRE(!stn_in || (frame_nhwc && stn_cs % 4 == 0 && stn_cs >= nc + 3 && stn_cs <= 16 && frame_cs >= 3),
              "outconv: bad stn_in geometry");
  const long npix = (long)batch * H * W;
  const unsigned grid = (unsigned)((npix + 127) / 128);
  const size_t lds = (size_t)nc * cin * sizeof(float);
#define SFH_OC(N)                                                                              \
  case N:                                                                                      \
    hipLaunchKernelGGL(outconv_kernel<N>, dim3(grid), dim3(256), lds, (hipStream_t)stream, x, cin, w, \
                       bias, npix, H * W, logits_nchw, argmax_u8, stn_in, stn_cs, frame_nhwc,  \
                       frame_cs);                                                              \
    break;
  switch (nc) {
    SFH_OC(1) SFH_OC(2) SFH_OC(3) SFH_OC(4) SFH_OC(5) SFH_OC(6) SFH_OC(7) SFH_OC(8)
  }
#undef SFH_OC
  return sfh_check_launch("outconv_kernel");
}

extern "C" int64_t sfh_ce_workspace_floats(int batch, int H, int W) {
  if (batch <= 0 || H <= 0 || W <= 0) return -1;
  return (int64_t)batch * sfh_cdiv(H * W, CE_PIX_PER_BLOCK);
}

extern "C" int sfh_consistency_ce_fwd(const float* logits, const int32_t* mask, int batch, int nc,
                                      int H, int W, int hm, int wm, float* partial, float* score,
                                      void* stream) {
  SFH_REQUIRE(logits && mask && partial && score, "consistency_ce: null pointer");
  SFH_REQUIRE(batch > 0 && nc > 0 && H > 0 && W > 0 && hm > 0 && wm > 0, "consistency_ce: bad geometry");
  const int bpi = sfh_cdiv(H * W, CE_PIX_PER_BLOCK);
  hipLaunchKernelGGL(ce_partial_kernel, dim3((unsigned)(batch * bpi)), dim3(CE_BLOCK), 0,
                     (hipStream_t)stream, logits, mask, nc, H, W, hm, wm, bpi, partial);
  int rc = sfh_check_launch("ce_partial_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(ce_final_kernel, dim3((unsigned)batch), dim3(64), 0, (hipStream_t)stream, partial,
                     bpi, 1.0f / (float)(H * W), score);
  return sfh_check_launch("ce_final_kernel");
}

extern "C" int sfh_maxpool3x3s2_fwd(const float* x, float* y, int batch, int H, int W, int C,
                                    void* stream) {
  SFH_REQUIRE(x && y && batch > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool3x3s2: bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total4 = (long)batch * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, y, H, W, C, Ho, Wo, total4);
  return sfh_check_launch("maxpool3x3s2_kernel");
}

extern "C" int sfh_avgpool_linear_fwd(const float* x, const float* w, const float* bias, int batch,
                                      int H, int W, int C, int nout, float* feat, float* out, void* stream) {
  SFH_REQUIRE(x && w && bias && feat && out && batch > 0 && batch <= 65535 && H > 0 && W > 0 && C > 0 && nout > 0,
              "avgpool_linear: bad argument");
  hipLaunchKernelGGL(avgpool_mean_kernel, dim3((unsigned)sfh_cdiv(C, 64), (unsigned)batch), dim3(256), 0,
                     (hipStream_t)stream, x, H * W, C, feat);
  int rc = sfh_check_launch("avgpool_mean_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(avgpool_linear_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, feat, w, bias, C,
                     nout, out);
  return sfh_check_launch("avgpool_linear_kernel");
}

extern "C" int sfh_mask_format_fwd(const void* src, int src_kind, int nc, int batch, int hs, int ws,
                                   int hd, int wd, int mode, const uint8_t* palette, uint8_t* out,
                                   void* stream) {
  SFH_REQUIRE(src && out && batch > 0 && hs > 0 && ws > 0 && hd > 0 && wd > 0, "mask_format: bad argument");
  SFH_REQUIRE(src_kind >= 0 && src_kind <= 2, "mask_format: src_kind %d (0 int32 ids, 1 uint8 ids, 2 logits)", src_kind);
  SFH_REQUIRE(mode >= 0 && mode <= 2, "mask_format: mode %d (0 gray, 1 bin, 2 rgb)", mode);
  SFH_REQUIRE(src_kind != 2 || nc >= 2, "mask_format: logits need nc >= 2");
  SFH_REQUIRE(mode != 2 || palette, "mask_format: rgb mode needs a palette of 8x3 bytes");
  MaskPalette pal = {};
  if (palette)
    for (int k = 0; k < 8; ++k)
      for (int c = 0; c < 3; ++c) pal.c[k][c] = palette[k * 3 + c];  // host pointer, copied by value
  const double ify = 1.0 / ((double)hd / (double)hs), ifx = 1.0 / ((double)wd / (double)ws);
  const long total4 = (long)batch * hd * ((wd + 3) / 4);
  const dim3 grid((unsigned)((total4 + 255) / 256));
#define SFH_MF(K)                                                                                   \
  hipLaunchKernelGGL(mask_format_kernel<K>, grid, dim3(256), 0, (hipStream_t)stream, src, nc, hs, ws, hd, \
                     wd, ify, ifx, mode, pal, out, total4)
  if (src_kind == 0) SFH_MF(0);
  else if (src_kind == 1) SFH_MF(1);
  else SFH_MF(2);
#undef SFH_MF
  return sfh_check_launch("mask_format_kernel");
}

extern "C" int sfh_upsample2x_bilinear_nhwc_bwd(const float* dy, float* dx, int batch, int H, int W, int C,
                                                void* stream) {
  SFH_REQUIRE(dy && dx && batch > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x_bwd: bad argument");
  const long total4 = (long)batch * H * W * (C / 4);
  hipLaunchKernelGGL(upsample2x_nhwc_bwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, dx, H, W, C, total4);
  return sfh_check_launch("upsample2x_nhwc_bwd_kernel");
}

extern "C" int sfh_resize_nearest_nchw_bwd(const float* dy, float* dx, int64_t planes, int hs, int ws, int hd,
                                           int wd, void* stream) {
  SFH_REQUIRE(dy && dx && planes > 0 && hs > 0 && ws > 0 && hd > 0 && wd > 0, "resize_nearest_bwd: bad argument");
  const long total = planes * hs * ws;
  hipLaunchKernelGGL(resize_nearest_nchw_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, dx, hs, ws, hd, wd, total);
  return sfh_check_launch("resize_nearest_nchw_bwd_kernel");
}

extern "C" int sfh_compose_up_weights(const float* wconv, int cout, int c0, int c1, const float* wt, int cx,
                                      const float* bt, const float* scale4, const float* shift4, float* w2,
                                      float* shift_border, void* stream) {
  SFH_REQUIRE(wconv && wt && bt && scale4 && shift4 && w2 && shift_border && cout > 0 && c0 >= 0 && c1 > 0 && cx > 0,
              "compose_up_weights: bad argument");
  SFH_REQUIRE(cout % 16 == 0 && cx % 16 == 0, "compose_up_weights: cout and cx must be multiples of 16");
  hipLaunchKernelGGL(compose_up_weights_kernel, dim3((unsigned)(cx / 16), (unsigned)(cout / 16), 4), dim3(256), 0,
                     (hipStream_t)stream, wconv, cout, c0, c1, wt, cx, w2);
  int rc = sfh_check_launch("compose_up_weights_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(compose_up_bias_kernel, dim3((unsigned)((64 * cout + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     wconv, cout, c0, c1, bt, scale4, shift4, shift_border);
  return sfh_check_launch("compose_up_bias_kernel");
}
