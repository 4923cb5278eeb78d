// hostprep.hip - the small kernels an engine needs while it is BUILT (weight packing, BatchNorm folding) or between
// launches, so that no stock torch kernel runs on the predict path (round 6): one batched |x| min / max reduction over all
// weights of a model (the exponent of every two-plane fp16 weight tensor comes from it), element-wise helpers on the folded
// scale / shift vectors, a pitched copy (channel slices of OIHW weights) and the assembly of the ResNet-STN input for the
// input modes the fused OutConv epilogue does not cover (models/reconstructor.py:174-183,214).
#include "common.h"

namespace {

// grid = (slices, tensors): slice s of tensor t covers elements [s, s + 1) * ceil(n / slices) rounded to 16-byte quads; |x| bit
// patterns are ordered like unsigned integers, NaN above Inf.  16-byte loads where the tensor's address allows.
__global__ __launch_bounds__(256) void multi_absminmax_kernel(const long* __restrict__ table, uint32_t* __restrict__ words) {
  const int t = blockIdx.y;
  const float* p = reinterpret_cast<const float*>(table[2 * t]);
  const long n = table[2 * t + 1];
  uint32_t mx = 0u, mn = 0x7FFFFFFFu;
  auto take = [&](float v) {
    const uint32_t b = __float_as_uint(v) & 0x7FFFFFFFu;
    mx = b > mx ? b : mx;
    mn = b < mn ? b : mn;
  };
  // small tensors are finished by the first few workgroups of their row; the others have no element and leave at once
  const bool quads = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  if (blockIdx.x != 0 && (long)blockIdx.x * 256 >= (quads ? (n >> 2) : n)) return;
  if (quads) {
    const long nq = n >> 2;
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long)gridDim.x * 256) {
      const f32x4 v = q[i];
      take(v[0]); take(v[1]); take(v[2]); take(v[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) take(p[(nq << 2) + threadIdx.x]);
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) take(p[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t a = __shfl_xor(mx, o), c = __shfl_xor(mn, o);
    mx = a > mx ? a : mx;
    mn = c < mn ? c : mn;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&words[2 * t], mx);
    atomicMax(&words[2 * t + 1], 0x7FFFFFFFu - mn);      // stored inverted: a zero-filled table is the identity of both
  }
}

template <int OP>
__global__ __launch_bounds__(256) void vec_op_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, int nb,
                                                     float factor, float* __restrict__ dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = a[i];
  if (OP == 1) v = v / b[i % nb];
  if (OP == 2) v = v * b[i % nb];
  if (OP != 1 || factor != 1.f) v = v * factor;
  dst[i] = v;
}

__global__ __launch_bounds__(256) void copy2d_kernel(const uint32_t* __restrict__ src, long src_pitch, uint32_t* __restrict__ dst,
                                                     long dst_pitch, int width, long rows) {
  const long r = blockIdx.y;
  for (int x = blockIdx.x * 256 + threadIdx.x; x < width; x += gridDim.x * 256) dst[r * dst_pitch + x] = src[r * src_pitch + x];
}

// dst (B,H,W,cs) NHWC = cat((logits, frame, uv), channel) zero-padded to cs; every source NCHW, any of them absent
__global__ __launch_bounds__(256) void stn_input_assemble_kernel(const float* __restrict__ logits, int nc,
                                                                 const float* __restrict__ frame, int cf,
                                                                 const float* __restrict__ uv, int cu, long HW, long npix,
                                                                 int cs, float* __restrict__ dst) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const long b = p / HW, i = p - b * HW;
  float* o = dst + p * cs;
  int c = 0;
  for (int k = 0; k < nc; ++k) o[c++] = logits[(b * nc + k) * HW + i];
  for (int k = 0; k < cf; ++k) o[c++] = frame[(b * cf + k) * HW + i];
  for (int k = 0; k < cu; ++k) o[c++] = uv[(b * cu + k) * HW + i];
  for (; c < cs; ++c) o[c] = 0.f;
}

// flag |= 1 if any of rows 1 .. rows-1 differs from row 0 (bit patterns)
__global__ __launch_bounds__(256) void rows_differ_kernel(const uint32_t* __restrict__ x, long row_words, long total,
                                                          uint32_t* __restrict__ flag) {
  bool d = false;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256)
    d |= x[row_words + i] != x[i % row_words];
  if (__any(d) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

__global__ __launch_bounds__(256) void fill_words_kernel(uint32_t* __restrict__ dst, long n, uint32_t v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = v;
}

}  // namespace

extern "C" int sfh_fill_words(void* dst, int64_t n, uint32_t value, void* stream) {
  SFH_REQUIRE(dst && n > 0, "fill_words: bad argument");
  long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint32_t*>(dst),
                     (long)n, value);
  return sfh_check_launch("fill_words_kernel");
}

extern "C" int sfh_rows_differ(const void* x, int64_t row_words, int rows, uint32_t* flag, void* stream) {
  SFH_REQUIRE(x && flag && row_words > 0 && rows >= 1, "rows_differ: bad argument");
  if (rows == 1) return SFH_OK;
  const long total = (long)row_words * (rows - 1);
  long nb = (total + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(rows_differ_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint32_t*>(x), (long)row_words, total, flag);
  return sfh_check_launch("rows_differ_kernel");
}

extern "C" int sfh_multi_absminmax(const void* table, int ntensors, uint32_t* words, void* stream) {
  SFH_REQUIRE(table && words && ntensors > 0 && ntensors <= 65535, "multi_absminmax: bad argument");
  hipLaunchKernelGGL(multi_absminmax_kernel, dim3(512, (unsigned)ntensors), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long*>(table), words);
  return sfh_check_launch("multi_absminmax_kernel");
}

extern "C" int sfh_vec_op(int op, const float* a, const float* b, int64_t n, int nb, float factor, float* dst, void* stream) {
  SFH_REQUIRE(a && dst && n > 0 && op >= 0 && op <= 2, "vec_op: bad argument");
  SFH_REQUIRE(op == 0 || (b && nb > 0), "vec_op: op %d needs a second operand", op);
  const dim3 grid((unsigned)((n + 255) / 256));
  switch (op) {
    case 0: hipLaunchKernelGGL(vec_op_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a, b, (long)n, nb, factor, dst); break;
    case 1: hipLaunchKernelGGL(vec_op_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a, b, (long)n, nb, factor, dst); break;
    default: hipLaunchKernelGGL(vec_op_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a, b, (long)n, nb, factor, dst); break;
  }
  return sfh_check_launch("vec_op_kernel");
}

extern "C" int sfh_copy2d_words(const void* src, int64_t src_pitch, void* dst, int64_t dst_pitch, int width, int64_t rows,
                                void* stream) {
  SFH_REQUIRE(src && dst && width > 0 && rows > 0 && rows <= 65535 && src_pitch >= width && dst_pitch >= width,
              "copy2d_words: bad argument");
  unsigned gx = (unsigned)((width + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(copy2d_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint32_t*>(src), (long)src_pitch, reinterpret_cast<uint32_t*>(dst), (long)dst_pitch,
                     width, (long)rows);
  return sfh_check_launch("copy2d_kernel");
}

extern "C" int sfh_stn_input_assemble(const float* logits, int nc, const float* frame, int cf, const float* uv, int cu,
                                      int batch, int H, int W, int cs, float* dst, void* stream) {
  SFH_REQUIRE(dst && batch > 0 && H > 0 && W > 0 && nc >= 0 && cf >= 0 && cu >= 0 && nc + cf + cu > 0 && nc + cf + cu <= cs,
              "stn_input_assemble: %d + %d + %d channels into %d", nc, cf, cu, cs);
  SFH_REQUIRE((nc == 0 || logits) && (cf == 0 || frame) && (cu == 0 || uv), "stn_input_assemble: null source");
  const long npix = (long)batch * H * W;
  hipLaunchKernelGGL(stn_input_assemble_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits,
                     nc, frame, cf, uv, cu, (long)H * W, npix, cs, dst);
  return sfh_check_launch("stn_input_assemble_kernel");
}
