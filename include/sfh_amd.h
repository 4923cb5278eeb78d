/*
 * sfh_amd.h - C ABI of libsfh_amd.so: the MI355X (gfx950) hot path of
 * darkAlert/sports-field-homography.
 *
 * The reference has no FFI layer: its operator boundary is the Python class
 * models.Reconstructor (models/reconstructor.py:30-247), which delegates every
 * device operation to ATen/cuDNN and Kornia.  Each entry point below replaces the
 * ATen/Kornia call(s) named in its comment (paths relative to the reference root).
 * The Python mirror of the class (sports-field-homography_amd/reconstructor.py)
 * binds these symbols with ctypes; INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - all pointers are DEVICE pointers (hipMalloc'd / torch CUDA tensors) unless named
 *    host_*; activations are fp32 NHWC ("channels last"), checkpoint tensors keep the
 *    reference layouts (OIHW conv weights, IOHW transposed-conv weights);
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*), allocates
 *    nothing, keeps no pointer and no global mutable state;
 *  - return value: 0 = success, negative = SFH_E_* (message via sfh_last_error()).
 */
#ifndef SFH_AMD_H
#define SFH_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFH_OK 0
#define SFH_E_ARG (-1)     /* invalid argument / unsupported shape */
#define SFH_E_LAUNCH (-2)  /* HIP launch error */

/* Tile shapes of the implicit-GEMM conv kernel (output pixels per workgroup). */
#define SFH_TILE_8x32 0   /* 8 rows x 32 cols, MFMA pixel groups of 1x16 */
#define SFH_TILE_16x16 1  /* 16 rows x 16 cols, 1x16 groups              */
#define SFH_TILE_32x8 2   /* 32 rows x 8 cols, 2x8 groups (narrow maps)  */
#define SFH_TILE_8x16 3   /* 8 rows x 16 cols, 1x16 groups (split-bf16 kernel: small maps, stride 2) */
#define SFH_TILE_16x8 4   /* 16 rows x 8 cols, 2x8 groups  (split-bf16 kernel: small maps, stride 2) */

/* Tensor formats of conv sources / destinations.
 * F32: fp32 NHWC (B,H,W,cs).
 * S3 : "split-3" bf16 (B, H, cs/32, 3 planes, 4 groups, W, 8): v = plane0 + plane1 + plane2 exactly,
 *      plane0 = bf16(v), plane1 = bf16(v - plane0), plane2 = bf16(v - plane0 - plane1); channel c of
 *      pixel (y, x) lives in block c/32, group (c%32)/8, lane c%8 - each (block, plane, group) of an
 *      image row is W x 16 contiguous bytes.
 * H2 : "split-2" fp16 (B, H, cs/32, 2 planes, 4 groups, W, 8), the same layout with two planes:
 *      u = v * 2^e saturated to +-65504, plane0 = f16(u), plane1 = f16(u - plane0) (both RNE):
 *      22 significand bits of v while plane1 is a normal fp16 number (|u| >= 2^-3), an absolute error
 *      <= 2^-25 * 2^-e below that (fp16 subnormals are honoured by the conversions and by the MFMA).  The
 *      exponent e belongs to the TENSOR and is chosen by the caller (sfh_conv_desc.h2_exp_dst of the producer =
 *      h2_exp_src / h2_exp_res of its consumers); |v| must stay below 65504 * 2^-e: the producing kernel
 *      saturates, raises sfh_conv_desc.h2_overflow and records the largest |u| it saw in sfh_conv_desc.h2_range,
 *      from which the caller picks a smaller exponent.  The exponent trades range against the absolute error
 *      floor; end to end the result does not depend on it over a span of 2^6 (tests/probes/f16x3_error_probe.py).
 *      SFH_H2_ACT_EXP (2: |v| < 16376) is the conventional default - 4x the largest activation the synthetic
 *      ResNet-STN checkpoints of the tests produce - and what the training kernels use throughout. */
#define SFH_FMT_F32 0
#define SFH_FMT_S3 1
#define SFH_FMT_H2 2
#define SFH_FMT_FH2 3 /* the frame tensor of sfh_frame_to_h2 (16 bytes per pixel: two fp16 planes of channels 0..3); the only
                         source format sfh_conv3x3_c4h2_fwd takes, and only that entry takes it */
#define SFH_H2_ACT_EXP 2

/* Output modes of sfh_conv_fwd. */
#define SFH_OUT_NHWC 0        /* dst[b][y][x][co]                                        */
#define SFH_OUT_UPSCATTER2 1  /* transposed conv k2 s2: virtual cout = (dy*2+dx)*Cout+co; with ksize 2
                                 (sfh_conv_s3_fwd only): the 2x2 window of quadrant (dy,dx) starts at
                                 (y + dy - 1, x + dx - 1) - the composed ConvTranspose2d + conv3x3 of
                                 sfh_compose_up_weights */

typedef struct sfh_conv_desc {
  /* source 0: channels [0, c0) of the conv input; physical NHWC tensor (B, h0, w0, cs0). */
  const float* src0;
  int32_t c0, cs0, h0, w0;
  int32_t pool0;  /* 1: 2x2/stride-2 max-pool applied on load (h0 >= 2*H, w0 >= 2*W) */
  /* source 1 (optional, NULL if absent): channels [c0, c0+c1); physical (B, h1, w1, cs1),
   * placed at (pad_top1, pad_left1) inside the H x W input frame, zero elsewhere.       */
  const float* src1;
  int32_t c1, cs1, h1, w1, pad_top1, pad_left1;
  /* geometry */
  int32_t batch, H, W;  /* conv input frame */
  int32_t ksize;        /* 3 (pad 1), 1 (pad 0) or 4 (pad 2 before, 1 after; stem) */
  int32_t stride;       /* 1 or 2 */
  int32_t tile;         /* SFH_TILE_* */
  /* weights packed by sfh_pack_conv_weights for this (ksize, c0, c1); per-channel epilogue */
  const float* wpacked;
  const float* scale;  /* [cout_virtual] */
  const float* shift;  /* [cout_virtual] */
  int32_t cout;        /* virtual output channels, multiple of 64 */
  int32_t relu;
  /* optional residual added before ReLU: NHWC tensor with the geometry of dst */
  const float* residual;
  /* destination */
  float* dst;
  int32_t dst_cs;   /* channel stride (elements per pixel and plane) of dst */
  int32_t out_mode; /* SFH_OUT_* */
  /* formats (SFH_FMT_*): both sources share src_fmt; residual and dst_pool share dst_fmt */
  int32_t src_fmt, dst_fmt;
  /* optional second output: MaxPool2d(2) (floor) of dst, (B, Ho/2, Wo/2, pool_cs) */
  float* dst_pool;
  int32_t pool_cs;
  /* 1: `residual` is fp32 NHWC (channel stride dst_cs) even though dst is S3 (sfh_conv_s3_fwd) */
  int32_t residual_f32;
  /* optional [16][cout] shift table replacing `shift`, indexed by the border class of the OUTPUT pixel
   * (4*(row class) + (col class), see sfh_compose_up_weights): the up-sampling fusion below needs it */
  const float* shift_border;
  /* 2x2 up-scatter conv only: size of the destination frame when the up-sampled tensor is one row /
   * column short of it (F.pad in Up, unet/unet_parts.py:59-63, with diff 1: pad 0 before, 1 after);
   * 0 = 2*Ho x 2*Wo.  The conv frame (H, W) is then source rows/cols + 1, the extra source row reads 0. */
  int32_t up_dst_h, up_dst_w;
  /* 1: walk the pixel tiles from the last to the first (sfh_conv_s3_fwd).  Alternating the direction from
   * one layer to the next makes a layer start on the part of its input that the previous launch wrote
   * last, i.e. the part still resident in L2 / Infinity Cache. */
  int32_t reverse_tiles;
  /* OutConv fused behind the last 3x3 conv (sfh_conv_s3_fwd, cout == 64, plain output): with head_w set the
   * launch also computes logits = head_w (head_nc x 64) . y + head_b for its pixels and writes them NCHW to
   * head_logits (B,head_nc,H,W) and, if head_stn is set, cat((logits, frame)) zero-padded to 8 channels to
   * head_stn (B,H,W,8) (frame: head_frame, fp32 NHWC with 4 stored channels) - what sfh_outconv_fwd does in
   * a second pass over y.  head_skip_dst != 0: y itself is not stored. */
  const float* head_w;
  const float* head_b;
  int32_t head_nc, head_skip_dst;
  float* head_logits;
  float* head_stn;
  const float* head_frame;
  /* H2 destinations: device word that is OR-ed with 1 when a value had to be saturated to the fp16 range
   * (optional).  The caller zeroes it and reads it back after the last launch of a forward pass. */
  uint32_t* h2_overflow;
  /* Exponents of the H2 tensors of this launch (see SFH_FMT_H2; every H2 tensor stores v * 2^e):
   * h2_exp_src - both sources (folded by the CALLER into `scale` together with the weight exponent; the
   *              kernels that split an fp32 source themselves, sfh_stem7x7_fwd, read it);
   * h2_exp_dst - dst and dst_pool;  h2_exp_res - an H2 residual.  Range -64 .. 64. */
  int32_t h2_exp_src, h2_exp_dst, h2_exp_res;
  /* H2 destinations (optional): device word that receives, by atomic max, the largest bit pattern of
   * |v * 2^h2_exp_dst| this launch produced BEFORE saturation (bit patterns of non-negative floats order like
   * the floats; an Inf / NaN reads >= 0x7F800000).  > 0x477FE000 (65504.f) means the tensor was saturated; the
   * value tells the caller which exponent would have fitted.  Never reset by the library. */
  uint32_t* h2_range;
  /* sfh_conv_s3_fwd, H2 sources: couts per workgroup. 0 = chosen by the launcher, 64 = the 4-wave workgroup
   * (256 pixels x 64 couts, two per CU), 128 = the 8-wave workgroup (256 pixels x 128 couts, one per CU, two LDS
   * buffers; needs stride 1, ksize 3 or 2, cout % 128 == 0 - per quadrant for the up-scatter conv -, at least 128
   * input channels, no fused head). */
  int32_t wg_couts;
  /* kernels that split an fp32 source themselves (sfh_stem7x7_fwd): 0 or SFH_FMT_S3 = three bf16 planes, six products;
   * SFH_FMT_H2 = two fp16 planes, three products (weights from sfh_pack_stem_weights with the same format). */
  int32_t split_arith;
  /* split-K (sfh_conv_s3_fwd): ksplit > 1 launches ksplit copies of the grid; copy k accumulates its share of the
   * 32-channel stages of the K loop and writes acc * scale + shift as fp32 NHWC into slab k of dst, slabs
   * ksplit_stride BYTES apart - for layers whose (tile, cout block) grid alone leaves most of the chip idle (ResNet
   * layer3 / layer4 at batch 16).  Needs a single source, dst_fmt F32, no ReLU / residual / dst_pool / head; pass an
   * all-zero `shift` and let sfh_splitk_finish add the slabs, the real shift, the residual and the activation. */
  int32_t ksplit;
  int64_t ksplit_stride;
  /* sfh_conv_s3_fwd, 3x3 stride 1, plain output (optional): fp32 NHWC tensor (B, H, W, cout) the accumulators START from,
   * in accumulator units - i.e. a residual r enters as r / scale[c], which the caller arranges in the producer of that
   * tensor (the fused Up block: the composed 2x2 conv writes its partial divided by the skip-half conv's scale).  Loaded
   * in the prologue instead of sixteen dependent reads in the epilogue. */
  const float* acc_init;
  /* sfh_conv_s3_fwd, H2 sources, 3x3 or 1x1 stride 1, plain fp32 destination without ReLU / residual (the raw z = conv + bias of
   * a training-mode layer; optional): batch-statistics BatchNorm sums from the epilogue.  stats_partial = float64 table
   * [stats_rows][2][cout], zero before the launch; every wave ADDS (fp64 sums, fp64 atomics) the sums of z and of z * z
   * over its in-frame pixels into row (wave's tile slot) % stats_rows.  sfh_bn_stats_partials then adds the rows up
   * - the separate statistics pass over z (sfh_bn_stats) is not needed.  stats_rows: a power of two, 64 .. 65536. */
  double* stats_partial;
  int32_t stats_rows;
  /* backward mode of the same table (optional, with stats_partial): this launch is the backward-data conv whose fp32 output
   * dy is the ONLY gradient of a BatchNorm (+ReLU) layer; bwd_z = that layer's pre-BatchNorm tensor (fp32 NHWC, the shape of
   * dst, dst_cs == cout), bwd_mi = its mean | invstd (2 * cout floats), bwd_gamma / bwd_beta = its affine parameters (both
   * NULL: the layer has no ReLU).  The table then receives sum g and sum g * xhat with g = dy * (y > 0),
   * y = (z - mean) * invstd * gamma + beta exactly as sfh_bn_apply computes it: sfh_bn_bwd_reduce is not needed. */
  const float* bwd_z;
  const float* bwd_mi;
  const float* bwd_gamma;
  const float* bwd_beta;
  /* sfh_conv_upfused_fwd only: the packed weights (sfh_pack_h2_weights, ksize 2, 4 * cout virtual couts) and the 4 * cout
   * epilogue scales of the composed 2x2 conv of a fused Up block (its border-class shifts ride in shift_border). */
  const float* up_wpacked;
  const float* up_scale;
} sfh_conv_desc;

const char* sfh_last_error(void);
int sfh_version(void);

/* conv3x3+BN+ReLU, 1x1 conv, ConvTranspose2d k2 s2 - fp32 MFMA implicit GEMM.
 * Replaces nn.Conv2d/BatchNorm2d/ReLU of DoubleConv (unet/unet_parts.py:14-21), the
 * MaxPool2d of Down (unet/unet_parts.py:33, via pool0), F.pad + torch.cat of Up
 * (unet/unet_parts.py:59-67, via src1), nn.ConvTranspose2d of Up (unet/unet_parts.py:52),
 * and the conv/BN/ReLU/residual of BasicBlock (models/resnet.py:64-82).                  */
int sfh_conv_fwd(const sfh_conv_desc* d, void* stream);

/* fp32-accurate convolution on the bf16 matrix cores: sources in S3 format, weights packed by
 * sfh_pack_s3_weights, six bf16 MFMAs per 32 k (w0x2 + w1x1 + w2x0 + w0x1 + w1x0 + w0x0) with fp32
 * accumulation.  Same descriptor and epilogue as sfh_conv_fwd; ksize 1, 3 (stride 1 or 2) or 4 (stem),
 * c0/c1 multiples of 32, no pool0 (use the producer's dst_pool).  Replaces the same reference calls. */
int sfh_conv_s3_fwd(const sfh_conv_desc* d, void* stream);
int64_t sfh_packed_s3_weight_bytes(int ksize, int c0, int c1, int cout_virtual);
/* mode 0: OIHW conv weight; mode 1: IOHW ConvTranspose2d weight (ksize 1, cout_virtual = 4*cout);
 * mode 2: the 7x7 s2 stem as a 4x4 conv over the space-to-depth input (ksize 4, aux = real cin),
 * mode 3 / 4: backward-data of Conv2d / ConvTranspose2d; all as in sfh_pack_conv_weights. */
int sfh_pack_s3_weights(const float* w, void* packed, int ksize, int c0, int c1, int cout_virtual,
                        int mode, int aux, void* stream);
/* fp32 NHWC (rows = B*H, W, cs) <-> S3 (rows, cs/32, 3, 4, W, 8) conversion */
int sfh_f32_to_s3(const float* src, void* dst, int64_t rows, int W, int cs, void* stream);
int sfh_s3_to_f32(const void* src, float* dst, int64_t rows, int W, int cs, void* stream);

/* The same convolution with TWO fp16 planes per operand ("f16x3": sources in H2 format, src_fmt = SFH_FMT_H2 in
 * the descriptor of sfh_conv_s3_fwd): three fp16 MFMAs per 32 k (w0x1 + w1x0 + w0x0, fp32 accumulation) instead
 * of six bf16 ones.  Operands carry 22 significand bits; the dropped product is <= 2^-22 of the full one.  Measured
 * end to end (tests/probes/f16x3_error_probe.py) the representation error is below half of what the fp32
 * accumulation order already costs.  Weights are packed as planes of w * 2^wexp (wexp chosen by the caller so
 * that max |w| * 2^wexp lies in [2^13, 2^14)): the caller multiplies the layer's `scale` by 2^-(wexp +
 * h2_exp_src).  Modes and geometry as sfh_pack_s3_weights; packed size = 2/3 of the S3 size. */
int64_t sfh_packed_h2_weight_bytes(int ksize, int c0, int c1, int cout_virtual);
int sfh_pack_h2_weights(const float* w, void* packed, int ksize, int c0, int c1, int cout_virtual,
                        int mode, int aux, int wexp, void* stream);
/* fp32 NHWC (rows = B*H, W, cs) <-> H2 (rows, cs/32, 2, 4, W, 8) with the tensor's exponent act_exp (-64 .. 64);
 * overflow: optional device word, OR-ed with 1 when a value was saturated; range: optional device word, atomic
 * max of the bit patterns of |v * 2^act_exp| (as sfh_conv_desc.h2_range). */
int sfh_f32_to_h2(const float* src, void* dst, int64_t rows, int W, int cs, int act_exp, uint32_t* overflow,
                  uint32_t* range, void* stream);
int sfh_h2_to_f32(const void* src, float* dst, int64_t rows, int W, int cs, int act_exp, void* stream);

/* The same 3x3 stride-1 convolution on two-plane fp16 operands for launches whose grid of 256-pixel x 64-cout workgroups
 * leaves the chip under-filled (ResNet layer4 at batch 16, small batches): tiles of 12 x 20 pixels x 32 couts, halo and weight
 * fragments through LDS.  Same descriptor, same packed weights (sfh_pack_h2_weights), same accumulation order per output -
 * results are bit-identical to sfh_conv_s3_fwd's.  Restrictions: src_fmt H2, one source, ksize 3, stride 1, plain NHWC output
 * (H2 or fp32), optional residual of the destination's format + ReLU, no pooled output / head / acc_init / split-K / statistics. */
int sfh_conv_small_fwd(const sfh_conv_desc* d, void* stream);

/* The first conv of a fused Up block - conv3x3(cat([skip, ConvTranspose2d(x)])) + BatchNorm + ReLU, unet/unet_parts.py:52-68 - as ONE
 * launch (round 5 experiment; the two-launch form is sfh_conv_s3_fwd with ksize 2 / SFH_OUT_UPSCATTER2 writing an fp32 partial +
 * sfh_conv_s3_fwd with acc_init): src0 = the skip tensor (B,H,W,cs0) H2, src1 = the low-resolution x (B,h1,w1,cs1) H2 with
 * H in {2*h1, 2*h1+1}, W alike; wpacked / scale / shift / relu = the skip-half 3x3 conv's (c0 -> cout); up_wpacked / up_scale /
 * shift_border = the composed 2x2 conv's (c1 -> 4*cout virtual couts), scale and border shifts expressed in the skip-half's
 * accumulator units (what the two-launch form passes to its first launch); dst H2 (B,H,W,dst_cs).  Same products in the same
 * order per output as the two-launch form: bit-identical results.                                                        */
int sfh_conv_upfused_fwd(const sfh_conv_desc* d, void* stream);

/* First UNet layer (inc.double_conv.0, unet/unet_parts.py:15; 3 input channels stored as 4):
 * tap-packed fp32 MFMA kernel, k = channel, one MFMA k-step per tap.  Same descriptor/epilogue as
 * sfh_conv_fwd; wpacked from sfh_pack_c4_weights ((cout/64)*9*256 floats), w is OIHW (cout,cin<=4,3,3). */
int sfh_conv3x3_c4_fwd(const sfh_conv_desc* d, void* stream);
int sfh_pack_c4_weights(const float* w, float* packed, int cin, int cout, void* stream);

/* The same first layer on the fp16 matrix cores ("f16x3": two fp16 planes per operand, three products, fp32 accumulation;
 * unet/unet_parts.py:15).  The frame is split ONCE by sfh_frame_to_h2 into the FH2 frame tensor, 16 bytes per pixel =
 * [fp16 plane 0 of channels 0..3 | fp16 plane 1 of channels 0..3] of x * 2^act_exp (format as SFH_FMT_H2: saturating,
 * `range` receives the largest |x * 2^act_exp|, `overflow` is OR-ed with 1 on saturation; both optional); dst_nhwc4
 * (optional) receives the fp32 NHWC copy with 4 stored channels that sfh_nchw_to_nhwc would write.
 * sfh_conv3x3_c4h2_fwd: src0 = the FH2 tensor (B,H,W) x 16 bytes (src_fmt = SFH_FMT_FH2, c0 <= 4, cs0 = 4; a plain fp32 NHWC4
 * frame - src_fmt SFH_FMT_F32, the source of sfh_conv3x3_c4_fwd - is rejected), h2_exp_src = act_exp, wpacked from sfh_pack_c4h2_weights (planes of
 * w * 2^wexp; the caller multiplies `scale` by 2^-(wexp + act_exp)), dst H2 or fp32; descriptor fields as sfh_conv_fwd. */
int sfh_frame_to_h2(const float* src_nchw, float* dst_nhwc4, void* dst_fh2, int batch, int C, int H, int W, int act_exp,
                    uint32_t* overflow, uint32_t* range, void* stream);
int64_t sfh_packed_c4h2_weight_bytes(int cout);
int sfh_pack_c4h2_weights(const float* w, void* packed, int cin, int cout, int wexp, void* stream);
int sfh_conv3x3_c4h2_fwd(const sfh_conv_desc* d, void* stream);

/* Number of floats of the packed weight buffer for a conv with the given geometry. */
int64_t sfh_packed_weight_floats(int ksize, int c0, int c1, int cout_virtual);

/* Re-layout checkpoint weights into MFMA fragment order.
 * mode = 0: w is OIHW (cout, c0+c1, k, k)  (nn.Conv2d), ksize 1 or 3
 * mode = 1: w is IOHW (cin, cout, 2, 2)    (nn.ConvTranspose2d), ksize must be 1 and
 *           cout_virtual = 4*cout.
 * mode = 2: w is OIHW (cout, aux, 7, 7), the stride-2 pad-3 stem of ResNetSTN
 *           (models/resnet.py:172), re-expressed as a 4x4 conv (pad 2 before / 1 after) over
 *           the 2x2 space-to-depth input produced by sfh_space_to_depth2; ksize = 4,
 *           c0 = 4 * (padded channels of the un-shuffled tensor), aux = real cin.
 * mode = 3: backward-data of a stride-1 Conv2d: w is OIHW (c0, aux, k, k); the packed conv maps dz
 *           (c0 = the layer's cout) to dx (cout_virtual >= aux = the layer's cin) with the taps
 *           flipped and in/out channels swapped.
 * mode = 4: backward-data of ConvTranspose2d k2 s2: w is IOHW (cout_virtual, aux, 2, 2); a 1x1 conv
 *           over sfh_space_to_depth2(dY) (c0 = 4*aux channels) producing dx.                      */
int sfh_pack_conv_weights(const float* w, float* packed, int ksize, int c0, int c1,
                          int cout_virtual, int mode, int aux, void* stream);

/* (B,H,W,cs) -> (B,ceil(H/2),ceil(W/2),4*cs): out[Y][X][(py*2+px)*cs + c] = in[2Y+py][2X+px][c]
 * (zero where 2Y+py >= H or 2X+px >= W). */
int sfh_space_to_depth2(const float* src, float* dst, int batch, int H, int W, int cs, void* stream);

/* scale = gamma / sqrt(var + eps); shift = (conv_bias - mean) * scale + beta
 * (eval-mode BatchNorm2d folded behind the conv; conv_bias may be NULL).
 * With gamma == NULL: scale = 1, shift = conv_bias (plain bias epilogue).
 * `repeat` replicates the n-vector (transposed conv: 4 sub-positions).                    */
int sfh_fold_bn(const float* conv_bias, const float* gamma, const float* beta,
                const float* mean, const float* var, float eps, int n, int repeat,
                float* scale, float* shift, void* stream);

/* Decoded frames uint8 (B,H,W,C) -> float32 (B,C,H,W) = value/255: the dataset's preprocessing
 * (utils/dataset.py:154-159, :323-330: `img.transpose((2,0,1)) / 255` -> FloatTensor) on the GPU. */
int sfh_u8hwc_to_f32nchw(const uint8_t* src, float* dst, int batch, int C, int H, int W, void* stream);

/* Same with the video path's 2x downscale in front (utils/dataset.py:312-316: frames wider than the target
 * go through cv2.resize(..., INTER_AREA)): src uint8 (B,2H,2W,C) -> dst (B,C,H,W) = ((a+b+c+d+2)>>2)/255,
 * OpenCV's 2x2 area fast path.  Other scale factors are not covered. */
int sfh_u8hwc_area2_to_f32nchw(const uint8_t* src, float* dst, int batch, int C, int H, int W, void* stream);

/* Any integer downscale factor k = 2 .. 16 of the same call (1920x1080 -> 640x360 is k = 3): src uint8 (B,kH,kW,C) ->
 * dst (B,C,H,W).  k = 2 is the special case above; otherwise OpenCV's resizeAreaFast_ rule: the k x k block summed in int,
 * times the float 1.f / (k * k), rounded half to even, saturated to 0 .. 255, then / 255 (utils/dataset.py:312-330). */
int sfh_u8hwc_areak_to_f32nchw(const uint8_t* src, float* dst, int batch, int C, int H, int W, int k, void* stream);
/* Integer factors that differ per axis, or exceed 16 (kx horizontal, ky vertical, 1 .. 64 each): the same resizeAreaFast_ rule
 * with a kx x ky block - the block summed in int, times the float 1.f / (kx * ky), rounded half to even.  src (B,ky*H,kx*W,C). */
int sfh_u8hwc_areaxy_to_f32nchw(const uint8_t* src, float* dst, int batch, int C, int H, int W, int kx, int ky, void* stream);

/* The same call for ANY downscale (both factors >= 1, at least one not an integer: e.g. 1920x1080 -> 1024x576): OpenCV's
 * generic INTER_AREA path, resizeArea_ over the per-axis tables of computeResizeAreaTab (published imgproc/resize.cpp): every
 * destination pixel is sum_j beta_j * (sum_k alpha_k * S[sy_j][sx_k]) in float, products and sums individually rounded in
 * table order, then rounded half to even, saturated to 0 .. 255, / 255.
 * sfh_resize_area_tab (HOST code, no GPU): the table of one axis - for destination index d the entries ofs[d] .. ofs[d+1]-1 give
 * source index si[] and weight alpha[]; ofs has dsize + 1 ints, si / alpha room for `cap` entries (2 * dsize + ssize always
 * suffices); returns the number of entries, or -1 (bad sizes, dsize > ssize, cap too small).
 * sfh_u8hwc_area_to_f32nchw: src uint8 (B,Hs,Ws,C) -> dst float32 (B,C,Hd,Wd); the six table arrays are DEVICE pointers. */
int sfh_resize_area_tab(int ssize, int dsize, int32_t* ofs, int32_t* si, float* alpha, int cap);
int sfh_u8hwc_area_to_f32nchw(const uint8_t* src, float* dst, int batch, int C, int Hs, int Ws, int Hd, int Wd,
                              const int32_t* xofs, const int32_t* xsi, const float* xalpha, const int32_t* yofs,
                              const int32_t* ysi, const float* ybeta, void* stream);

/* (B, C, H, W) fp32 -> (B, H, W, cs) fp32, channels >= C zero-filled. */
int sfh_nchw_to_nhwc(const float* src, float* dst, int batch, int C, int H, int W, int cs,
                     void* stream);
/* (B, H, W, cs) -> (B, C, H, W) */
int sfh_nhwc_to_nchw(const float* src, float* dst, int batch, int C, int H, int W, int cs,
                     void* stream);

/* F.interpolate on NCHW planes (planes = B*C): mode 1 = 'bilinear' (align_corners as given; the input
 * resize of forward_unet, models/reconstructor.py:136, uses False), mode 0 = 'nearest' (the logits / uv
 * resize, models/reconstructor.py:153,156). */
int sfh_resize_nchw(const float* src, float* dst, int64_t planes, int hs, int ws, int hd, int wd, int mode,
                    int align_corners, void* stream);
/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) of the bilinear Up variant
 * (unet/unet_parts.py:49) on fp32 NHWC: (B,H,W,C) -> (B,2H,2W,C). */
int sfh_upsample2x_bilinear_nhwc(const float* src, float* dst, int batch, int H, int W, int C, void* stream);

/* OutConv (unet/unet_parts.py:74-77): 1x1 conv cin -> nc (nc <= 8) + bias.
 * x: NHWC (B,H,W,cin), w: (nc,cin,1,1), logits_nchw: (B,nc,H,W) or NULL.
 * Optional fused consumers:
 *   argmax_u8 (B,H,W): argmax over classes == preds_to_masks (utils/postprocess.py:7-18);
 *   stn_in  (B,H,W,stn_cs): [logits(nc), frame channels (3, read from frame_nhwc with
 *   stride frame_cs), zero padding] == torch.cat((logits, x), 1) of
 *   models/reconstructor.py:179,214 in NHWC.                                               */
int sfh_outconv_fwd(const float* x, int cin, const float* w, const float* bias, int nc,
                    int batch, int H, int W, float* logits_nchw, uint8_t* argmax_u8,
                    float* stn_in, int stn_cs, const float* frame_nhwc, int frame_cs,
                    void* stream);

/* Homography warp of the court template == kornia HomographyWarper(h, w, mode,
 * normalized_coordinates=True)(template, theta) as called by Reconstructor.warp
 * (models/reconstructor.py:109-118), fused with the predict-mode epilogue
 * `* mask_classes` and `.type(int32)` (models/reconstructor.py:223,240).
 * theta: (B,3,3); tmpl: (.,1,ht,wt) with batch stride tmpl_bstride floats (0 = one shared
 * image); out_f32 (B,h,w) and/or out_i32 (B,h,w) = trunc(value * out_scale).
 * mode: 0 nearest, 1 bilinear.                                                            */
int sfh_homography_warp_fwd(const float* theta, const float* tmpl, int64_t tmpl_bstride,
                            int ht, int wt, int batch, int h, int w, int mode,
                            float out_scale, float* out_f32, int32_t* out_i32, void* stream);
/* Test hook (never on the product path): exhaustive GPU sweep of the two exact-division shortcuts of the
 * warp kernel against the IEEE divisions they replace - 1/z for every float with 2^-64 <= |z| <= 2^64, and
 * create_meshgrid's (i/(n-1) - 0.5)*2 for every 0 <= i < n <= 16385.  mismatches: DEVICE int64[2].        */
int sfh_selftest_warp_arith(int64_t* mismatches, void* stream);

/* transform_poi (models/reconstructor.py:120-130): inverse(theta) applied to the court
 * points, then p/2 + 0.5 if normalize.  theta (B,3,3), poi (B,N,2) -> out (B,N,2).        */
int sfh_poi_project_fwd(const float* theta, const float* poi, int batch, int npts,
                        int normalize, float* out, void* stream);

/* Consistency score (models/reconstructor.py:226-238): mean over pixels of
 * cross_entropy(logits[:, :, y, x], mask[y, x]).  logits NCHW (B,nc,H,W), mask int32
 * (B,hm,wm) (nearest-resized to HxW when sizes differ), partial: workspace of
 * sfh_ce_workspace_floats(batch, H, W) floats, score: (B).                                */
int64_t sfh_ce_workspace_floats(int batch, int H, int W);
int sfh_consistency_ce_fwd(const float* logits, const int32_t* mask, int batch, int nc, int H,
                           int W, int hm, int wm, float* partial, float* score, void* stream);

/* The two calls above fused for predict()'s usual cases (nearest mode, 4 classes; the logits (B,nc,hl,wl) have the warp's size,
 * or exactly half of it in both directions - predict.py's default geometry, where the reference scores through the nearest-
 * resized mask, i.e. logit pixel (y, x) against mask pixel (2y, 2x); models/reconstructor.py:223-240): out_i32 (B,h,w) = trunc(warp * out_scale) bit-identical to sfh_homography_warp_fwd, and
 * score (B) = mean over the pixels of cross_entropy(logits[:, :, y, x], out_i32[y, x]) - every wave scores the pixels it warps
 * while the class ids are still in registers, the logits (B,nc,h,w) are streamed once, the mask is never read back.  Two
 * launches (the fused kernel + a one-wave-per-frame sum of its partials in fp64, fixed order: deterministic) instead of three.
 * partial: workspace of sfh_warp_consistency_workspace_floats(batch, h, w) floats.  Other class counts / a mask of another
 * size: the two separate entries.                                                                                       */
int64_t sfh_warp_consistency_workspace_floats(int batch, int h, int w);
int sfh_warp_consistency_fwd(const float* theta, const float* tmpl, int64_t tmpl_bstride, int ht, int wt, int batch, int h,
                             int w, float out_scale, const float* logits, int nc, int hl, int wl, int32_t* out_i32,
                             float* partial, float* score, void* stream);

/* Output masks as predict.py writes them (predict.py:286-315; utils/postprocess.py:7-61):
 * src is int32 class ids (src_kind 0, e.g. predict()'s warp_mask), uint8 ids (1) or fp32 logits
 * NCHW (2: argmax over nc classes, first maximum wins); nearest resize (hs,ws)->(hd,wd) with
 * OpenCV's INTER_NEAREST index rule; mode 0 gray (ids), 1 bin ((id>0)*255), 2 rgb (palette:
 * HOST pointer to 8x3 bytes, colour of class k at palette[3k..3k+2]; out is (B,hd,wd,3)).   */
int sfh_mask_format_fwd(const void* src, int src_kind, int nc, int batch, int hs, int ws, int hd,
                        int wd, int mode, const uint8_t* palette, uint8_t* out, void* stream);

/* ResNetSTN pieces (models/resnet.py:235-254). */
/* MaxPool2d(kernel 3, stride 2, padding 1) on NHWC (B,H,W,C) -> (B,Ho,Wo,C). */
int sfh_maxpool3x3s2_fwd(const float* x, float* y, int batch, int H, int W, int C, void* stream);
/* The same pooling written straight into a split tensor (dst_fmt SFH_FMT_H2 with exponent act_exp + the optional overflow /
 * range words of the format, or SFH_FMT_S3): what sfh_maxpool3x3s2_fwd followed by sfh_f32_to_h2 / sfh_f32_to_s3 gives, in
 * one pass.  C a multiple of 32. */
int sfh_maxpool3x3s2_split_fwd(const float* x, void* y, int batch, int H, int W, int C, int dst_fmt, int act_exp,
                               uint32_t* overflow, uint32_t* range, void* stream);
/* AdaptiveAvgPool2d(1) + flatten + Linear(C -> nout): x NHWC (B,H,W,C), w (nout,C), feat (B,C) = the
 * pooled features (workspace / second output), out (B,nout). */
int sfh_avgpool_linear_fwd(const float* x, const float* w, const float* bias, int batch, int H,
                           int W, int C, int nout, float* feat, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training mode (Reconstructor.forward under net.train() + loss.backward(), train.py:170,233).
 * Activations fp32 NHWC with channel stride == C; reductions over pixels accumulate in fp64 into
 * caller-zeroed `acc` buffers.
 * --------------------------------------------------------------------------------------------- */
/* nn.BatchNorm2d(training=True) (unet/unet_parts.py:16,19; models/resnet.py:174 etc.):
 * acc[0][c] += sum_p z[p][c], acc[1][c] += sum_p z[p][c]^2   (acc: 2*C doubles)                  */
int sfh_bn_stats(const float* z, int64_t npix, int C, double* acc, void* stream);
/* the same sums from the table a convolution's epilogue left (sfh_conv_desc.stats_partial: [rows][2][C] float64):
 * acc[0][c] += sum_r partial[r][0][c], acc[1][c] += sum_r partial[r][1][c].                                */
int sfh_bn_stats_partials(const double* partial, int rows, int C, double* acc, void* stream);
/* mean_invstd[0][c] = mean, [1][c] = 1/sqrt(biased var + eps); running stats (optional pair) updated
 * with `momentum` and the unbiased variance like torch.                                           */
int sfh_bn_finalize(const double* acc, int64_t npix, int C, float eps, float momentum, float* running_mean,
                    float* running_var, float* mean_invstd, int64_t* num_batches_tracked, void* stream);

/* sfh_bn_stats_partials + sfh_bn_finalize in one launch (round 6): `partial` = the (rows, 2, C) float64 table of per-wave sums of
 * z and z^2 that a conv epilogue left (sfh_conv_desc.stats_partial); summed in a fixed order (no atomics), then finished exactly
 * as sfh_bn_finalize does.                                                                                              */
int sfh_bn_finalize_partials(const double* partial, int rows, int64_t npix, int C, float eps, float momentum,
                             float* running_mean, float* running_var, float* mean_invstd, int64_t* num_batches_tracked,
                             void* stream);   /* num_batches_tracked
                    (optional): nn.BatchNorm2d's int64 step counter, incremented by one */
/* y = [relu]((z - mean) * invstd * gamma + beta [+ residual]); y_s3 (optional, C % 32 == 0, W = row
 * length of the (rows, W, C) tensor): the same values again in the split layout split_fmt (SFH_FMT_S3 or
 * SFH_FMT_H2; H2: `overflow` as in sfh_f32_to_h2) for the next convolution.  With y_s3, y may be NULL (only the
 * split copy is written: a layer whose consumers all read the split copy moves a third less). */
int sfh_bn_apply(const float* z, const float* mean_invstd, const float* gamma, const float* beta,
                 const float* residual, int relu, int64_t npix, int C, float* y, void* y_s3, int W,
                 int split_fmt, uint32_t* overflow, void* stream);
/* backward of bn_apply: with g = dy * (y > 0) (or dy when relu == 0; y == NULL with relu: the layer has no residual and
 * the sign of y is recomputed from z, gamma, beta - 8 instead of 12 bytes read per element), acc[0][c] += sum g,
 * acc[1][c] += sum g * xhat  (= dbeta, dgamma);  then
 * dz = gamma * invstd * (g - acc[0]/N - xhat * acc[1]/N), and dres = g (gradient of the residual
 * branch, optional).                                                                              */
int sfh_bn_bwd_reduce(const float* dy, const float* y, const float* z, const float* mean_invstd,
                      const float* gamma, const float* beta, int relu, int64_t npix, int C, double* acc, void* stream);
int sfh_bn_bwd_apply(const float* dy, const float* y, const float* z, const float* mean_invstd,
                     const float* gamma, const float* beta, const double* acc, int relu, int64_t npix, int C,
                     float* dz, float* dres, void* dz_s3, int W, int split_fmt, uint32_t* overflow, float* acc_f32,
                     void* stream);   /* dz_s3: optional split copy of dz (split_fmt, overflow: as sfh_bn_apply; with it dz may be NULL); acc_f32 (optional,
                                         2*C floats): acc as float32 = dbeta | dgamma for the caller */
/* acc[c] += sum_p x[p][c] over a channel slice of a (npix, cs) tensor: conv / transposed-conv bias
 * gradients.                                                                                       */
int sfh_colsum(const float* x, int64_t npix, int C, int cs, double* acc, void* stream);
/* First pass of nn.ConvTranspose2d(c, c/2, 2, stride=2).backward (unet/unet_parts.py:52, called from train.py:233):
 * du (B, 2h, 2w, cout) fp32 -> the space-to-depth tensor s (B, h, w, 4*cout), s[(py*2+px)*cout + co] at (y, x) =
 * du[2y+py][2x+px][co], written in the split format only (split_fmt SFH_FMT_S3 / SFH_FMT_H2 at the default exponent;
 * overflow as sfh_bn_apply), and the column sums of du (fp64, the bias gradient) - one pass instead of
 * sfh_colsum + sfh_space_to_depth2 + sfh_f32_to_h2.  acc = [acc_rows][cout] doubles, zeroed by the caller: a workgroup adds
 * its sums into row (its index) % acc_rows - same-address fp64 atomics are slow, 32 rows take them out of the way - and the
 * caller adds the rows up (sfh_bn_stats_partials with C = cout / 2; acc_rows = 1: acc is the result).  cout % 8 == 0.   */
int sfh_s2d_split_colsum(const float* du, int batch, int h, int w, int cout, void* s_split, int split_fmt,
                         double* acc, int acc_rows, uint32_t* overflow, void* stream);
/* nn.MaxPool2d(2) (unet/unet_parts.py:33) on NHWC, forward (floor) and backward: the gradient goes to
 * the first maximum of each window in scan order, like ATen; accumulate != 0 adds into dx.          */
int sfh_maxpool2_fwd(const float* x, float* y, int batch, int H, int W, int C, void* stream);
int sfh_maxpool2_bwd(const float* x, const float* dy, float* dx, int batch, int H, int W, int C,
                     int accumulate, void* stream);
/* The encoder's skip tensors in training mode (DoubleConv -> [MaxPool2d(2), Up], unet/unet_parts.py:16-20,33; split formats):
 * sfh_bn_apply_pool: y = relu(bn(z)) with the statistics in mean_invstd AND maxpool2(y) from one pass over z, both written in
 *   the split format only (y_s3: (B,H,W,C), pool_s3: (B,H/2,W/2,C); split_fmt / overflow as sfh_bn_apply) - bit for bit the
 *   planes of sfh_bn_apply -> sfh_maxpool2_fwd -> sfh_f32_to_h2.  C % 32 == 0.
 * sfh_pool2_bwd_bn_reduce: dx (+)= max-pool routing of dpool (window values recomputed from z; accumulate != 0: dx holds the
 *   gradient from y's other consumer) and, dx being the layer's total gradient then, acc (2*C doubles, zeroed by the caller)
 *   += [sum g | sum g * xhat], g = dx * (y > 0): sfh_maxpool2_bwd + sfh_bn_bwd_reduce in one pass.                       */
int sfh_bn_apply_pool(const float* z, const float* mean_invstd, const float* gamma, const float* beta, int batch, int H,
                      int W, int C, void* y_s3, void* pool_s3, int split_fmt, uint32_t* overflow, void* stream);
int sfh_pool2_bwd_bn_reduce(const float* z, const float* mean_invstd, const float* gamma, const float* beta,
                            const float* dpool, int batch, int H, int W, int C, int accumulate, float* dx, double* acc,
                            void* stream);
/* dst (B,h,w,C) (+)= src[b, y+oy, x+ox, c_off:c_off+C], src (B,Hs,Ws,cs), zero outside: the backward of
 * torch.cat / F.pad in Up (unet/unet_parts.py:59-67).                                              */
int sfh_slice_add(const float* src, int Hs, int Ws, int cs, int c_off, int oy, int ox, float* dst,
                  int batch, int h, int w, int C, int accumulate, void* stream);
/* dst (B,H,W,C): dst[2j][2i] = src[j][i], zero elsewhere - turns the backward of a stride-2 conv into
 * a stride-1 problem (models/resnet.py:56,172 stride-2 convs).                                     */
int sfh_zero_stuff2(const float* src, float* dst, int batch, int ho, int wo, int H, int W, int C,
                    void* stream);
/* Weight gradient of a stride-1 conv on the fp32 matrix cores (backward-filter):
 *   raw[m][tap][n_off + n] += sum_{b,y,x} dz[b][y][x][m] * xin[b][y + ky - pad][x + kx - pad][n]
 * dz (B,H,W,dz_cs) with M channels used; x (B,xh,xw,x_cs) with N channels, placed at (pad_top,pad_left)
 * inside the HxW conv input frame (second source of a concat), zero outside; ksize 1, 3 (pad 1) or
 * 4 (pad 2 before / 1 after: the space-to-depth stem, N <= 32); raw is (M, ksize^2, raw_n) fp32,
 * caller-zeroed, accumulated with atomics.                                                          */
int sfh_conv_wgrad(const float* dz, int dz_cs, int M, const float* x, int x_cs, int xh, int xw, int N,
                   int pad_top, int pad_left, int batch, int H, int W, int ksize, float* raw, int raw_n,
                   int n_off, void* stream);
/* The UNet's first layer (3 input channels stored as 4; unet/unet_parts.py:15-17) with its BatchNorm backward applied while
 * the gradient tiles are loaded: dy = gradient of the BatchNorm + ReLU output, z = the conv output, acc = the finished sums
 * [sum g | sum g * xhat] (sfh_bn_bwd_reduce, or the consumer's backward-data epilogue); raw as sfh_conv_wgrad (M, 9, raw_n).
 * Same values as sfh_bn_bwd_apply followed by sfh_conv_wgrad, without writing and re-reading dz.                        */
int sfh_conv_wgrad_c4_bn(const float* dy, const float* z, const float* mean_invstd, const float* gamma, const float* beta,
                         const double* acc, int M, const float* x, int N, int batch, int H, int W, float* raw, int raw_n,
                         void* stream);

/* The same weight gradient for 3x3 (pad 1) and 1x1 stride-1 convs on the 16-bit matrix cores, fp32-equivalent ("bf16x6",
 * six bf16 MFMA products per fp32 product, as sfh_conv_s3_fwd): dz and the layer input are S3 (split-bf16)
 * tensors - dz (B,H,M/32,3,4,W,8) with M a multiple of 64; x (B,xh,x_channels/32,3,4,xw,8) of which the
 * first N channels (multiple of 32) are used.  raw (M, ksize^2, raw_n) fp32, caller-zeroed, atomics.    */
int sfh_conv_wgrad_s3(const void* dz_s3, int M, const void* x_s3, int x_channels, int xh, int xw, int N,
                      int pad_top, int pad_left, int batch, int H, int W, int ksize, float* raw, int raw_n, int n_off,
                      int fmt, void* stream);   /* fmt: SFH_FMT_S3, or SFH_FMT_H2 (both tensors two-plane fp16, three
                                                   fp16 MFMA products per product) */

/* Backward of OutConv (unet/unet_parts.py:74-77): dlogits NCHW (B,nc,H,W), x NHWC (B,H,W,cin);
 * dx NHWC (optional), acc_w (nc*cin doubles) += dW, acc_b (nc doubles) += db (caller-zeroed).        */
int sfh_outconv_bwd(const float* x, int cin, const float* w, const float* dlogits_nchw, int nc, int batch,
                    int H, int W, float* dx, double* acc_w, double* acc_b, void* stream);
/* The same pass when x is the BatchNorm + ReLU output of the layer in front and the head is its only consumer (up4.conv.3 ->
 * OutConv, unet/unet_model.py): x is recomputed from that layer's conv output z (mean_invstd, gamma, beta: sfh_bn_apply's
 * arithmetic, same bits) instead of read, and acc_bn (2 * cin doubles, zeroed by the caller) += [sum g | sum g * xhat] with
 * g = dx * (x > 0) - the layer's BatchNorm backward sums, without sfh_bn_bwd_reduce's pass over dx and z.                  */
int sfh_outconv_bwd_bn(const float* z, const float* mean_invstd, const float* gamma, const float* beta, int cin,
                       const float* w, const float* dlogits_nchw, int nc, int batch, int H, int W, float* dx,
                       double* acc_w, double* acc_b, double* acc_bn, void* stream);

/* ResNetSTN backward pieces (models/resnet.py:235-254).
 * MaxPool2d(3, stride 2, padding 1) on NHWC: dx (B,H,W,C) from dy (B,Ho,Wo,C), first maximum wins. */
int sfh_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int batch, int H, int W, int C,
                         void* stream);
/* AdaptiveAvgPool2d(1) + Linear: dx (B,H,W,C); acc_w (nout*C doubles) += dW, acc_b (nout) += db.     */
int sfh_avgpool_linear_bwd(const float* x, const float* w, const float* dout, int batch, int H, int W, int C,
                           int nout, float* dx, double* acc_w, double* acc_b, void* stream);
/* Backward-data of the 7x7 s2 p3 stem for input channels [c_off, c_off+nc), nc <= 8 - the logits
 * (or the uv map) inside torch.cat((logits, x[, uv]), 1) (models/reconstructor.py:174-183):
 * dlogits_nchw (B,nc,H,W) += ...; dz (B,Ho,Wo,64) NHWC, w OIHW (64,cin,7,7).                          */
int sfh_stem_bwd_data(const float* dz, const float* w, int cin, int c_off, int nc, int batch, int H, int W,
                      float* dlogits_nchw, void* stream);
/* d loss / d theta of the bilinear homography warp (sfh_homography_warp_fwd mode 1): acc (B*9 doubles,
 * caller-zeroed) += sum over pixels; dout (B,h,w).                                                   */
int sfh_homography_warp_bwd_theta(const float* theta, const float* tmpl, int64_t tmpl_bstride, int ht, int wt,
                                  int batch, int h, int w, const float* dout, double* acc, void* stream);
/* d loss / d theta of sfh_poi_project_fwd: dout (B,N,2) -> dtheta (B,9).                              */
int sfh_poi_project_bwd_theta(const float* theta, const float* poi, int batch, int npts, int normalize,
                              const float* dout, float* dtheta, void* stream);

/* Losses of the training step (train.py:181-224, models/losses.py:33-41) in one pass: segmentation
 * CE(logits, gt) and SmoothL1 (rec_mse = 0) or MSE (rec_mse = 1) of (warp, gt/nc), both averaged per frame, weighted by weight[b] and averaged
 * over the batch, and the consistency CE(logits, trunc(warp*nc)) averaged over all pixels; each times its
 * lambda (0 disables).  focal_flags bit 0 / bit 1: the segmentation / consistency term is
 * kornia.losses.FocalLoss(alpha=1, gamma=2) instead of CE (train.py:101,126; Kornia 0.5/0.6 formula:
 * sum_k (onehot_k + 1e-6) * -(1 - q_k)^2 log q_k with q = softmax + 1e-8).  gt_mask int64 (B,H,W); outputs d/dlogits (NCHW), d/dwarp (B,H,W, optional) and
 * loss3[0..2] += seg, rec, consistency (fp64, caller-zeroed).                                           */
int sfh_train_losses(const float* logits_nchw, const int64_t* gt_mask, const float* weight,
                     const float* warp_mask, int nc, int batch, int H, int W, float lambda_seg,
                     float lambda_rec, int rec_mse, float lambda_cons, int focal_flags, float* dlogits_nchw,
                     float* dwarp, double* loss3, void* stream);
/* ReprojectionLoss (models/losses.py:6-31), reduction 'mean', times lambda: *loss += value; dpoi (B,N,2). */
int sfh_reproj_loss(const float* poi, const float* gt_poi, const float* nonzeros, const float* num_nonzero,
                    int batch, int npts, float lambda, float* dpoi, double* loss, void* stream);
/* nn.utils.clip_grad_value_(clip) + torch.optim.RMSprop step (train.py:88,234-237) over many tensors in
 * one launch.  tensor_table: device array of {float* param; const float* grad; float* square_avg;
 * float* momentum_buf;}; chunk_table: device array of {int32 tensor; int32 count; int64 offset;}, one
 * workgroup per chunk.  clip_value <= 0 disables clipping, momentum == 0 skips the buffer; the gradient
 * is multiplied by grad_scale first (1/world after a data-parallel all-reduce sum, else 1).            */
int sfh_rmsprop_step(const void* tensor_table, const void* chunk_table, int nchunks, float lr, float alpha,
                     float eps, float weight_decay, float momentum, float clip_value, float grad_scale,
                     void* stream);

/* The reference's other two optimizers (train.py:89-92) behind clip_grad_value_(clip), same tables as sfh_rmsprop_step:
 * sfh_sgd_step  = torch.optim.SGD(lr, momentum, weight_decay): buf (4th table pointer) = momentum buffer, zero-filled at first;
 * sfh_adam_step = torch.optim.Adam(lr, (beta1, beta2), eps, weight_decay): buf = exp_avg, sq (3rd pointer) = exp_avg_sq, both
 *                 zero-filled at first; `step` counts from 1 (bias corrections 1 - beta^step, evaluated in double on the host). */
int sfh_sgd_step(const void* tensor_table, const void* chunk_table, int nchunks, float lr, float weight_decay, float momentum,
                 float clip_value, float grad_scale, void* stream);
int sfh_adam_step(const void* tensor_table, const void* chunk_table, int nchunks, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float clip_value, float grad_scale, int step, void* stream);

/* UV-head loss (train.py:136-144,203-208): loss[0] += lambda * per_sample_weighted_criterion(MSELoss | SmoothL1Loss, uv, gt_uv,
 * weight) on (B,C,H,W) tensors and duv = its gradient.  models/losses.py:38-39 reduces a 4-D loss map with
 * torch.mean(loss, dim=(1, 2)) - over channel and row, leaving (B, W) - and multiplies by the weights, which broadcasts along
 * the LAST axis: nweights must be 1 (weight[0] everywhere; the B == 1 case) or W (column w takes weight[w]; the B == W case);
 * anything else is the reference's shape error and returns SFH_E_ARG.  mse: 1 = MSELoss, 0 = SmoothL1Loss(beta 1).          */
int sfh_uv_loss(const float* uv, const float* gt_uv, const float* weight, int nweights, int batch, int C, int H, int W,
                float lambda, int mse, float* duv, double* loss, void* stream);

/* Many small copies in one launch (training: the assembly of all parameter gradients into the flat gradient buffer -
 * weight gradients are permuted views of the backward-filter buffers - and the copy of the BatchNorm statistics a repeated
 * step starts from; torch issues one copy per tensor).  tensor_table: device array of {void* dst (contiguous); const
 * void* src; int32 d1, d2, d3, pad; int64 s0, s1, s2, s3} = logical shape (d0,d1,d2,d3) with the source's element strides,
 * 4-byte elements; chunk_table as sfh_rmsprop_step ({int32 tensor; int32 count; int64 offset}, one workgroup per chunk of
 * dst).  scale != 1: dst = src * scale (float32); scale == 1: the words are moved untouched.                     */
int sfh_multi_copy(const void* tensor_table, const void* chunk_table, int nchunks, float scale, void* stream);

/* sfh_multi_copy with the factor read from device memory (*scale_dev), float32 tensors.                                    */
int sfh_multi_copy_dscale(const void* tensor_table, const void* chunk_table, int nchunks, const float* scale_dev, void* stream);

/* The power-of-two scale of a training step's backward pass (two-plane fp16 gradients), chosen on the device: words = the output
 * of sfh_multi_absminmax over the nheads head-gradient tensors followed by the theta gradient (if has_theta); the largest head
 * gradient goes to [2^(1+shift), 2^(2+shift)) (no heads: theta's to [2^(12+shift), 2^(13+shift))), raised if theta's gradient
 * would sit below 2^-16 while the heads stay below 2^6.  scale2[0] = S, scale2[1] = 1 / S.  *overflow |= 2 for a non-finite
 * seed, |= 4 if no scale fits both (train.py:233: loss.backward() has no such limit - the caller then repeats the step with
 * three-plane bf16 operands).  No host read-back: the following launches take S through the pointer.                      */
int sfh_grad_scale(const uint32_t* words, int nheads, int has_theta, int shift, float* scale2, uint32_t* overflow, void* stream);

/* Backward of sfh_upsample2x_bilinear_nhwc (the bilinear Up variant, unet/unet_parts.py:49): dy (B,2H,2W,C)
 * -> dx (B,H,W,C).                                                                                        */
int sfh_upsample2x_bilinear_nhwc_bwd(const float* dy, float* dx, int batch, int H, int W, int C, void* stream);
/* Backward of sfh_resize_nchw mode 0 ('nearest': the logits / uv resize, models/reconstructor.py:153,156):
 * dy (planes,hd,wd) -> dx (planes,hs,ws).                                                                */
int sfh_resize_nearest_nchw_bwd(const float* dy, float* dx, int64_t planes, int hs, int ws, int hd, int wd,
                                void* stream);

/* Up block fusion (unet/unet_parts.py:52,59-67 + the first conv of DoubleConv): the u-half of
 *   conv3x3(cat([skip, ConvTranspose2d(x)]))  =  a 2x2 conv over the LOW-resolution x per output parity.
 * wconv OIHW (cout, c0+c1, 3, 3), wt IOHW (cx, c1, 2, 2), bt (c1); scale4 / shift4 (4*cout) = the folded
 * BatchNorm epilogue of the conv repeated per quadrant:
 *   w2 (4*cout, cx, 2, 2): virtual cout (py*2+px)*cout + co, tap (a,b) reads x[Y + py - 1 + a][X + px - 1 + b];
 *   shift_border (16, 4*cout): shift4 + scale4 * (what the transposed conv's bias contributes through the conv
 *   taps that fall inside the up-sampled tensor), per class of the output pixel: 4*(row class) + (col class),
 *   class 0: first row/col of the up-sampled tensor, 2: its last, 3: the padded row/col after it (F.pad with
 *   diff 1), 1: everything else (sfh_conv_desc.shift_border).                                             */
int sfh_compose_up_weights(const float* wconv, int cout, int c0, int c1, const float* wt, int cx,
                           const float* bt, const float* scale4, const float* shift4, float* w2,
                           float* shift_border, void* stream);

/* Second half of a split-K convolution (sfh_conv_desc.ksplit): y = [relu](sum_k slab_k + shift[c] [+ residual]) over
 * (npix, C) fp32 NHWC slabs slab_stride bytes apart -> dst in dst_fmt (F32 NHWC, S3 or H2 with exponent exp_dst; W =
 * row length of the split layouts).  residual (optional): same geometry, res_fmt / exp_res.  overflow / range: as
 * sfh_f32_to_h2.  Replaces the BatchNorm shift + residual add + ReLU of BasicBlock (models/resnet.py:74-80) that
 * sfh_conv_s3_fwd applies in its epilogue when the K loop is not split. */
int sfh_splitk_finish(const float* slabs, int nslabs, int64_t slab_stride, const float* shift, const void* residual,
                      int res_fmt, int exp_res, int relu, int64_t rows, int W, int C, void* dst, int dst_fmt,
                      int exp_dst, uint32_t* overflow, uint32_t* range, void* stream);

/* The ResNetSTN stem (models/resnet.py:172,241-243: 7x7 stride-2 pad-3 conv + BatchNorm + ReLU, <= 8 input
 * channels -> 64) with the split-bf16 arithmetic, K packed by tap (4 taps x 8 channels per MFMA): d->src0 fp32
 * NHWC (B,H,W,8) = cat((logits, frame)) zero-padded, d->wpacked from sfh_pack_stem_weights (w OIHW
 * (64,cin,7,7)), d->dst fp32 NHWC (B,Ho,Wo,64), scale/shift/relu as in sfh_conv_fwd; the other descriptor
 * fields must describe a plain single-source launch.                                                    */
int sfh_stem7x7_fwd(const sfh_conv_desc* d, void* stream);
int64_t sfh_packed_stem_weight_bytes(void);
int sfh_pack_stem_weights(const float* w, void* packed, int cin, int fmt, int wexp, void* stream);   /* fmt: SFH_FMT_S3, or
    SFH_FMT_H2 (planes of w * 2^wexp; the caller folds 2^-(wexp + h2_exp_src) into the layer's scale) */

/* Device calibration probe (bench.py `device_calibration`; not on the hot path): `workgroups` x 4 waves each run `iters` x 64
 * register-resident v_mfma_f32_16x16x32_f16 (fp16 operands with random mantissas): executed FLOPs = workgroups * 4 * iters *
 * 64 * 2 * 16 * 16 * 32.  out: workgroups * 256 floats (sink); clk: two 64-bit words zeroed by the caller - sums of the
 * shader-clock cycles (s_memtime) and of the 100 MHz ticks (s_memrealtime) the sampled waves spent in the loop, so the clock
 * the chip HELD inside the kernel is 100 MHz * clk[0] / clk[1].  The caller times the launch with events on `stream`. */
int sfh_probe_mfma_f16(int iters, int workgroups, float* out, uint64_t* clk, void* stream);

/* ---- engine-build helpers (csrc/hostprep.hip, round 6): what an engine needs while it packs a checkpoint, so that no stock
 * torch kernel runs on the predict path (the reference does this work inside nn.Module construction / cuDNN descriptors).
 *
 * sfh_multi_absminmax: ONE launch over `ntensors` float32 tensors - table = device array of {const float* ptr; int64 numel}
 * pairs - leaves in words[2 t] the bit pattern of max |x| of tensor t and in words[2 t + 1] 0x7FFFFFFF - (bit pattern of min
 * |x|); `words` must be zero-filled by the caller (zero is the identity of both).  A NaN / Inf element shows as a pattern >=
 * 0x7F800000.  The exponent of every two-plane fp16 weight tensor of a model comes from one call and one read-back
 * (replaces a torch abs().max() + host sync per layer); the min leg answers "does any folded BatchNorm scale vanish".      */
int sfh_multi_absminmax(const void* table, int ntensors, uint32_t* words, void* stream);

/* dst[i] = a[i] * factor (op 0), a[i] / b[i % nb] (op 1; factor applied after the division when != 1) or a[i] * b[i % nb] *
 * factor (op 2), float32, n elements; dst may alias a.  The folded scale / shift vectors of a layer (<= 16 K floats).    */
int sfh_vec_op(int op, const float* a, const float* b, int64_t n, int nb, float factor, float* dst, void* stream);

/* Pitched copy of 4-byte words: rows x width, pitches in words (channel slices of OIHW weights: the skip half of an Up
 * block's first conv, unet/unet_parts.py:67).                                                                           */
int sfh_copy2d_words(const void* src, int64_t src_pitch, void* dst, int64_t dst_pitch, int width, int64_t rows, void* stream);

/* dst[0 .. n) = value (4-byte words): range words, flags, constant scale / shift vectors.                                 */
int sfh_fill_words(void* dst, int64_t n, uint32_t value, void* stream);

/* *flag |= 1 (flag zero-filled by the caller) if any of rows 1 .. rows-1 of a (rows, row_words) array of 4-byte words differs
 * from row 0: "is the court template ONE image replicated over the batch" (utils/dataset.py:59), asked once per template. */
int sfh_rows_differ(const void* x, int64_t row_words, int rows, uint32_t* flag, void* stream);

/* ResNet-STN input for the modes the fused OutConv epilogue does not assemble (models/reconstructor.py:174-183,214: "img",
 * "mask", "img+mask+uv", and "img+mask" with resized logits): dst (B,H,W,cs) NHWC = cat((logits (B,nc,H,W), frame (B,cf,H,W),
 * uv (B,cu,H,W)), channel), zero-padded to cs; a source with 0 channels may be NULL.  Replaces torch.cat + a layout pass. */
int sfh_stn_input_assemble(const float* logits, int nc, const float* frame, int cf, const float* uv, int cu, int batch, int H,
                           int W, int cs, float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SFH_AMD_H */
