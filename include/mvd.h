/*
 * mvd.h — C ABI of libmvd_hip.so: the MI355X (gfx950) plane-sweep cost-volume engine that sits
 * behind the reference's model protocol (rmvd.create_model("robust_mvd") / "mvsnet_train").
 *
 * Every entry point replaces a composition of stock torch ops in the reference (the reference has
 * no native code); the file:line each one replaces is cited per function, relative to
 * the reference repository root.
 *
 * Conventions
 *   - all tensors fp32, contiguous, resident in device (HBM) memory; `const float* const*` arguments
 *     are HOST arrays of V device pointers (one per source view), V <= MVD_MAX_VIEWS;
 *   - calibration (intrinsics, poses, projection matrices, depth samples) is passed as DEVICE
 *     pointers too: the kernels derive their per-view coefficients themselves, so a call never
 *     synchronises with the host and can be captured into a hipGraph;
 *   - the caller owns every buffer (inputs, outputs, workspace); the library allocates nothing and
 *     keeps no pointer after the call returns; kernels are enqueued on `stream` (a hipStream_t;
 *     NULL = the default stream) of the CURRENT device and the call returns without waiting;
 *   - re-entrant, no global mutable state except a thread-local error string;
 *   - return value: 0 = MVD_OK, otherwise an mvd_status; mvd_last_error() describes the failure.
 */
#ifndef MVD_H_
#define MVD_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVD_VERSION 100 /* 0.1.0 */
#define MVD_MAX_VIEWS 32

typedef void* mvd_stream_t; /* hipStream_t */

typedef enum {
    MVD_OK = 0,
    MVD_ERR_INVALID_ARG = 1, /* NULL pointer, non-positive dimension, unsupported channel count ... */
    MVD_ERR_WORKSPACE = 2,   /* workspace NULL or smaller than the *_workspace_bytes() answer */
    MVD_ERR_LAUNCH = 3,      /* HIP reported an error when enqueueing (message has hipGetErrorString) */
    MVD_ERR_NO_DEVICE = 4    /* no gfx950 device visible */
} mvd_status;

/* layouts of a 5-D cost volume */
#define MVD_LAYOUT_NCDHW 0 /* (B, C, D, h, w): what the reference's homo_warp / CostRegNet use */
#define MVD_LAYOUT_NDHWC 1 /* (B, D, h, w, C): channel-last, what the engine uses between its own kernels */
/* OR into `out_layout` of mvd_warp_variance_f32: compute the sampling positions with the reference's own
 * operation chain (IEEE divisions, normalise then un-normalise), rounding for rounding, instead of the default
 * folded form ix = X * rcp(Z) * W/(W-1) - 0.5.  The default is within 1e-4 px of the chain; exact costs ~5 % (C = 32,
 * channel-last: only the locate phase of the tile kernel changes). */
#define MVD_GRID_EXACT 0x100
/* OR into `out_layout` of mvd_warp_variance_f32: key_feat and every src_feat are not (B,C,h,w) maps but already the
 * zero-bordered channel-last copies (B,h+3,w+3,C) the kernel gathers from (map at rows/cols 1..h / 1..w, zeros around),
 * as mvd_conv2d_bn_relu_f32 writes them with MVD_LAYOUT_NHWC_BORDER: the re-packing launches are skipped and the
 * workspace only has to hold the composed transforms (MVD_MAX_VIEWS*B*12 floats, rounded up to 256 bytes). */
#define MVD_FEAT_NHWC_BORDER 0x200

/* layouts of a 4-D feature map */
#define MVD_LAYOUT_NCHW 0        /* (B, C, h, w): the reference's */
#define MVD_LAYOUT_NHWC 1        /* (B, h, w, C): channel-last, between the engine's own 2-D layers */
#define MVD_LAYOUT_NHWC_BORDER 2 /* (B, h+3, w+3, C): channel-last with the map at (1,1) and zeros around: what K3 gathers from */

int mvd_version(void);
/* thread-local; valid until the next failing call on this thread */
const char* mvd_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Path A (robust_mvd): inverse-depth plane sweep with dot-product correlation
 * ---------------------------------------------------------------------------------------------- */

/* K1 — replaces PlanesweepCorrelation.forward for ALL source views in one call:
 *   EpipolarCoeffs.from_calib            rmvd/models/blocks/planesweep_corr.py:228-300
 *   us_from_ds / vs_from_ds (+non-finite replacement)                         :333-349
 *   visibility mask of get_plane_sweep_sampling_points                        :489-512
 *   TorchCorr.forward (all-pairs matmul, /sqrt(C), grid_sample lookup, masks) :152-195 with warp() :49-104
 *   correlate() loop over views                                               :514-521
 *
 *   feat_key   (N,C,h,w)              key-view features
 *   feat_src   V x (N,C,hs,ws)        source-view features
 *   K_key      (N,3,3)                relative intrinsics of the key view (fx,fy,cx,cy in [0,1])
 *   K_src      V x (N,3,3)            relative intrinsics of each source view
 *   T_src2key  V x (N,4,4)            source_to_key_transform (p_src = T p_key)
 *   invdepths  (N,S) if invdepth_batched else (1,S)   sampling inverse depths
 *   corr_out   V x (N,S,h,w)          mask * (1/sqrt(C)) * sum_c f_key * bilinear(f_src)
 *   mask_out   V x (N,S,h,w)          1.0 where all bilinear taps are in bounds (weight sum >= 0.9999)
 *                                     and the plane is in front of both cameras, else 0.0
 * C must be a multiple of 64 (256 in robust_mvd). */
size_t mvd_sweep_corr_workspace_bytes(int N, int C, int h, int w, int hs, int ws, int V);
int mvd_sweep_corr_f32(const float* feat_key, const float* const* feat_src, const float* K_key,
                       const float* const* K_src, const float* const* T_src2key, const float* invdepths,
                       int invdepth_batched, int N, int C, int h, int w, int hs, int ws, int S, int V,
                       float* const* corr_out, float* const* mask_out, void* workspace, size_t workspace_bytes,
                       mvd_stream_t stream);

/* K1 with the block's other options (planesweep_corr.py:371-394, 465-487): sampling inverse depths shared (1,S), per batch
 * element (N,S) or PER KEY PIXEL (N,S,h,w); `corr_scale` multiplies the dot products: 1/sqrt(C) for normalize="dim" (what
 * mvd_sweep_corr_f32 uses), 1 for normalize=False / "before" (the caller normalises the features beforehand). */
#define MVD_INVDEPTH_SHARED 0
#define MVD_INVDEPTH_BATCHED 1
#define MVD_INVDEPTH_PER_PIXEL 2
int mvd_sweep_corr_ex_f32(const float* feat_key, const float* const* feat_src, const float* K_key,
                          const float* const* K_src, const float* const* T_src2key, const float* invdepths,
                          int invdepth_mode, float corr_scale, int N, int C, int h, int w, int hs, int ws, int S, int V,
                          float* const* corr_out, float* const* mask_out, void* workspace, size_t workspace_bytes,
                          mvd_stream_t stream);
/* K1 on the layouts the kernel works in, without the re-packing launches: feat_key (N,h,w,C) channel-last, feat_src[v] zero-bordered
 * channel-last (N,hs+3,ws+3,C) with the map at rows/columns 1.. (what mvd_conv2d_split_f32 writes with its row/image strides),
 * outputs pixel-major: corr_out[v][(n h w + pixel) * out_pixel_stride + s] (and mask_out alike), i.e. (N,h,w,S) maps that the
 * 2-D convolutions behind the sweep read as S channels.  No workspace.  corr_absmax: NULL, or a device float that receives
 * max |corr| over all views by atomic maximum (the caller zeroes it; C <= 256). */
int mvd_sweep_corr_nhwc_f32(const float* feat_key, const float* const* feat_src, const float* K_key, const float* const* K_src,
                            const float* const* T_src2key, const float* invdepths, int invdepth_mode, float corr_scale, int N, int C,
                            int h, int w, int hs, int ws, int S, int V, float* const* corr_out, float* const* mask_out,
                            int out_pixel_stride, float* corr_absmax, mvd_stream_t stream);

/* The sweep of PlanesweepCorrelation(warp_only=True) — replaces WarpOnlyCorr.forward + warp_multi
 *   rmvd/models/blocks/planesweep_corr.py:107-140, 13-45 (reached through correlate(), :514-521):
 * the source features sampled at the S sweep positions of every key pixel (same grids as K1, :228-349, 489-512), times the
 * SAMPLING mask ([sum of in-bounds tap weights >= 0.9999], :96-102; WarpOnlyCorr ignores the visibility mask it is handed).
 * feat_src[v] (N,C,hs,ws); warped_out[v] (N,S,C,h,w); mask_out[v] (N,S,h,w).  normalize_after != 0: the warped features are
 * L2-normalised along C, x / (|x| + 1e-9) (:8-10, 135-136), before the mask is applied.  invdepth_mode as in
 * mvd_sweep_corr_ex_f32.  No workspace. */
int mvd_sweep_warp_f32(const float* const* feat_src, const float* K_key, const float* const* K_src,
                       const float* const* T_src2key, const float* invdepths, int invdepth_mode, int normalize_after, int N,
                       int C, int h, int w, int hs, int ws, int S, int V, float* const* warped_out, float* const* mask_out,
                       mvd_stream_t stream);

/* K2 — replaces the view-weighting arithmetic of LearnedFusion.forward
 *   rmvd/models/blocks/learned_fusion.py:32-48 (softmax over views + 1e-9, mask-weighted mean, fused mask).
 * The per-view score maps (conv3x3+ReLU+conv1x1, :13-17,:28-30) are 2-D convolutions that stay on
 * MIOpen and are passed in.  V == 1 is the caller's pass-through (:50-52) and is rejected here.
 *   corr, mask  V x (N,S,h,w);  score  V x (N,1,h,w);  fused, fused_mask  (N,S,h,w) */
int mvd_fuse_views_f32(const float* const* corr, const float* const* mask, const float* const* score, int N, int S,
                       int h, int w, int V, float* fused, float* fused_mask, mvd_stream_t stream);
/* K2 on pixel-major volumes (N,h,w,S) with pixels in_pixel_stride floats apart (mvd_sweep_corr_nhwc_f32's outputs); score[v]
 * (N,1,h,w).  fused: channel slice of a channel-last buffer (pixels out_pixel_stride floats apart); fused_mask NULL or the same
 * layout; fused_absmax NULL or a device float receiving max |fused| by atomic maximum (the caller zeroes it). */
int mvd_fuse_views_nhwc_f32(const float* const* corr, const float* const* mask, const float* const* score, int N, int S, int h,
                            int w, int V, int in_pixel_stride, float* fused, float* fused_mask, int out_pixel_stride,
                            float* fused_absmax, mvd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Path B (mvsnet): fronto-parallel homography warp, variance aggregation, 3-D regulariser, soft argmin
 * ---------------------------------------------------------------------------------------------- */

/* K3 — replaces the cost-volume construction of MVSNet.forward:
 *   homo_warp for every source view                rmvd/models/blocks/utils.py:222-268
 *   key volume repeat + sum / sum-of-squares loop  rmvd/models/mvsnet.py:124-133
 *   variance                                       rmvd/models/mvsnet.py:135
 *
 *   key_feat      (B,C,h,w)
 *   src_feat      V x (B,C,h,w)
 *   src_proj      V x (B,4,4)      source projection matrices (mvsnet.py:76-88)
 *   key_proj_inv  (B,4,4)          inverse key projection (mvsnet.py:85-86)
 *   depth_values  (B,D)
 *   var_out       (B,C,D,h,w) for MVD_LAYOUT_NCDHW or (B,D,h,w,C) for MVD_LAYOUT_NDHWC
 * C must be a multiple of 4 and <= 64 (32 in MVSNet).  Algorithmic HBM bytes: 4*((V+1)*C*h*w + C*D*h*w)*B. */
size_t mvd_warp_variance_workspace_bytes(int B, int C, int h, int w, int V);
int mvd_warp_variance_f32(const float* key_feat, const float* const* src_feat, const float* const* src_proj,
                          const float* key_proj_inv, const float* depth_values, int B, int C, int D, int h, int w,
                          int V, float* var_out, int out_layout, void* workspace, size_t workspace_bytes,
                          mvd_stream_t stream);
/* The same, and max |var_out| over the whole volume into *absmax_out (device, one float): what
 * mvd_conv3d_bn_relu_f32_split scales its activations by.  A by-product of the kernel's store epilogue for C = 32 /
 * MVD_LAYOUT_NDHWC (no extra pass over the volume), a separate streaming read otherwise. */
int mvd_warp_variance_absmax_f32(const float* key_feat, const float* const* src_feat, const float* const* src_proj,
                                 const float* key_proj_inv, const float* depth_values, int B, int C, int D, int h, int w,
                                 int V, float* var_out, float* absmax_out, int out_layout, void* workspace,
                                 size_t workspace_bytes, mvd_stream_t stream);

/* K3, fp16-feature variant (BASELINE.json configs[3]: "fp16 features"; SURVEY.md 8b names mvd_warp_variance_{f32,f16}).
 * Same operation as mvd_warp_variance_f32 with MVD_FEAT_NHWC_BORDER | MVD_LAYOUT_NDHWC, C = 32:
 *   key_feat, src_feat[v]   fp16 zero-bordered channel-last maps (B,h+3,w+3,32) (map at rows/cols 1..h / 1..w, zeros around)
 *   var_out                 fp16 channel-last volume (B,D,h,w,32)
 * Sampling positions, bilinear weights, the blend and the variance are computed in fp32 exactly as in the f32 entry point
 * (the fp16 taps enter the same fmaf chain): the result equals the f32 kernel's on the same fp16-representable feature
 * values, rounded once (to nearest even) to fp16 on the way out.  Calibration stays fp32.
 * Algorithmic HBM bytes: 2*((V+1)*C*h*w + C*D*h*w)*B — the volume is stored fp16 (1,137.5 MB at configs[3]). */
size_t mvd_warp_variance_f16_workspace_bytes(int B);
int mvd_warp_variance_f16(const void* key_feat, const void* const* src_feat, const float* const* src_proj,
                          const float* key_proj_inv, const float* depth_values, int B, int D, int h, int w, int V,
                          void* var_out, void* workspace, size_t workspace_bytes, mvd_stream_t stream);
/* elementwise fp32 <-> fp16 (round to nearest even); n must be a multiple of 4.
 * Range of the fp16-feature variant (this converter, mvd_warp_variance_f16, mvd_conv3d_bn_relu_f16in): plain IEEE fp16, no
 * scaling — |x| > 65504 becomes +-inf, |x| < 6.1e-5 loses precision (subnormal), |x| < 3e-8 becomes 0; inf / NaN propagate.
 * FeatureNet's last convolution has no normalisation, so the magnitude of the features (and of their variance, which is
 * stored in fp16 too) depends on the checkpoint: the variant is for feature magnitudes of O(1), as with BASELINE.json
 * configs[3]'s weights; the fp32 path (range-scaled split first layer) has no such limit. */
int mvd_convert_f32_to_f16(const float* src, void* dst, long long n, mvd_stream_t stream);
int mvd_convert_f16_to_f32(const void* src, float* dst, long long n, mvd_stream_t stream);

/* homo_warp alone (one view, no aggregation): rmvd/models/blocks/utils.py:222-268 -> (B,C,D,h,w). */
int mvd_homo_warp_f32(const float* src_feat, const float* src_proj, const float* key_proj_inv,
                      const float* depth_values, int B, int C, int D, int h, int w, float* warped_out,
                      void* workspace, size_t workspace_bytes, mvd_stream_t stream);

/* K4 — one layer of CostRegNet (rmvd/models/blocks/mvsnet_components.py:69-123) as an implicit GEMM
 * on the fp32 matrix cores: 3x3x3 Conv3d (stride 1 or 2, padding 1; ConvBnReLU3D :25-41) or
 * ConvTranspose3d (stride 2, padding 1, output_padding 1; :84-109), followed by a per-channel affine
 * (eval-mode BatchNorm folded to scale/shift, or scale=1 / shift=bias for `prob`), optional ReLU and
 * optional skip addition AFTER the ReLU (the `conv4 + conv7(x)` form of :116-121).
 * Activations are channel-last (B,D,h,w,C).  Weights must first be packed with mvd_pack_conv3d_weights_f32. */
#define MVD_CONV3D_STRIDE1 0
#define MVD_CONV3D_STRIDE2 1
#define MVD_DECONV3D_STRIDE2 2
size_t mvd_conv3d_packed_weight_floats(int Cin, int Cout);
/* w: Conv3d layout (Cout,Cin,3,3,3), or ConvTranspose3d layout (Cin,Cout,3,3,3) when mode == MVD_DECONV3D_STRIDE2 */
int mvd_pack_conv3d_weights_f32(const float* w, int Cin, int Cout, int mode, float* packed, mvd_stream_t stream);
/* x (B,Di,hi,wi,Cin) -> y (B,Do,ho,wo,Cout); Do = Di (stride 1), Di/2 (stride 2; Di,hi,wi even), 2*Di (deconv).
 * scale, shift (Cout); skip NULL or (B,Do,ho,wo,Cout).  Cin in {8,16,32,64}; Cout in {1,8,16,32,64}. */
int mvd_conv3d_bn_relu_f32(const float* x, const float* packed_w, const float* scale, const float* shift,
                           const float* skip, float* y, int B, int Di, int hi, int wi, int Cin, int Cout, int mode,
                           int relu, mvd_stream_t stream);
/* The same, and max |y| over the finite outputs into *absmax_out (device, one float): what mvd_conv3d_bn_relu_f32_split scales
 * the activations of a following layer by.  A by-product of the store epilogue of the stride-2 layers (the regulariser's conv1
 * and conv3, whose outputs its split-operand conv2 and conv4 consume); after any other mode a pass over y (mvd_absmax_f32). */
int mvd_conv3d_bn_relu_absmax_f32(const float* x, const float* packed_w, const float* scale, const float* shift,
                                  const float* skip, float* y, float* absmax_out, int B, int Di, int hi, int wi, int Cin,
                                  int Cout, int mode, int relu, mvd_stream_t stream);

/* K4, fp16-input first layer (BASELINE.json configs[3]: "3D-conv regulariser on MFMA, fp16 features"): conv0 of
 * CostRegNet (mvsnet_components.py:78; ConvBnReLU3D 32 -> 8, 3x3x3, stride 1, padding 1, :25-41) on
 * v_mfma_f32_16x16x32_f16 — fp16 operands (the volume of mvd_warp_variance_f16 and the weights rounded to fp16),
 * fp32 accumulation, fp32 BN scale/shift + ReLU epilogue, fp32 output that feeds mvd_conv3d_bn_relu_f32 layers.
 *   x (B,D,h,w,32) fp16 channel-last -> y (B,D,h,w,8) fp32 channel-last.  Only Cin = 32, Cout = 8 is built. */
size_t mvd_conv3d_f16_packed_weight_bytes(int Cin, int Cout);
/* w: Conv3d layout (Cout,Cin,3,3,3) fp32; rounded to fp16 and laid out in MFMA fragment order */
int mvd_pack_conv3d_weights_f16(const float* w, int Cin, int Cout, void* packed, mvd_stream_t stream);
int mvd_conv3d_bn_relu_f16in(const void* x, const void* packed_w, const float* scale, const float* shift, float* y, int B,
                             int D, int h, int w, int Cin, int Cout, int relu, mvd_stream_t stream);

/* K4 first layer, split-operand form (what the model uses by default; mvd_conv3d_bn_relu_f32 on fp32 MFMA is the other
 * form): fp32 input and fp32 weights, each split exactly into two fp16 terms (a = a_hi + 2^-11 a_lo), products
 * a_hi w_hi + 2^-11 (a_hi w_lo + a_lo w_hi) on v_mfma_f32_16x16x32_f16 with fp32 accumulation; the dropped term is
 * 2^-22 a_lo w_lo, i.e. relative error per product <= ~3 * 2^-22 against fp32's own 2^-24, and one accumulator rounding per
 * 32 products instead of per product: the result is at least as close to the exact convolution as the fp32-MFMA kernel's
 * (tests/test_hip_f16.py measures both against a float64 oracle).
 * Range: both operands are scaled by exact powers of two before the split and the result is scaled back, so inputs of any
 * uniform magnitude (1e-30 .. 1e30, denormals included) keep that accuracy.  The activations' scale comes from
 * `x_absmax`, a DEVICE pointer to one float holding max |x| over the whole input (mvd_absmax_f32, or the by-product of
 * mvd_warp_variance_absmax_f32); any upper bound is safe, a tight one most precise: a value v is represented to
 * |error| <= max(2^-22 |v|, 2^-50 max|x|).  inf / NaN inputs give inf / NaN outputs where they reach, as on fp32.
 * x (B,D,h,w,Cin) fp32 -> y (B,D,h,w,Cout) fp32; Cin = 16 or 32, Cout = 8, 16, ... 64 (a workgroup computes 8 output channels
 * of its tile; the regulariser's conv0 (32 -> 8), conv2 (16 -> 16) and conv4 (32 -> 32) use it).  With 16 input channels a
 * 64-byte MFMA fragment spans two x-adjacent voxels: the kw taps go in pairs (0,1), (2, zero weights). */
size_t mvd_conv3d_split_packed_weight_bytes(int Cin, int Cout);
int mvd_pack_conv3d_weights_split(const float* w, int Cin, int Cout, void* packed, mvd_stream_t stream);
int mvd_conv3d_bn_relu_f32_split(const float* x, const float* x_absmax, const void* packed_w, const float* scale, const float* shift,
                                 float* y, int B, int D, int h, int w, int Cin, int Cout, int relu, mvd_stream_t stream);
/* The same, and max |y| over the finite outputs into *y_absmax (device, one float): a by-product of the store epilogue. */
int mvd_conv3d_bn_relu_absmax_f32_split(const float* x, const float* x_absmax, const void* packed_w, const float* scale, const float* shift,
                                        float* y, float* y_absmax, int B, int D, int h, int w, int Cin, int Cout, int relu,
                                        mvd_stream_t stream);
/* K4's stride-2 and transposed layers (and stride-1 layers with many channels) on the split-operand implicit-GEMM kernel of the
 * 2-D engine (csrc/conv2d_split.hip: the depth taps are extra chunks of the reduction; a transposed 3x3x3 stride-2 layer runs as a
 * 2x2x2 convolution with 8 Cout output channels = (parity class, channel)).  x (B,Di,Hi,Wi,Cin) channel-last fp32, Cin a multiple of 8
 * (16 for the transposed form), Cout a multiple of 4; mode MVD_CONV3D_STRIDE1 / _STRIDE2 / MVD_DECONV3D_STRIDE2 with the output
 * sizes of mvd_conv3d_bn_relu_f32; weights in Conv3d / ConvTranspose3d layout.  y = relu?(conv * scale + shift) (+ skip, laid out
 * like y, may be NULL).  x_absmax as for mvd_conv3d_bn_relu_f32_split; y_absmax NULL or a device float that is RAISED to max |y|
 * (atomic maximum: the caller zeroes it, as for mvd_conv2d_split_f32).  workspace NULL or mvd_conv3d_igemm_workspace_bytes bytes (few-voxel layers split the reduction). */
size_t mvd_conv3d_igemm_packed_weight_bytes(int Cin, int Cout, int mode);
int mvd_pack_conv3d_weights_igemm(const float* w, int Cin, int Cout, int mode, void* packed, mvd_stream_t stream);
size_t mvd_conv3d_igemm_workspace_bytes(int B, int Di, int Hi, int Wi, int Cin, int Cout, int mode);
int mvd_conv3d_bn_relu_igemm_f32(const float* x, const float* x_absmax, const void* packed_w, const float* scale, const float* shift,
                                 const float* skip, float* y, float* y_absmax, int B, int Di, int Hi, int Wi, int Cin, int Cout, int mode,
                                 int relu, void* workspace, size_t workspace_bytes, mvd_stream_t stream);
/* ---- Path A's 2-D CNN: split-operand implicit-GEMM convolutions ------------------------------------------------------------
 * Replace torch.nn.functional.conv2d / conv_transpose2d + bias + LeakyReLU (+ torch.cat of the inputs) of the DispNet blocks of
 * robust_mvd (rmvd/models/blocks/dispnet_encoder.py:6-27, dispnet_context_encoder.py, learned_fusion.py:8-20,
 * dispnet_costvolume_encoder.py:7-50, dispnet_decoder.py:36-138).  fp32 in, fp32 out; inside, both operands are split into two
 * fp16 terms (block-scaled by exact powers of two) and multiplied on fp16 MFMA with fp32 accumulation: fp32-grade results
 * (csrc/conv2d_split.hip states the bound).
 * Activations are NHWC with a free pixel stride: x points at the first channel of a slice of Cin_pad channels (a multiple of 8;
 * channels Cin .. Cin_pad-1 must hold zeros) inside pixels `x_pixel_stride` floats apart, y likewise for Cout channels, so that
 * layers read and write slices of the decoder's concat buffers directly.  x, y 16-byte aligned, strides multiples of 4.
 * mode MVD_CONV2D: Conv2d, padding k/2: 1x1 (Cin_pad % 32 == 0), 3x3 stride 1, 3x3 stride 2, 5x5 stride 2; weights (Cout,Cin,k,k).
 * mode MVD_DECONV2D: ConvTranspose2d 4x4, stride 2, padding 1 (Cin_pad % 32 == 0); weights (Cin,Cout,4,4); output 2Hi x 2Wi.
 * mode MVD_CONV2D_IMAGE: Conv2d 7x7 stride 2 padding 3 on a PLANAR (B,3,Hi,Wi) image (Cin = 3, Cin_pad = 8, stride argument unused).
 * x_absmax: device float, max |x| over the input tensor (an upper bound is safe); y_absmax: NULL, or a device float that
 * receives max |y| by atomic maximum (the caller zeroes it; several layers may share one to cover a concat buffer).
 * act: 0 none, 1 LeakyReLU(slope), 2 ReLU; bias NULL or (Cout).  workspace: NULL, or mvd_conv2d_split_workspace_bytes bytes:
 * lets layers with few pixels and many weights split the reduction over workgroups (partial sums added in a fixed order).
 * Output strides (floats): y_pixel_stride between pixels; y_row_stride / y_image_stride between rows / images (0 = dense: a layer
 * may write the interior of a zero-bordered map, which is what mvd_sweep_corr_nhwc_f32 reads); y_channel_stride between channels
 * (0 or 1 = channel-last; with y_pixel_stride = 1 and y_channel_stride = Ho Wo the output is planar (B,Cout,Ho,Wo)). */
#define MVD_CONV2D 0
#define MVD_DECONV2D 1
#define MVD_CONV2D_IMAGE 2
size_t mvd_conv2d_split_packed_weight_bytes(int Cin_pad, int Cout, int KH, int KW, int stride, int mode);
int mvd_pack_conv2d_weights_split(const float* w, int Cin, int Cin_pad, int Cout, int KH, int KW, int stride, int mode, void* packed,
                                  mvd_stream_t stream);
size_t mvd_conv2d_split_workspace_bytes(int B, int Hi, int Wi, int Cin_pad, int Cout, int KH, int KW, int stride, int mode);
int mvd_conv2d_split_f32(const float* x, const float* x_absmax, const void* packed_w, const float* bias, float* y, float* y_absmax, int B,
                         int Hi, int Wi, int Cin_pad, int x_pixel_stride, int Cout, int y_pixel_stride, long long y_row_stride,
                         long long y_image_stride, long long y_channel_stride, int KH, int KW, int stride, int mode, int act, float slope,
                         void* workspace, size_t workspace_bytes, mvd_stream_t stream);
/* F.interpolate(x, size=(2h,2w), mode="bilinear", align_corners=False) of a planar (B,C,h,w) map (the decoder's up-sampled
 * prediction, dispnet_decoder.py:131), written as C channels of a channel-last slice y (pixels y_pixel_stride floats apart);
 * torch's formula and operation order.  y_absmax as above (NULL or atomic maximum). */
int mvd_upsample2x_nhwc_f32(const float* x, float* y, float* y_absmax, int B, int C, int h, int w, int y_pixel_stride, mvd_stream_t stream);
/* max |x[i]| over the FINITE values of n floats (inf and NaN left out, 0 if there is none) into *absmax (device, one
 * float); a streaming read */
int mvd_absmax_f32(const float* x, long long n, float* absmax, mvd_stream_t stream);

/* K6 — one layer of MVSNet's FeatureNet (rmvd/models/blocks/mvsnet_components.py:44-66; ConvBnReLU :8-22) as an
 * implicit GEMM on the fp32 matrix cores: Conv2d k x k with padding k/2 (k = 3 stride 1, or k = 5 stride 2), then a
 * per-channel affine (eval-mode BatchNorm2d folded to scale/shift; scale = 1, shift = bias for the final `feature`
 * conv) and optional ReLU, in one pass.  Weights must first be packed with mvd_pack_conv2d_weights_f32.
 *   x   in_layout MVD_LAYOUT_NCHW: (B,3,hi,wi) — the normalised image, Cin must be 3
 *       in_layout MVD_LAYOUT_NHWC: (B,hi,wi,Cin), Cin in {8,16,32}
 *   y   out_layout MVD_LAYOUT_NHWC: (B,ho,wo,Cout);  MVD_LAYOUT_NCHW: (B,Cout,ho,wo) (the reference's layout);
 *       MVD_LAYOUT_NHWC_BORDER: (B,ho+3,wo+3,Cout) with the map at rows/cols 1..ho/1..wo — the zero-bordered layout
 *       K3 gathers from (mvd_warp_variance_f32 with MVD_FEAT_NHWC_BORDER); the caller zeroes the buffer once, the
 *       kernel writes only the interior.
 *   ho = (hi-1)/stride + 1, wo likewise; Cout in {8,16,32}; scale, shift (Cout). */
size_t mvd_conv2d_packed_weight_floats(int Cin, int Cout, int ksize);
/* w: Conv2d layout (Cout,Cin,k,k) */
int mvd_pack_conv2d_weights_f32(const float* w, int Cin, int Cout, int ksize, float* packed, mvd_stream_t stream);
int mvd_conv2d_bn_relu_f32(const float* x, int in_layout, const float* packed_w, const float* scale, const float* shift,
                           float* y, int out_layout, int B, int hi, int wi, int Cin, int Cout, int ksize, int stride,
                           int relu, mvd_stream_t stream);
/* The same, and *y_absmax (device, one float; the caller zeroes it) RAISED to max |y| over the finite outputs: what a split-operand
 * layer behind this one (mvd_conv2d_split_f32) scales its activations by.  A by-product of the store epilogue. */
int mvd_conv2d_bn_relu_absmax_f32(const float* x, int in_layout, const float* packed_w, const float* scale, const float* shift,
                                  float* y, float* y_absmax, int out_layout, int B, int hi, int wi, int Cin, int Cout, int ksize,
                                  int stride, int relu, mvd_stream_t stream);

/* FeatureNet's two full-resolution layers in one pass — replaces conv0 -> conv1 of FeatureNet
 *   rmvd/models/blocks/mvsnet_components.py:47-48 (ConvBnReLU(3,8,3,1,1), ConvBnReLU(8,8,3,1,1)), BN folded as above.
 *   image (B,3,H,W); w0 (3,3,3,8) and w1 (3,3,8,8): the Conv2d weights permuted to [ky][kx][cin][cout]; scale*, shift* (8);
 *   y (B,H,W,8) channel-last.  The 8-channel intermediate stays in LDS (the two launches write and read it once each).
 *   All pointers except image 16-byte aligned.
 *   tile_absmax (optional, mvd_conv2d_head_tile_count(B,H,W) floats): max |y| over the finite outputs of each workgroup's tile,
 *   written with plain stores — mvd_absmax_f32 over that array is max |y| of the layer (what the split-operand layer behind it
 *   scales by) at the cost of a 35 KB pass instead of one over y. */
size_t mvd_conv2d_head_tile_count(int B, int H, int W);
int mvd_conv2d_head_f32(const float* image, const float* w0, const float* scale0, const float* shift0, const float* w1,
                        const float* scale1, const float* shift1, float* y, float* tile_absmax, int B, int H, int W,
                        mvd_stream_t stream);

/* K5 — replaces F.softmax + depth_regression + the 4-bin confidence of MVSNet.forward
 *   rmvd/models/mvsnet.py:139-160, rmvd/models/blocks/utils.py:271-274.
 *   cost (B,D,h,w) raw regulariser output; depth_values (B,D);
 *   depth_out (B,h,w) = sum_d softmax(cost)_d * depth_d
 *   conf_out  (B,h,w) = sum_{j=idx-1..idx+2, 0<=j<D} softmax(cost)_j, idx = trunc(sum_d p_d * d)
 * (the reference returns uncertainty = 1 - conf). conf_out may be NULL. */
int mvd_softmax_regress_f32(const float* cost, const float* depth_values, int B, int D, int h, int w,
                            float* depth_out, float* conf_out, mvd_stream_t stream);

/* Measurement aid (bench.py's `measured_stream_peak_gbs`): fills n floats (n % 4 == 0, dst 16-byte aligned) with a pure
 * streaming-store pass, the access pattern an HBM-write-bound kernel can reach at best on this device. */
int mvd_stream_fill_f32(float* dst, long long n, float value, mvd_stream_t stream);

/* Measurement hook (bench.py): arms a pair of hipEvent_t for THIS thread; the next mvd_warp_variance_f32,
 * mvd_homo_warp_f32 or mvd_sweep_corr_f32 call records `start` on its stream immediately before its main kernel
 * (after the small feature re-packing launches) and `stop` immediately after it, then disarms.  Pass NULLs to
 * disarm.  Nothing is synchronised; the caller reads the events after synchronising the stream. */
int mvd_arm_kernel_timing(void* start_event, void* stop_event);

/* Epilogue of the DispNet 2-D convolutions around the Path-A sweep (rmvd/models/blocks/dispnet_encoder.py,
 * dispnet_costvolume_encoder.py, dispnet_decoder.py: Conv2d / ConvTranspose2d with bias followed by LeakyReLU(0.2)):
 *   x[n][c][i] = leaky_relu(x[n][c][i] + bias[c], slope), in place, x (N,C,HW) contiguous.
 * The convolutions themselves stay on the vendor library; this replaces the two elementwise passes after each. */
int mvd_bias_leaky_relu_f32(float* x, const float* bias, int N, int C, long long HW, float slope, mvd_stream_t stream);

/* Other consumers of the sweep as reduction modes of one generic kernel: CVP-MVSNet's variance
 * with PER-PIXEL depth hypotheses (rmvd/models/cvp_mvsnet.py:125-168, blocks/cvp_mvsnet_components.py:375-456) and
 * Vis-MVSNet's group-wise correlation (blocks/utils.py:71-89, blocks/vis_mvsnet_singlestage.py:230-260).
 *   key_feat (B,C,h,w); src_feat[v] (B,C,h,w); M[v] (B,3,4) = [R | t] with (X,Y,Z) = R (x+o, y+o, 1)^T d + t;
 *   depth (B,D), or (B,D,h,w) when depth_per_pixel; sample index = (X/Z) * scale + bias, bilinear, zero padding
 *   (homo_warp / cvp homo_warping: o = 0, scale = W/(W-1), H/(H-1), bias = -0.5; vis homography_warping: o = 0.5, scale = 1,
 *   bias = -0.5);
 *   mode MVD_REDUCE_VARIANCE / MVD_REDUCE_VARIANCE_KEYSQ: out[0] (B,C,D,h,w) — KEYSQ reproduces the reference's aliasing
 *   of the running sum with the squared key volume (cvp_mvsnet.py:129-130);
 *   mode MVD_REDUCE_GROUPCORR: out[v] (B,groups,D,h,w) = sum over each group's channels of key * warped source v. */
#define MVD_REDUCE_VARIANCE 0
#define MVD_REDUCE_VARIANCE_KEYSQ 1
#define MVD_REDUCE_GROUPCORR 2
size_t mvd_sweep_reduce_workspace_bytes(int B, int C, int h, int w, int V);
int mvd_sweep_reduce_f32(const float* key_feat, const float* const* src_feat, const float* const* M, const float* depth,
                         int depth_per_pixel, float pix_offset, float scale_x, float scale_y, float bias, int mode, int groups,
                         int B, int C, int D, int h, int w, int V, float* const* out, void* workspace, size_t workspace_bytes,
                         mvd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the sweep operators w.r.t. the feature maps, for the training loop
 * (rmvd/train/multi_view_depth_training.py:231-246).  The sampling grids carry no gradient (planesweep_corr.py:436,464,489;
 * homo_warp's grid depends on calibration only).  Scatter-adds use float atomics: run-to-run summation order varies.
 * All feature / gradient maps are CHANNEL-LAST; source maps are zero-bordered (.., h+3, w+3, C) with the map at (1,1), as in
 * the forward kernels.  The callee zeroes the gradient outputs it accumulates into.
 * ---------------------------------------------------------------------------------------------- */

/* VJP of mvd_warp_variance_f32.  key_feat, src_feat[v], grad_key, grad_src[v]: (B,h+3,w+3,C) zero-bordered channel-last
 * (gradient border entries = the share of taps that fell on the zero padding; discard them); grad_var (B,D,h,w,C). */
size_t mvd_warp_variance_backward_workspace_bytes(int B);
int mvd_warp_variance_backward_f32(const float* key_feat, const float* const* src_feat, const float* const* src_proj,
                                   const float* key_proj_inv, const float* depth_values, const float* grad_var, int B, int C,
                                   int D, int h, int w, int V, float* grad_key, float* const* grad_src, void* workspace,
                                   size_t workspace_bytes, mvd_stream_t stream);

/* VJP of mvd_sweep_corr_f32 (masks are constants).  feat_key, grad_key (N,h,w,C) channel-last; feat_src[v], grad_src[v]
 * (N,hs+3,ws+3,C) zero-bordered channel-last; grad_corr[v] (N,S,h,w).  C in {64,128,192,256}.  invdepth_mode / corr_scale as
 * in mvd_sweep_corr_ex_f32. */
int mvd_sweep_corr_backward_f32(const float* feat_key, const float* const* feat_src, const float* K_key,
                                const float* const* K_src, const float* const* T_src2key, const float* invdepths,
                                int invdepth_mode, float corr_scale, const float* const* grad_corr, int N, int C, int h, int w,
                                int hs, int ws, int S, int V, float* grad_key, float* const* grad_src, mvd_stream_t stream);

/* VJP of mvd_fuse_views_f32 w.r.t. corr[v] (N,S,h,w) and score[v] (N,1,h,w); masks and the fused mask are constants. */
int mvd_fuse_views_backward_f32(const float* const* corr, const float* const* mask, const float* const* score,
                                const float* grad_fused, int N, int S, int h, int w, int V, float* const* grad_corr,
                                float* const* grad_score, mvd_stream_t stream);

/* Input resize of the model adapters: replaces ResizeInputs / UpscaleInputsToNextMultipleOf
 * (rmvd/data/transforms.py:40-98, called from robust_mvd.py:104-113 and mvsnet.py:178), i.e.
 * skimage.transform.resize(order=1) for UPSCALING (ho >= hi, wo >= wi): no anti-aliasing, float32 kept, mirror boundary,
 * half-pixel centres, float64 interpolation — the arithmetic of scipy.ndimage.zoom(order=1, mode='mirror',
 * grid_mode=True), which skimage delegates to.  src (planes,hi,wi) -> dst (planes,ho,wo); planes = N*C <= 65535. */
int mvd_resize_order1_f32(const float* src, float* dst, long long planes, int hi, int wi, int ho, int wo,
                          mvd_stream_t stream);

/* Prediction head of the DispNet decoder in one pass (rmvd/models/blocks/dispnet_decoder.py:17-22,126-138; ReLUAndSigmoid,
 * blocks/utils.py:30-41 with min -10 / max 10): x (N,2,HW) = output of a pred_k convolution;
 * pred (N,2,HW): channel 0 = relu(x0), channel 1 = sigmoid(x1 * 0.2) * 20 - 10; ent (N,1,HW) = log(2 exp(pred1) + 1e-4) + 1. */
int mvd_dispnet_head_f32(const float* x, float* pred, float* ent, int N, long long HW, mvd_stream_t stream);

/* layout helpers used at the operator-level boundary (reference tensors are NCHW / NCDHW) */
int mvd_nchw_to_nhwc_f32(const float* src, float* dst, int N, int C, long long HW, mvd_stream_t stream);
int mvd_nhwc_to_nchw_f32(const float* src, float* dst, int N, int C, long long HW, mvd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVD_H_ */
