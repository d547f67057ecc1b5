/*
 * include/refign_hip.h -- C ABI of librefign_hip.so, the MI355X (gfx950) native implementation of the
 * Refign align-and-refine hot path.
 *
 * This is the drop-in boundary: plain pointers (DEVICE memory unless stated otherwise), plain ints, and a
 * hipStream_t passed as void*.  No torch / pybind types.  Every entry point
 *   - is asynchronous on `stream` (pass NULL for the legacy default stream, which is what the reference's
 *     CUDA path launches on: correlation_cuda_kernel.cu:271,311,321),
 *   - borrows its inputs, writes only the caller-allocated outputs (the reference allocates with
 *     torch::zeros / zeros_like and returns by value: correlation_cuda_kernel.cu:259,291-292 -- allocation is
 *     the host shim's job, see refign_amd/correlation.py),
 *   - never throws: returns RFN_OK (0) or a negative RFN_E* code; rfn_last_error() gives the message.  The
 *     Python host turns a non-zero code into RuntimeError, which is what TORCH_CHECK produces in the reference
 *     (correlation_sampler.cpp:13-16).
 *
 * Paths cited below are relative to the reference checkout (brdav/refign).
 */
#ifndef REFIGN_HIP_H
#define REFIGN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RFN_OK 0
#define RFN_EINVAL (-1)    /* bad argument (null pointer, non-positive size, unsupported parameterisation) */
#define RFN_ELAUNCH (-2)   /* hipLaunchKernel / runtime error, see rfn_last_error() */
#define RFN_ENOTSUP (-3)   /* valid in the reference but not built here (documented per function) */

/* 2: rfn_global_corr_layer_f32 takes a workspace; rfn_dacs_mix_jitter accepts one half of the mix.
 * 3: rfn_local_corr_layer_split_f32's workspace carries tickets (rfn_local_corr_layer_split_workspace_bytes; zero before first use);
 *    the transpose-cast table holds rfn_multi_transpose_tile() x rfn_multi_transpose_tile() tiles (64; 32 before).
 * 4: new entry points, none changed: rfn_attn32_fwd / rfn_attn32_bwd (fp32 attention), rfn_split3_bf16 / rfn_split3_cat_bf16
 *    (operands of the split-bf16 products), rfn_ffn_fc1_dw_gelu_bf16 (fused Mix-FFN front half). */
#define RFN_ABI_VERSION 4

typedef void* rfn_stream_t; /* hipStream_t */

int rfn_abi_version(void);
const char* rfn_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Spatial correlation sampler -- replaces the pybind module `models.correlation_ops.correlation`
 *   forward : correlation_sampler.cpp:62-90  -> correlation_cuda_forward  (correlation_cuda_kernel.cu:241-278)
 *                                            == correlation_cpp_forward   (correlation.cpp:80-129)
 *   backward: correlation_sampler.cpp:92-127 -> correlation_cuda_backward (correlation_cuda_kernel.cu:280-332)
 *                                            == correlation_cpp_backward  (correlation.cpp:131-183)
 * in1,in2: (B,C,iH,iW) NCHW contiguous.  out / grad_out: (B,patchH,patchW,oH,oW) contiguous with
 *   oH = (iH + 2*padH - ((kH-1)*dilH+1)) / dH + 1   (correlation.cpp:98-99).
 * The 12 ints are the same 12 ints, in the same order, as the reference's forward()/backward().
 * The one parameterisation the hot path uses (kernel 1, patch 9, stride 1, pad 0, dilation 1, dilation_patch 1:
 * modules.py:268-270) takes the LDS-tiled kernel; anything else takes a generic kernel.
 * Patch offsets follow the CPU reference and both backward kernels: shift = (p - (patch - 1) / 2) * dilation_patch
 * (correlation.cpp:28-30).  The reference's CUDA FORWARD computes p * dilation_patch - dilation_patch * (patch - 1) / 2
 * (correlation_cuda_kernel.cu:50-56), which differs from its own CPU path for EVEN patch sizes with dilation_patch > 1
 * (patch 2, dilation 2: {-1, 1} vs {0, 2}); the hot path (patch 9, dilation_patch 1) is not affected, and the oracle
 * (oracle/corr_oracle.c) restates the CPU path.
 * f32 and f64 are provided (the CPU reference dispatches float/double: correlation.cpp:107), and f16 because the CUDA
 * reference additionally dispatches half (correlation_cuda_kernel.cu:267, AT_DISPATCH_FLOATING_TYPES_AND_HALF) -- although
 * the Python wrapper always casts to float32 (correlation_function.py:51) and never reaches it.  f16: IEEE half elements,
 * every product formed and summed in fp32, ONE rounding per result element (forward: generic kernel; backward: a
 * deterministic gather kernel, no 16-bit atomics); any parameterisation.
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_corr_fwd_f32(const float* in1, const float* in2, float* out, int B, int C, int iH, int iW,
                     int kH, int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW,
                     int dpH, int dpW, int dH, int dW, rfn_stream_t stream);
int rfn_corr_fwd_f64(const double* in1, const double* in2, double* out, int B, int C, int iH, int iW,
                     int kH, int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW,
                     int dpH, int dpW, int dH, int dW, rfn_stream_t stream);
int rfn_corr_bwd_f32(const float* in1, const float* in2, const float* grad_out, float* grad_in1,
                     float* grad_in2, int B, int C, int iH, int iW, int kH, int kW, int patchH, int patchW,
                     int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH, int dW,
                     rfn_stream_t stream);
int rfn_corr_bwd_f64(const double* in1, const double* in2, const double* grad_out, double* grad_in1,
                     double* grad_in2, int B, int C, int iH, int iW, int kH, int kW, int patchH, int patchW,
                     int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH, int dW,
                     rfn_stream_t stream);

int rfn_corr_fwd_f16(const void* in1, const void* in2, void* out, int B, int C, int iH, int iW,
                     int kH, int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW,
                     int dpH, int dpW, int dH, int dW, rfn_stream_t stream);
int rfn_corr_bwd_f16(const void* in1, const void* in2, const void* grad_out, void* grad_in1,
                     void* grad_in2, int B, int C, int iH, int iW, int kH, int kW, int patchH, int patchW,
                     int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH, int dW,
                     rfn_stream_t stream);

/* LocalFeatureCorrelationLayer.forward (modules.py:266-274) in ONE kernel:
 *   out[b, ph*9+pw, h, w] = L2-normalise_over_81( relu( sum_c trg[b,c,h,w] * src[b,c,h+ph-4,w+pw-4] ) ), eps 1e-12.
 * feature_target is the sampler's input1 and feature_source its input2 (modules.py:268-269).
 * If `flow` is non-NULL, `src` is the UN-warped source feature map and the bilinear warp of
 * helpers/matching_utils.py:11-49 (align_corners=True, zero padding) is applied on the fly while the source
 * tile is staged in LDS, i.e. warp() -> local_corr() of uawarpc.py:149-152,223-226,251-254 without ever
 * materialising the warped features.  flow: (B,2,H,W), channel 0 = x displacement in pixels of THIS level. */
int rfn_local_corr_layer_f32(const float* feature_target, const float* feature_source, const float* flow,
                             float* out, int B, int C, int H, int W, rfn_stream_t stream);

/* The same for SMALL maps (no flow): the channels are split into `splits` chunks that run as separate workgroups (a
 * 2 x 256 x 32 x 32 level is 8 tiles -- 8 workgroups walking 256 channels serially, ~100 us of latency); partial sums go
 * to `workspace` and are added in chunk order, then ReLU + L2 norm.  ABI 3: ONE launch when the chunks are multiples of 32
 * channels and at least 64 (the tile's last workgroup joins the chunks; tickets live behind the partial volumes), a second
 * kernel otherwise.  workspace: rfn_local_corr_layer_split_workspace_bytes(B, H, W, splits) bytes (splits * B * 81 * H * W floats
 * + one unsigned per 8 x 32 tile -- the tickets: ZERO before the first call with a workspace; every call leaves them zero),
 * 16-byte aligned.  Deterministic; differs from the one-kernel path by the rounding of the
 * chunked channel sum only.  W % 4 == 0, C % splits == 0, (C / splits) % 8 == 0, 2 <= splits <= 64. */
long rfn_local_corr_layer_split_workspace_bytes(int B, int H, int W, int splits);
int rfn_local_corr_layer_split_f32(const float* feature_target, const float* feature_source, float* out,
                                   float* workspace, int B, int C, int H, int W, int splits, rfn_stream_t stream);

/* GlobalFeatureCorrelationLayer.forward (modules.py:294-308): '3D' H-first correlation (modules.py:361-375),
 * mutual matching with eps 1e-5 (modules.py:310-333), ReLU, L2-normalise over the source axis.
 * src: (B,C,Hs,Ws), trg: (B,C,Ht,Wt) -> out: (B,Hs*Ws,Ht,Wt).  Requires Hs*Ws <= 1024 and Ht*Wt <= 1024
 * (the reference asserts 16x16: uawarpc.py:114).  workspace (ABI 2): B*Hs*Ws floats of device scratch for the row maxima of
 * the mutual matching (may be NULL when cyclic_consistency == 0): the scores are normalised in place by one workgroup per 16
 * target positions, so the maxima every workgroup needs are taken by a launch of their own first. */
int rfn_global_corr_layer_f32(const float* feature_source, const float* feature_target, float* out,
                              float* workspace, int B, int C, int Hs, int Ws, int Ht, int Wt,
                              int cyclic_consistency, rfn_stream_t stream);

/* warp() (helpers/matching_utils.py:11-49): bilinear grid_sample, align_corners=True, zero padding.
 * x: (B,C,H,W); flow: (B,2,H,W) pixels; out: (B,C,H,W); mask (nullable): (B,H,W) uint8, 1 where the
 * normalised sampling position is strictly inside (-1,1)^2 (matching_utils.py:46-47).
 * The reference's `if torch.all(flo == 0): return x` early-out (matching_utils.py:19-22) is a host-side
 * decision; this kernel computes the general case, which returns x bit-exactly for zero flow except on the
 * last row/column where the mask differs (handled by the host wrapper, refign_amd/matching.py). */
int rfn_warp_f32(const float* x, const float* flow, float* out, unsigned char* mask, int B, int C, int H,
                 int W, rfn_stream_t stream);
/* Backward of rfn_warp_f32 (grid_sample backward composed with the flow -> grid map): grad_x (B,C,H,W) and / or
 * grad_flow (B,2,H,W), either may be NULL; both are zeroed inside (scatter / cross-channel sums use float atomics).
 * Used by matcher training (models/alignment_model.py:81-146: the head warps source features and the W-bipath loss
 * warps flows with differentiable flows). */
int rfn_warp_bwd_f32(const float* x, const float* flow, const float* grad_out, float* grad_x, float* grad_flow, int B,
                     int C, int H, int W, rfn_stream_t stream);

/* F.interpolate(x, size=(OH,OW), mode='area') of align() (segmentation_model.py:498-501): adaptive average pooling
 * with ATen's window rule [floor(o*I/O), ceil((o+1)*I/O)).  x: (planes,H,W) -> out: (planes,OH,OW). */
int rfn_area_resize_f32(const float* x, float* out, int planes, int H, int W, int OH, int OW, rfn_stream_t stream);

/* F.normalize(x, p=2, dim=1) on NCHW (uawarpc.py:101-108), eps 1e-12. */
int rfn_l2norm_channels_f32(const float* x, float* out, int B, int C, int HW, rfn_stream_t stream);

/* The same (uawarpc.py:101-108) from what the matcher's convolutions deliver under the AMP recipe: x is channels-last
 * 16-bit, (B, HW, C) contiguous, dtype 1 = bfloat16, 2 = float16; out is NCHW float32, (B, C, HW).  The 16-bit values
 * are widened exactly, the norm and the division are fp32 -- what `.float()` -> NCHW copy -> rfn_l2norm_channels_f32
 * computes, in one pass.  C a multiple of 8, <= 2048. */
int rfn_l2norm_channels_nhwc16_f32(const void* x, float* out, int B, int C, int HW, int dtype, rfn_stream_t stream);
/* 2 x 2 / stride 2 max-pool (floor mode) of a channels-last 16-bit map: the pools of the VGG-16 pyramid
 * (models/backbones/vgg.py:33-60: nn.MaxPool2d(2, 2)).  x (B, H, W, C) -> y (B, H / 2, W / 2, C); C % 8 == 0; dtype 1 bf16, 2 f16 */
int rfn_maxpool2x2_nhwc16(const void* x, void* y, int B, int H, int W, int C, int dtype, rfn_stream_t stream);
/* Re-tiling between the layers of the uncertainty head's micro-image chain (models/modules.py:528-545: valid 3x3 convolutions of
 * one s x s correlation patch per pixel) when the patches of an h x w map are held as the (k + 2) x (k + 2) tiles of ONE channels-last
 * image: forward (backward = 0) src (B, (k+2) h - 2, (k+2) w - 2, .) -> dst (B, k h, k w, .), the k x k result of every tile;
 * backward = 1: src = the gradient of that dst, dst = the gradient of that src (zero at the dropped positions).  A pixel is `units16`
 * 16-byte units of any element type (C * element size / 16); consecutive pixels of src are `src_stride16` >= units16 such units apart
 * (a channel-padded convolution result), dst is dense. */
int rfn_retile_copy(const void* src, void* dst, int B, int h, int w, int k, int units16, int src_stride16, int backward,
                    rfn_stream_t stream);

/* refine() + eta() (segmentation_model.py:438-491), gamma = trust-score exponent.
 * logits_trg, logits_ref: (B,19,H,W); warp_mask (nullable): (B,H,W) uint8; certs (nullable): (B,1,H,W);
 * out: (B,19,H,W) refined probabilities (NOT a simplex);
 * workspace: at least rfn_refine_workspace_bytes(B) bytes of device memory (entropy partial sums);
 * flags: bit0 = disable_M, bit1 = disable_P. */
unsigned long rfn_refine_workspace_bytes(int B);
int rfn_refine_f32(const float* logits_trg, const float* logits_ref, const unsigned char* warp_mask,
                   const float* certs, float* out, void* workspace, int B, int C, int H, int W, float gamma,
                   int flags, rfn_stream_t stream);

/* Majority label per scale x scale window (models/segmentation_model.py:637-668, downscale_label_ratio -- the labels the
 * ImageNet feature-distance loss is masked with): gt (B,H,W) int64 -> out (B, ceil(H/scale), ceil(W/scale)) int64 = the
 * most frequent class of the window (smallest index among equals), or ignore_index when that is the ignore label or its
 * share of the window is below min_ratio.  Edge windows are partial (share = count / pixels inside the image).
 * n_classes <= 32. */
int rfn_label_majority(const long* gt, long* out, int B, int H, int W, int scale, int n_classes, int ignore_index,
                       float min_ratio, rfn_stream_t stream);

/* Tail of align() (segmentation_model.py:514-522) fused: bilinear (align_corners=False) upsampling of the
 * quarter-resolution flow (B,2,h,w) and log-variance (B,1,h,w) to (H,W), confidence
 * P_R = 1 - exp(-1/(2 exp(logvar))) (matching_utils.py:52-57), and warp of logits_ref (B,C,H,W) with the
 * upsampled flow.  Outputs: warped (B,C,H,W), mask (B,H,W) uint8, cert (B,1,H,W), and (nullable) flow_up
 * (B,2,H,W). */
int rfn_align_tail_f32(const float* logits_ref, const float* flow_q, const float* logvar_q, float* warped,
                       unsigned char* mask, float* cert, float* flow_up, int B, int C, int H, int W, int h,
                       int w, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Depthwise 3x3 convolution on channels-last maps -- the DWConv of every Mix-FFN block
 * (models/backbones/mix_transformer.py:96-103,556-568: nn.Conv2d(dim, dim, 3, 1, 1, groups=dim) between two token
 * transposes) and the dilated depthwise branches of the DAFormer ASPP (models/heads/daformer.py:46-62).
 * x, y, grad_y: (B,H,W,C) contiguous == the (B, N, C) token layout; dtype 0 = float32, 1 = bfloat16.
 * weight / grad_weight: TAP-MAJOR (9, C) float32 (= conv.weight.view(C, 9).t()); bias / grad_bias: (C) float32,
 * nullable.  Zero padding = dilation (same-size output).  flip = 1 applies the taps mirrored (that IS the
 * backward-data pass: grad_x = fwd(grad_y, weight, NULL, flip = 1)).  bwd_weight needs
 * rfn_dwconv3x3_bwd_weight_workspace_bytes(C) bytes of device workspace (per-stripe partial sums, reduced in a fixed
 * order: deterministic).  bwd_weight `flags`: bit 0 = ACCUMULATE into grad_weight / grad_bias (they are the
 * parameters' .grad views of the flat gradient buffer: no separate add kernel); bit 1 = grad_weight in the parameter's
 * own (C,1,3,3) layout instead of tap-major.
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_dwconv3x3_nhwc_fwd(const void* x, const float* weight, const float* bias, void* y, int B, int H, int W,
                           int C, int dilation, int dtype, int flip, rfn_stream_t stream);
/* the same (bf16 only) + the BatchNorm statistics of the result in `sums` (2 C + 1 doubles: the buffer of rfn_bn_stats_fwd,
 * zeroed inside): depthwise 3x3 -> BN of the DAFormer ASPP branches (daformer.py:10-62) without the statistics pass */
int rfn_dwconv3x3_nhwc_fwd_stats(const void* x, const float* weight, const float* bias, void* y, double* sums, int B, int H,
                                 int W, int C, int dilation, int dtype, rfn_stream_t stream);
/* gradient-free depthwise 3x3 -> BatchNorm(batch statistics) -> ReLU (the EMA teacher's ASPP branches) in two
 * passes over the INPUT: statistics of the (rounded) convolution result without storing it, then convolution + normalisation
 * + activation in one pass -- 3 tensor passes instead of 5.  A SyncBatchNorm all-reduces `sums` in between. */
int rfn_dwconv3x3_nhwc_stats(const void* x, const float* weight, const float* bias, double* sums, int B, int H, int W, int C,
                             int dilation, int dtype, rfn_stream_t stream);
int rfn_dwconv3x3_bn_act_nhwc_fwd(const void* x, const float* weight, const float* bias, const float* gamma, const float* beta,
                                  const double* sums, float* running_mean, float* running_var, void* y, int B, int H, int W,
                                  int C, int dilation, float eps, float momentum, int relu, int dtype, rfn_stream_t stream);

/* Three dilated depthwise 3x3 branches (dilations g, 2 g, 3 g; padding = dilation) of ONE bf16 NHWC input, one pass each
 * (round 4; the EMA teacher's ASPP, daformer.py:46-62,65-126: dilations 6 / 12 / 18 of the concatenated 1024-channel map).  A
 * workgroup holds one g x g phase sub-image of one image for 64 channels in LDS and computes the three branches from it: every
 * input byte is read once per pass instead of 4.5 times per branch.
 *   rfn_dwconv3x3_tri_usable      1 if (B, H, W, C, g) is inside the kernel's domain: C % 64 == 0, g <= min(H, W),
 *                                 ceil(H / g) ceil(W / g) <= 944 pixels per phase
 *   rfn_dwconv3x3_tri_stats       sums3 [3][2 C + 1] doubles <- (sum, sum of squares, rows) of the three ROUNDED results (the
 *                                 buffers of rfn_bn_stats_fwd, zeroed here); nothing stored
 *   rfn_dwconv3x3_tri_bn_act_fwd  y3[k] = act(bn_k(conv_k(x))) with batch statistics sums3; running buffers updated as
 *                                 rfn_bn_apply_fwd does.  weight3 [3][9][C] tap-major fp32, bias3 [3][C] or NULL; gamma3 / beta3 /
 *                                 running_mean3 / running_var3 / y3 / eps3 / momentum3: HOST arrays of 3 (entries of the first four
 *                                 may be NULL), read at call time. */
int rfn_dwconv3x3_tri_usable(int B, int H, int W, int C, int g);
int rfn_dwconv3x3_tri_stats(const void* x, const float* weight3, const float* bias3, double* sums3, int B, int H, int W, int C, int g,
                            rfn_stream_t stream);
int rfn_dwconv3x3_tri_bn_act_fwd(const void* x, const float* weight3, const float* bias3, const float* const* gamma3,
                                 const float* const* beta3, const double* sums3, float* const* running_mean3,
                                 float* const* running_var3, void* const* y3, int B, int H, int W, int C, int g, const float* eps3,
                                 const float* momentum3, int relu, rfn_stream_t stream);
/* The same convolution (dilation 1) followed by GELU (exact erf) -- the DWConv + act of the Mix-FFN
 * (mix_transformer.py:99-101) in one pass: y_act = gelu(conv(x) + bias); y_pre (may be NULL) = the pre-activation,
 * which the backward of GELU needs and a gradient-free pass does not. */
int rfn_dwconv3x3_gelu_nhwc_fwd(const void* x, const float* weight, const float* bias, void* y_pre, void* y_act, int B,
                                int H, int W, int C, int dtype, rfn_stream_t stream);
unsigned long rfn_dwconv3x3_bwd_weight_workspace_bytes(int C);
int rfn_dwconv3x3_nhwc_bwd_weight(const void* x, const void* grad_y, float* grad_weight, float* grad_bias,
                                  void* workspace, int B, int H, int W, int C, int dilation, int dtype, int flags,
                                  rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * LayerNorm over the last dim of a (rows, C) matrix, C <= 1024 -- the 213 LayerNorms of a MiT-B5 forward
 * (models/backbones/mix_transformer.py:135,188-207,234,369-419; eps 1e-6 in blocks/stage norms, 1e-5 in
 * patch-embed and spatial-reduction norms).  Statistics in fp32; activations float32 (dtype code 0) or bfloat16 (1),
 * independently for input and output, so the residual stream can stay in bf16 with no separate cast passes.
 * gamma, beta, grad_gamma, grad_beta: (C) float32; mean, rstd: (rows) float32 saved by fwd for bwd.
 * bwd: grad_x has x's dtype; needs rfn_layernorm_bwd_workspace_bytes(C) bytes of workspace (deterministic two-stage
 * reduction of the parameter gradients); accumulate != 0 adds them to grad_gamma / grad_beta instead of overwriting.
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      long rows, int C, float eps, int in_dtype, int out_dtype, rfn_stream_t stream);
unsigned long rfn_layernorm_bwd_workspace_bytes(int C);
int rfn_layernorm_bwd(const void* x, const void* grad_y, const float* gamma, const float* mean, const float* rstd,
                      void* grad_x, float* grad_gamma, float* grad_beta, void* workspace, long rows, int C,
                      int x_dtype, int gy_dtype, int accumulate, rfn_stream_t stream);
/* rfn_layernorm_bwd with a second gradient of x folded in: grad_x = LayerNorm-backward(grad_y) + add (`add` has x's dtype and
 * shape; NULL = plain backward).  The residual stream of a MiT block feeds the LayerNorm and the residual add
 * (mix_transformer.py:203-207): this is the sum autograd would otherwise make in an element-wise kernel.  C % 8 == 0. */
int rfn_layernorm_bwd_add(const void* x, const void* grad_y, const void* add, const float* gamma, const float* mean,
                          const float* rstd, void* grad_x, float* grad_gamma, float* grad_beta, void* workspace, long rows, int C,
                          int x_dtype, int gy_dtype, int accumulate, rfn_stream_t stream);
/* the same with a SECOND gradient of the LayerNorm output summed in (either of grad_y2 / add may be null): in a MiT attention block
   norm1(x) feeds the q projection and the spatial-reduction convolution (mix_transformer.py:142-150) */
int rfn_layernorm_bwd_add2(const void* x, const void* grad_y, const void* grad_y2, const void* add, const float* gamma,
                           const float* mean, const float* rstd, void* grad_x, float* grad_gamma, float* grad_beta, void* workspace,
                           long rows, int C, int x_dtype, int gy_dtype, int accumulate, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Front end of the UAWarpC UncertaintyModule for search size 9 (models/modules.py:529-551, eval mode): every pixel's
 * 9x9 correlation patch -> conv3x3(1->32)+BN+LeakyReLU(0.1) -> conv3x3(32->32)+BN+LeakyReLU -> conv3x3(32->16)+BN+
 * LeakyReLU -> conv3x3(16->6), all VALID, i.e. 9x9 -> 7x7 -> 5x5 -> 3x3 -> 1x1; one fused kernel, fp32 MFMA.
 * corr (B,81,H,W), out (B,6,H,W) float32 contiguous.  weights: rfn_uncertainty9_weights_len() floats, BatchNorm
 * folded in (w' = w*gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps)), packed as
 *   W0[tap9][c32] b0[32] | W1[k288][n32] b1[32] | W2[k288][n16] b2[16] | W3[k144][c6] b3[6]
 * with k = (ky*3+kx)*Cin + ci, i.e. W[k][n] = conv.weight[n][ci][ky][kx].
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_uncertainty9_weights_len(void);
int rfn_uncertainty9_frontend_f32(const float* corr, const float* weights, float* out, int B, int H, int W,
                                  rfn_stream_t stream);
/* The same with the two matrix layers (32 -> 32, 32 -> 16) on the f16 matrix pipe, fp32 accumulation, f16 activations in between --
 * the precision the reference's AMP recipe runs these convolutions in (autocast; modules.py:529-551 forces nothing).  Same packed
 * fp32 weights, fp32 input and result.  (ABI 3.) */
int rfn_uncertainty9_frontend_f16mm(const float* corr, const float* weights, float* out, int B, int H, int W,
                                    rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * out[i] (+)= sum_{s < S} x[s*n + i]: sum over the leading dimension of a row-major (S, n) matrix into float32.
 * (Replaces what autograd's backward of nn.Linear launches for the layers of models/backbones/mix_transformer.py:47-78,
 * 100-186 and models/heads/daformer.py, segformer.py.)  Two uses on the training step, both formerly a library reduction followed by a separate gradient-accumulation add:
 *   - bias gradient of every token-wise Linear: x = grad_y (tokens, features), S = tokens (8 160 ... 259 200)
 *   - reduction of the split-T weight-gradient partials (refign_amd/linear.py): x = (S <= 64, N*K)
 * x dtype 0 = float32, 1 = bfloat16; n must be a multiple of 8.  accumulate != 0: out += sum (out is the parameter's
 * .grad view).  Tall inputs (S > 64) go through per-stripe partial sums in `workspace`
 * (rfn_sum_rows_workspace_bytes(S, n) bytes; may be NULL when that is 0) reduced in a fixed order: deterministic.
 * ---------------------------------------------------------------------------------------------------------- */
unsigned long rfn_sum_rows_workspace_bytes(long S, long n);
int rfn_sum_rows(const void* x, float* out, void* workspace, long S, long n, int x_dtype, int accumulate,
                 rfn_stream_t stream);
/* Both of the above for one Linear in two launches instead of three: grad_bias (N) (+)= column sum of grad_y (T, N),
 * T > 64, workspace = rfn_sum_rows_workspace_bytes(T, N) bytes; grad_weight (N*K) (+)= sum over the S <= 64 slabs of
 * w_partials (S, N*K).  grad_y and w_partials share `dtype`. */
int rfn_linear_param_grads(const void* grad_y, float* grad_bias, void* workspace, long T, long N, int acc_bias,
                           const void* w_partials, float* grad_weight, int S, long NK, int acc_weight, int dtype,
                           rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Multi-tensor cast float32 -> bfloat16 (round to nearest even) in one launch: refresh of the cached bf16 copies of
 * all parameters after an optimizer step / EMA update -- the per-use weight casts torch.autocast performs under the
 * reference's `--trainer.precision 16` recipe (README.md:262), done once per update instead.  `table`: DEVICE array of nchunks entries
 * { const float* src; uint16_t* dst; long n; } (24 bytes each), one per chunk of at most
 * rfn_multi_cast_chunk_elems() elements; the host splits every tensor into such chunks once per parameter set.
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_multi_cast_chunk_elems(void);
/* layout-changing refresh of cached weight copies in one launch: every row of `table` (12 int64: src fp32 pointer, dst
 * pointer, extents of dst dims 1..3, source strides of dims 0..3 in elements, first dst element and element count of the
 * chunk, dst dtype 0 fp32 / 1 bf16 / 2 f16) copies <= rfn_multi_permute_chunk_elems() consecutive elements of a contiguous
 * <= 4-D dst from a strided view of an fp32 parameter (autocast's / the kernels' permuted weight forms after an optimiser step) */
int rfn_multi_permute_chunk_elems(void);
int rfn_multi_permute_cast_f32(const void* table, int nchunks, rfn_stream_t stream);
int rfn_multi_cast_f32_bf16(const void* table, int nchunks, rfn_stream_t stream);
/* EMA-teacher update of a whole parameter set in ONE launch (models/segmentation_model.py:676-689):
 * ema <- momentum * ema + (1 - momentum) * live, fp32 in place; table = nchunks x {float* ema, const float* live,
 * bf16* copy_or_NULL, long n} in device memory (chunks of at most rfn_multi_cast_chunk_elems() elements); where a chunk
 * has a bf16 copy pointer, the rounded new value is written there in the same pass (the teacher's cached 16-bit weight). */
int rfn_multi_ema_f32(const void* table, int nchunks, float momentum, rfn_stream_t stream);
/* (Same purpose as rfn_multi_cast_f32_bf16.)  Transposed bf16 copies of a set of fp32 matrices in ONE launch: dst (K, N) = bf16(src (N, K)^T), both row-major and
 * contiguous; table = ntiles x {const float* src, bf16* dst, int N, int K, int n0, int k0} (one T x T tile each, T = rfn_multi_transpose_tile(): 64
 * since ABI 3, 32 before) in device memory.  (The cached W^T operands of the input-gradient GEMMs, refreshed after optimiser / EMA updates.) */
int rfn_multi_transpose_cast_f32_bf16(const void* table, int ntiles, rfn_stream_t stream);
int rfn_multi_transpose_tile(void);
/* AdamW step of a whole parameter set in ONE launch (what the reference's optimizer section instantiates --
 * configs/cityscapes_darkzurich/refign_hrda_star.yaml:176-180, stepped by models/segmentation_model.py:252 --
 * torch.optim.AdamW, decoupled weight decay, no amsgrad / maximize; fp32 state).  table = nchunks x {float* p,
 * const float* grad, float* exp_avg, float* exp_avg_sq, long n | group << 56} in DEVICE memory; group_args = HOST array of
 * ngroups (<= 8) x {lr, beta1, beta2, eps, weight_decay, 1 - beta1^t, sqrt(1 - beta2^t), 1 - beta1, 1 - beta2} for this
 * step (the derived values computed in double by the host, as torch does). */
int rfn_multi_adamw_f32(const void* table, int nchunks, const float* group_args, int ngroups, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Multi-resolution fusion front end of the decode heads (DAFormerHead.forward, models/heads/daformer.py:205-222;
 * SegFormerHead.forward, models/heads/segformer.py:86-104): bilinear up-sampling (align_corners=False) of up to four
 * embedded stage maps to (H, W) and their channel concatenation, one pass, channels-last output (n, H, W, sum C_l).
 * src_l: TOKEN maps (n, hs[l]*ws[l], cs[l]) contiguous (= channels-last), cs[l] % 8 == 0; a level that already has
 * (H, W) is copied.  hs / ws / cs: HOST arrays of nlev ints.  dtype 0 = float32, 1 = bfloat16 (fp32 blend).
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_upsample_concat_nhwc(const void* src0, const void* src1, const void* src2, const void* src3, const int* hs,
                             const int* ws, const int* cs, int nlev, void* out, int n, int H, int W, int dtype,
                             rfn_stream_t stream);
/* Its backward (what autograd runs for F.interpolate + torch.cat of daformer.py:205-222 / segformer.py:86-104):
 * grad_out (n, H, W, sum C_l) channels-last -> grad_l (n, hs[l]*ws[l], cs[l]) for every level, in one pass; the
 * bilinear weights of the forward, fp32 accumulation.  Deterministic (a gather, no atomics). */
int rfn_upsample_concat_nhwc_bwd(const void* grad_out, void* grad0, void* grad1, void* grad2, void* grad3, const int* hs,
                                 const int* ws, const int* cs, int nlev, int n, int H, int W, int dtype,
                                 rfn_stream_t stream);

/* Token map (B, H*W, C) -> non-overlapping r x r patches (B*(H/r)*(W/r), r*r*C) with (ry, rx, c) fastest (inverse = 0),
 * or back (inverse = 1; a ragged border of the token map is NOT written -- zero it first): gather / scatter around the
 * spatial-reduction convolution of the MiT attention run as a Linear (mix_transformer.py:133-146).  C % 8 == 0. */
int rfn_patchify_tokens(const void* src, void* dst, int B, int H, int W, int C, int r, int dtype, int inverse,
                        rfn_stream_t stream);

/* The same with the patch row in (c, ry, rx) order -- the (C, r, r) layout of one output channel of the convolution weight
 * (mix_transformer.py:133-136 `self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)`): the patch GEMM multiplies
 * by the parameter's own (Co, C r r) matrix and its weight gradient lands in the parameter's layout.  r in {2, 4, 8}. */
int rfn_patchify_tokens_cmajor(const void* src, void* dst, int B, int H, int W, int C, int r, int dtype, int inverse,
                               rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Hand-written matrix-core (MFMA 32x32x16) GEMMs of a token-wise nn.Linear -- q / kv / proj / fc1 / fc2 / sr-as-Linear
 * of MiT (mix_transformer.py:96-103,137-164) and the MLP embeds of the decode heads (daformer.py:129-149).
 * dtype 1 = bfloat16, 2 = float16 operands and result, fp32 accumulation.  Row-major, leading dimensions in elements.
 *
 * rfn_gemm_nt:  Y[M,N] = res + rowscale[m / rows_per_sample] * act( X[M,K] . W[N,K]^T + bias[N] )   (res, rowscale, bias: NULL = absent)
 *   forward: X = tokens, W = weight.  dgrad: X = grad_y [T,N], W = weight^T [K,N] (host keeps the transposed copy).
 *   bias / res / rowscale may be NULL (res NULL: Y = act(...); rowscale needs res: the stochastic-depth residual
 *   `x + drop_path(branch)` of mix_transformer.py:203-207 with per-sample keep masks).  act: 0 none, 1 ReLU, 3 LeakyReLU(0.1);
 *   4: Y = (rowscale * (X W^T + b)) * gelu'(res) -- `res` is the pre-activation of an exact-erf GELU in front of this layer's
 *   input, Y the gradient with respect to it (input-gradient GEMM of the Mix-FFN's fc2, mix_transformer.py:99-102).
 *   K % 64 == 0, N % 8 == 0, ldx / ldw / ldy % 8 == 0.
 * rfn_gemm_tn:  sum over rows t of slab s of rowscale[t / rows_per_sample] * G[t,n] * X[t,k] (rowscale may be NULL), S = ceil(T / rows_per_slab) slabs computed by separate
 *   workgroups (the reduction of a weight gradient is the TOKEN dimension: 8 160 ... 259 200 rows for a <= 2048 x 2048
 *   result).  wgrad: G = grad_y [T,N], X = tokens [T,K].
 *     accumulate = 0: P[s][N,K] = fp32 partial of slab s (deterministic; caller reduces);
 *     accumulate = 1: P[N,K] += every slab, fp32 atomics (the parameter's view of the flat gradient buffer);
 *     grad_bias (may be NULL): [N] fp32, += column sums of G (the bias gradient), fp32 atomics.
 *   N % 64 == 0, K % 64 == 0, rows_per_slab % 32 == 0.
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_gemm_nt(const void* X, const void* W, const void* bias, const void* res, const float* rowscale,
                int rows_per_sample, int act, void* Y, long M, long N, long K, long ldx, long ldw, long ldy, int dtype,
                rfn_stream_t stream);
/* Convolution as an implicit GEMM on the same kernel: channels-last activations X (B, H, W, C), weights W[n][(ky, kx, c)]
 * with rows zero-padded to ldw >= roundup(KH*KW*C, 64), output Y (B, OH, OW, N) with row stride ldy (so a layer can write
 * its channel slice of a concatenation buffer).  Y = res + act(conv(X) + bias); act 0 none, 1 ReLU, 3 LeakyReLU(0.1).
 * Replaces F.conv2d for VGG-16 (vgg.py:108-120), the flow decoders / refinement modules (modules.py:395-477), the MiT patch
 * embeddings (mix_transformer.py:210-242) and the decode-head convolutions (daformer.py:65-126) -- eval-mode BatchNorm is
 * folded into W / bias by the host.  C % 8 == 0, N % 8 == 0, zero padding, any stride / dilation / kernel size. */
int rfn_conv2d_nhwc(const void* X, const void* W, const void* bias, const void* res, int act, void* Y, int B, int H,
                    int Wd, int C, int N, int KH, int KW, int stride, int pad, int dil, long ldw, long ldy, int dtype,
                    rfn_stream_t stream);
int rfn_gemm_tn(const void* G, const void* X, float* P, long T, long N, long K, long ldg, long ldx, int rows_per_slab,
                int accumulate, float* grad_bias, const float* rowscale, int rows_per_sample, int dtype,
                rfn_stream_t stream);
/* Up to 8 weight gradients of the accumulate form (P[i] += (diag(rowscale[i]) G[i])^T X[i], grad_bias[i] += column sums; fp32
 * atomics into the parameters' gradient views) in ONE launch of 64 x 64 tiles (round 5): the weight gradients of a MiT block's
 * Linear layers are off the backward pass's dependency chain, the host queues them and hands a block's worth over at once
 * (refign_amd/linear.py).  Arrays of `count` entries; grad_bias[i] / rowscale[i] may be NULL, but a group is all-with or
 * all-without a row scale; operands 16-byte aligned, ldg / ldx % 8 == 0, N / K % 64 == 0, rows_per_slab % 32 == 0. */
int rfn_gemm_tn_grouped(int count, const void* const* G, const void* const* X, float* const* P, float* const* grad_bias,
                        const float* const* rowscale, const long* T, const long* N, const long* K, const long* ldg,
                        const long* ldx, const int* rows_per_slab, const int* rows_per_sample, int dtype, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Hand-written matrix-core attention for MiT's efficient self-attention (mix_transformer.py:137-164):
 * softmax(scale q k^T) v with head_dim 64, long query / short key sequences.  dtype 1 = bf16, 2 = f16; fp32 softmax.
 * Tensors are addressed as element (b, row, head, d) at base + b*batch_stride + row*row_stride + head*64 + d, i.e. the
 * (B, N, heads*64) output of the q Linear and the K / V halves of the (B, Nkv, 2, heads, 64) kv Linear output are used
 * in place.
 *   rfn_attn_pack     rows of one tensor -> R-pack and/or T-pack (MFMA A-operand images, 4096 B per 32-row block and
 *                     (b, head); nblk blocks, zero padded).  Needed: K R+T, V R+T (forward uses K R, V T), Q R+T, dO R+T.
 *                     A second tensor of the same geometry (src2 -> rpack2 / tpack2) rides in the same launch.  K and V
 *                     of a kv tensor are packed in ONE call as 2 x heads "heads" (V's head h is pack head heads + h):
 *                     the kernels take the pack pointers of K / V and kv_pack_heads = 2 x heads.
 *   rfn_attn_fwd      O and lse2[b*heads+h][nqpad] (base-2 log-sum-exp of the scaled scores), nkblk even.
 *   rfn_attn_bwd_dq   dQ, and delta[bh][nqpad] = rowsum(dO o O) for the dK/dV kernel.
 *   rfn_attn_bwd_dkv  dKV (B, Nkv, 2, heads, 64); accT: fp32 scratch of B*heads*2*64*nkpad floats; the query dimension is
 *                     split into chunks of `blocks_per_chunk` 32-row blocks per workgroup.
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_attn_pack(const void* src, long batch_stride, long row_stride, int B, int heads, int nrows, int nblk, void* rpack,
                  void* tpack, const void* src2, void* rpack2, void* tpack2, rfn_stream_t stream);
int rfn_attn_fwd(const void* Q, long q_batch_stride, long q_row_stride, const void* k_rpack, const void* v_tpack, void* O,
                 long o_batch_stride, long o_row_stride, float* lse2, int B, int heads, int Nq, int Nkv, int nkblk,
                 int nqpad, float scale, int kv_pack_heads, int dtype, rfn_stream_t stream);
int rfn_attn_bwd_dq(const void* Q, long q_batch_stride, long q_row_stride, const void* dO, const void* O,
                    long o_batch_stride, long o_row_stride, const void* k_rpack, const void* v_rpack, const void* k_tpack,
                    const float* lse2, float* delta, void* dQ, long dq_batch_stride, long dq_row_stride, int B, int heads,
                    int Nq, int Nkv, int nkblk, int nqpad, float scale, int kv_pack_heads, int dtype,
                    rfn_stream_t stream);
int rfn_attn_bwd_dkv(const void* K, const void* V, long kv_batch_stride, long kv_row_stride, const void* q_rpack,
                     const void* q_tpack, const void* do_rpack, const void* do_tpack, const float* lse2,
                     const float* delta, float* accT, void* dKV, int B, int heads, int Nq, int Nkv, int nqblk, int nqpad,
                     int nkpad, int blocks_per_chunk, float scale, int dtype, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * The same attention core for FLOAT32 tensors (the fp32 parity mode; mix_transformer.py:150-160 materialises the score
 * matrix in fp32): every product on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32), softmax in fp32, one launch per pass.
 * Q (B, Nq, heads*D) and KV (B, Nkv, 2, heads, D), head_dim D = 64 or 32 (MiT-B0), are used in place through their strides
 * (in floats, multiples of 4; 16-byte aligned bases); lse2 / delta: fp32 [B*heads][nqpad].
 *   rfn_attn32_fwd   O and lse2 (base-2 log-sum-exp of the scaled scores).
 *   rfn_attn32_bwd   dQ, dKV and delta = rowsum(dO o O), two launches.  dK / dV: the query dimension is split into
 *                    `query_chunks` chunks per 128-key tile whose partial sums are ADDED with fp32 atomics -- dKV must be
 *                    ZERO on entry when query_chunks > 1 (one chunk: plain stores, any contents).
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_attn32_fwd(const float* Q, long q_batch_stride, long q_row_stride, const float* KV, long kv_batch_stride,
                   long kv_row_stride, float* O, long o_batch_stride, long o_row_stride, float* lse2, int B, int heads,
                   int head_dim, int Nq, int Nkv, int nqpad, float scale, rfn_stream_t stream);
int rfn_attn32_bwd(const float* Q, long q_batch_stride, long q_row_stride, const float* KV, long kv_batch_stride,
                   long kv_row_stride, const float* dO, const float* O, long o_batch_stride, long o_row_stride,
                   const float* lse2, float* delta, float* dQ, long dq_batch_stride, long dq_row_stride, float* dKV,
                   long dkv_batch_stride, long dkv_row_stride, int B, int heads, int head_dim, int Nq, int Nkv, int nqpad,
                   int query_chunks, float scale, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Operand preparation of the split-bf16 products (an fp32 product as three bf16 MFMA products, refign_amd/split32.py):
 * x (rows, K) fp32 with row stride x_row_stride -> hi = bf16(x), lo = bf16(x - hi), written as three terms
 *   term i, row r, column k -> out[i * term_stride + r * out_row_stride + k]  (bf16; columns K .. Kp-1 zero; Kp % 4 == 0)
 * order 0: (hi, hi, lo) -- activations / gradients; order 1: (hi, lo, hi) -- weights.  term_stride = Kp with
 * out_row_stride = 3 Kp puts the terms side by side along the reduction index; term_stride = rows * out_row_stride stacks them.
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_split3_bf16(const float* x, long x_row_stride, void* out, long out_row_stride, long term_stride, long rows, int K,
                    int Kp, int order, rfn_stream_t stream);
/* The (hi, hi, lo) split of the channel CONCATENATION of 1..4 fp32 maps of one (B, H, W), each addressed through its own
 * strides[4 i .. 4 i + 3] = (batch, channel, row, column) in floats (NCHW and channels-last parts mix freely), written as the
 * (B, H, W, 3 Cp) bf16 operand of rfn_conv2d_nhwc_o32; sum of channels <= Cp <= 96, Cp % 4 == 0 (uawarpc.py:136-160: the decoder
 * inputs cat(correlation, flow, ...)).  `parts`, `strides`, `channels` are HOST arrays. */
int rfn_split3_cat_bf16(const float* const* parts, const long* strides, const int* channels, int nparts, void* out, int B, int H,
                        int W, int Cp, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Mix-FFN front half of the gradient-free MiT passes (mix_transformer.py:79-103) in one kernel:
 *   a = gelu(dwconv3x3(x W1^T + b1) + bdw)      x (views, H*W, C) bf16 tokens, a (views, H*W, HID) bf16
 * W1 (HID, C) bf16 row-major, b1 (HID) bf16, wdw_tap (9, HID) fp32 tap-major (tap = ky * 3 + kx), bdw (HID) fp32;
 * C % 64 == 0, HID % 128 == 0.  The hidden pre-activation is rounded to bf16 (as the fc1 kernel stores it) and is zero
 * outside the image (the convolution pads the hidden map).
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_ffn_fc1_dw_gelu_bf16(const void* x, const void* w1, const void* b1, const float* wdw_tap, const float* bdw, void* a,
                             int views, int H, int W, int C, int HID, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Training-mode BatchNorm2d (+ ReLU) on channels-last 16-bit tensors viewed as (T = B*H*W, C): the norm + activation of
 * the decode heads' ConvBNReLU blocks (models/modules.py:16-56), which use BATCH statistics in the student and in the EMA
 * teacher.  dtype 1 = bf16, 2 = f16; statistics, affine parameters and running buffers fp32; C % 8 == 0.
 * `relu`: activation after the affine map, 0 none, 1 ReLU, 3 LeakyReLU(0.1) (the codes of rfn_gemm_nt; the flow decoders
 * of the matcher use LeakyReLU, models/modules.py:395-477).
 * Statistics buffer `sums`: 2 C + 1 DOUBLES = (sum x, sum x^2, number of rows) -- fp64 from the first addition, so that the
 * one-pass variance E[x^2] - mean^2 does not cancel (csrc/bn.hip header); the apply passes normalise with the row
 * count they find THERE, so a SUM all-reduce of the buffer between a stats pass and an apply pass turns batch statistics
 * into cross-replica ones -- SyncBatchNorm (torch/nn/modules/_functions.py:SyncBatchNorm; the reference trains with
 * `sync_batchnorm: True`).  `bwd_sums`: 2 C floats.
 *   rfn_bn_stats_fwd  sums <- (sum_t x, sum_t x^2, T)  [zeroed inside]
 *   rfn_bn_apply_fwd  y = relu?((x - mean) rstd gamma + beta); running_mean / running_var (may be NULL) <-
 *                     (1 - momentum) old + momentum (mean, unbiased var)
 *   rfn_bn_stats_bwd  bwd_sums <- (sum g', sum g' xhat) = the LOCAL (grad beta, grad gamma), g' = g masked by the ReLU
 *   rfn_bn_apply_bwd  grad_x = gamma rstd (g' - (bwd_sums[0] + xhat bwd_sums[1]) / rows)
 *   rfn_bn_train_fwd / _bwd   one replica: the two passes back to back
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_bn_stats_fwd(const void* x, double* sums, long T, int C, int dtype, rfn_stream_t stream);
int rfn_bn_apply_fwd(const void* x, const float* gamma, const float* beta, void* y, const double* sums, float* running_mean,
                     float* running_var, long T, int C, float eps, float momentum, int relu, int dtype, rfn_stream_t stream);
/* rfn_bn_apply_fwd with y at a row pitch of ld_y elements: y = a channel slice of a wider channels-last tensor (the ASPP
 * branches write into their concatenation, daformer.py:110-118) */
int rfn_bn_apply_fwd_ld(const void* x, const float* gamma, const float* beta, void* y, long ld_y, const double* sums,
                        float* running_mean, float* running_var, long T, int C, float eps, float momentum, int relu, int dtype,
                        rfn_stream_t stream);
int rfn_bn_stats_bwd(const void* x, const void* grad_y, const double* fwd_sums, const float* gamma, const float* beta,
                     float* bwd_sums, long T, int C, float eps, int relu, int dtype, rfn_stream_t stream);
int rfn_bn_apply_bwd(const void* x, const void* grad_y, const double* fwd_sums, const float* bwd_sums, const float* gamma,
                     const float* beta, void* grad_x, long T, int C, float eps, int relu, int dtype, rfn_stream_t stream);
int rfn_bn_train_fwd(const void* x, const float* gamma, const float* beta, void* y, double* sums, float* running_mean,
                     float* running_var, long T, int C, float eps, float momentum, int relu, int dtype,
                     rfn_stream_t stream);
int rfn_bn_train_bwd(const void* x, const void* grad_y, const double* fwd_sums, const float* gamma, const float* beta,
                     void* grad_x, float* bwd_sums, long T, int C, float eps, int relu, int dtype, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Segmentation loss of the student passes in one kernel (csrc/loss.hip): bilinear up-sampling (align_corners = False) of
 * the class logits (B, C, h, w) to the label size (H, W) + pixel-weighted cross-entropy with ignore_index, summed over all
 * pixels -- models/segmentation_model.py:163-179, :226-250 (F.interpolate) + models/losses.py:10-22
 * (PixelWeightedCrossEntropyLoss).  loss_sum[0..63] <- 64 partial sums (the host adds them and divides by B H W: the
 * reference's mean over ALL pixels; one address for 8 000 workgroups would serialise in the L2); grad_lo (B, C, h, w) fp32 <- d loss_sum / d logits.  Both are zeroed inside.  dtype of logits: 0 fp32, 1 bf16,
 * 2 f16; target int64 (B, H, W); weight fp32 (B, H, W) or NULL; round16: round the interpolated logits to `dtype` (what an
 * unfused 16-bit F.interpolate stores).  C <= 19, H >= 2 h, W >= 2 w.
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_upsample_ce(const void* logits, const long* target, const float* weight, float* grad_lo, double* loss_sum, int B,
                    int C, int h, int w, int H, int W, int ignore_index, int dtype, int round16, rfn_stream_t stream);

/* fp32-RESULT variants (split-bf16 parity mode, refign_amd/split32.py): bf16 operands whose reduction index carries the
 * three split products side by side, fp32 accumulate, fp32 bias / residual / result (leading dimension ldy in floats).
 * rfn_conv2d_nhwc_o32: (B, H, W, C) = the convolution's input side, N output channels; transposed = 0: Y (B, OH, OW, N) from
 * X (B, H, W, C), W[n][(tap, c)]; transposed = 1: the data gradient Y (B, H, W, C) from X = grad_y (B, OH, OW, N), W[c][(tap, n)]. */
int rfn_gemm_nt_o32(const void* X, const void* W, const float* bias, const float* res, const float* rowscale,
                    int rows_per_sample, int act, float* Y, long M, long N, long K, long ldx, long ldw, long ldy,
                    rfn_stream_t stream);
int rfn_conv2d_nhwc_o32(const void* X, const void* W, const float* bias, int act, float* Y, int B, int H, int Wd, int C, int N,
                        int KH, int KW, int stride, int pad, int dil, long ldw, long ldy, int transposed,
                        rfn_stream_t stream);
/* Backward of rfn_conv2d_nhwc on the same kernels (the student's trainable convolutions: DAFormer 3x3 bottleneck
 * daformer.py:65-126, MiT overlap patch embeddings mix_transformer.py:210-242, 19-class 1x1; matcher decoders modules.py:395-477).
 *   rfn_conv2d_nhwc_dgrad  DX (B, H, W, C) = data gradient.  GY (B, OH, OW, N) channels-last, Wt[c][(ky, kx, n)] rows zero-padded
 *                          to ldw >= roundup(KH*KW*N, 64); the implicit-GEMM kernel in transposed-gather mode (no flipped copy of
 *                          the filter, no col2im buffer).  stride must be a power of two.  C % 8 == 0, N % 8 == 0.
 *   rfn_conv2d_nhwc_wgrad  P = weight gradient in the PACKED layout [n][(ky, kx, c)], row length Kpad (% 64, >= KH*KW*C), fp32:
 *                          the split-T kernel of rfn_gemm_tn with the im2col rows gathered on the fly.  accumulate = 0: one
 *                          (N, Kpad) partial per slab of rows_per_slab output pixels; 1: atomics into one.  grad_bias (may be
 *                          NULL) += column sums of GY.  N % 64 == 0, C % 2 == 0, ldg = row stride of GY in elements. */
int rfn_conv2d_nhwc_dgrad(const void* GY, const void* Wt, void* DX, int B, int H, int W, int C, int N, int KH, int KW,
                          int stride, int pad, int dil, long ldw, long ldy, int dtype, rfn_stream_t stream);
int rfn_conv2d_nhwc_wgrad(const void* GY, const void* X, float* P, float* grad_bias, int B, int H, int W, int C, int N, int KH,
                          int KW, int stride, int pad, int dil, long ldg, long Kpad, int rows_per_slab, int accumulate,
                          int dtype, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * N4 -- GPU-side data step of the UDA iteration (csrc/dacs.hip): DACS class-mix + colour jitter + Gaussian blur of
 * helpers/dacs_transforms.py:14-24,43-78,81-112 as called per sample by models/segmentation_model.py:525-582.  Images
 * (B, 3, H, W) fp32 ImageNet-normalised, labels (B, H, W) int64, weights (B, H, W) fp32; B <= 8, H*W % 4 == 0.
 *   rfn_dacs_mix_jitter  mixed = mask ? source : target for image / label / weight (source weight 1), mask = label in the class
 *                        set class_bits[n] (DEVICE int64: bit c = class c, bit 31 = label 255); then, where jitter_on[n], the
 *                        colour-jitter chain on the de-normalised image: order[4n + k] = operator applied k-th (0 brightness
 *                        additive, 1 contrast about the image mean, 2 saturation about the luma, 3 hue = 3x3 matrix hue[9n..]),
 *                        factor[4n + op], clamp to [0, 1] after every operator.  mean_ws: 8 doubles of device scratch.  Host
 *                        arrays are read at call time (they travel as kernel arguments).  Either half may be left out (round 5:
 *                        only the mixed LABEL depends on the teacher's pseudo-labels, segmentation_model.py:558-574, so the
 *                        student's forward on the mixed image starts before they exist): mixed_img == NULL -> labels and weights
 *                        only (src / trg unused), mixed_lbl == NULL -> image only (pseudo_label / pseudo_weight / mixed_weight
 *                        unused); both halves cut with the same mask when gt_src / class_bits are the same.
 *   rfn_dacs_blur        separable Gaussian of ksize_y x ksize_x taps (odd; kornia: ~0.1 x the image extent) normalised over the
 *                        window, reflect border, where blur_on[n]; tmp = scratch like x.  sigma <= 1.25: taps beyond +-16
 *                        vanish in fp32 for kornia's sigma range 0.15 ... 1.15, so at most 33 taps are evaluated.
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_dacs_mix_jitter(const float* src, const float* trg, const long* gt_src, const long* pseudo_label,
                        const float* pseudo_weight, float* mixed_img, long* mixed_lbl, float* mixed_weight, double* mean_ws,
                        int B, int H, int W, const long* class_bits, const int* jitter_on, const int* order,
                        const float* factor, const float* hue, const float* mean3, const float* std3,
                        rfn_stream_t stream);
int rfn_dacs_blur(const float* x, float* tmp, float* y, int B, int C, int H, int W, int ksize_y, int ksize_x, const int* blur_on,
                  const double* sigma_y, const double* sigma_x, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * N4, second part -- source / target SAMPLING on the device (csrc/datastep.hip): the crop re-draws of rare-class sampling
 * (data_modules/datasets/cityscapes.py:139-158) and of RandomCrop's category ratio (data_modules/transforms.py:282-361),
 * RandomHorizontalFlip (:363-390), ConvertImageDtype (:438-464), Normalize (:467-495).  uint8 images (C, H, W) and label maps
 * (H, W) as the reference's ToTensor leaves them (:250-279).
 *   rfn_crop_label_hist_u8  hist[k][v] = number of pixels with label v in crop box k = (top, left, h, w) = boxes[4k..] (HOST
 *                           array, read at call time), K <= 16 candidate boxes in one launch; hist: K x 256 DEVICE ints (zeroed
 *                           by the call).
 *   rfn_crop_flip_norm_u8   out_image (C, h, w) fp32 = (u8 / 255 - mean[c]) / std[c] of the crop, mirrored along x when flip;
 *                           out_label (h, w) int64.  Either half may be NULL.  mean3 / std3: HOST arrays.
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_crop_label_hist_u8(const void* label, int H, int W, const int* boxes, int K, int* hist, rfn_stream_t stream);
int rfn_crop_flip_norm_u8(const void* image, const void* label, int C, int H, int W, int top, int left, int h, int w, int flip,
                          const float* mean3, const float* std3, float* out_image, long* out_label, rfn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * K5 (BASELINE.json config 5, "bf16 HRDA + fp8 MFMA attention"): fp8 (OCP e4m3, fp32 accumulate) matrix-core path of
 * the gradient-free EMA teacher (segmentation_model.py:204-209 runs MiT-B5, mix_transformer.py:96-103,137-164, on 40
 * HRDA views per GPU).  No reference analogue (the reference's recipe is 16-bit AMP, README.md:262); csrc/f8.hip.
 * Scales: `x_scale` / `q_scale` / ... = what a stored e4m3 byte is multiplied by to give the value; `out_q` = what a value
 * is multiplied by before it is stored as e4m3 (saturating at +-448, round to nearest even).
 *   rfn_gemm_nt_f8      Y[M,N] = res + rowscale[m / rows_per_sample] * act( x_scale * wscale[n] * (X8[M,K] . W8[N,K]^T) + bias[n] )
 *                       X8, W8 e4m3 bytes (ldx, ldw in bytes, % 16); wscale fp32 [N]; bias / res bf16 or NULL;
 *                       out_f8 = 0: Y bf16 (ldy in elements); out_f8 = 1: Y e4m3 = act(...) * out_q (ldy in bytes; no res).
 *                       K % 16 == 0 (need not be a multiple of the 128-byte K-step), N % 16 == 0.  act: 0 none, 1 ReLU.
 *   rfn_quant_rows_f8   multi-tensor weight quantisation.  table: nchunks rows of 4 int64 {bf16 src row 0, e4m3 dst row 0,
 *                       fp32 scales, K | nrows << 32} (nrows <= 4 consecutive rows of K % 4 == 0 elements): dst = rne(src /
 *                       scale), scale = amax(row) / 448.
 *   rfn_quant_f8        y8[i] = e4m3(x_bf16[i] * q), n % 4 == 0 (entry of an fp8 chain, tests).
 *   rfn_layernorm_fwd_f8          LayerNorm over C of bf16 rows, result * out_q stored as e4m3 (C % 8 == 0, C <= 1024).
 *   rfn_dwconv3x3_gelu_nhwc_fwd_f8  GELU(depthwise 3x3 + bias) of the Mix-FFN on e4m3 channels-last maps, weights (9, C) fp32.
 *   rfn_attn_pack_f8    K / V of an e4m3 kv tensor (B, Nkv, 2 * heads * 64; strides in bytes) -> nst = ceil(Nkv / 64) stages of
 *                       8 192 bytes per (batch, head) in MFMA operand order (zero padded).
 *   rfn_attn_fwd_f8     O8 = softmax(scale Q K^T) V per head of 64, Q8 / O8 (B, Nq, heads * 64) e4m3, fp32 softmax, the
 *                       probabilities enter the P.V product as e4m3(256 p).
 * ---------------------------------------------------------------------------------------------------------- */
int rfn_gemm_nt_f8(const void* X8, const void* W8, const float* wscale, float x_scale, const void* bias, const void* res,
                   const float* rowscale, int rows_per_sample, int act, void* Y, int out_f8, float out_q, long M, long N,
                   long K, long ldx, long ldw, long ldy, rfn_stream_t stream);
int rfn_quant_rows_f8(const void* table, int nchunks, rfn_stream_t stream);
int rfn_quant_f8(const void* x_bf16, void* y8, long n, float q, rfn_stream_t stream);
int rfn_layernorm_fwd_f8(const void* x_bf16, const float* gamma, const float* beta, void* y8, long rows, int C, float eps,
                         float out_q, rfn_stream_t stream);
int rfn_dwconv3x3_gelu_nhwc_fwd_f8(const void* x8, const float* weight, const float* bias, void* y8, int B, int H, int W, int C,
                                   float x_scale, float out_q, rfn_stream_t stream);
int rfn_attn_pack_f8(const void* kv8, long batch_stride, long row_stride, int B, int heads, int Nkv, int nst, void* pack,
                     rfn_stream_t stream);
int rfn_attn_fwd_f8(const void* q8, long q_batch_stride, long q_row_stride, const void* pack, void* o8, long o_batch_stride,
                    long o_row_stride, int B, int heads, int Nq, int Nkv, int nst, float scale, float q_scale, float k_scale,
                    float v_scale, float out_q, rfn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* REFIGN_HIP_H */
