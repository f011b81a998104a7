/* erfnet_mi355.h -- C ABI of libermi355.so: hand-written gfx950 (MI355X) kernels for the
 * ERFNet forward/backward hot path.
 *
 * The reference (pandongwei/erfnet_pytorch) is pure Python on torch.nn layers and has no
 * FFI of its own; each entry point below names the reference code whose arithmetic it
 * replaces (paths relative to the reference checkout).  The Python package
 * erfnet_pytorch_amd binds these with ctypes and exposes them behind the reference's
 * nn.Module / iouEval / criterion surface.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless a comment says "host";
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work on it, never
 *     synchronise, never allocate, never retain pointers beyond the call;
 *   - return value: 0 ok, -1 invalid argument, -2 unsupported shape/dtype, -3 launch failure,
 *     -4 size beyond a fixed limit (only the *_static entry point);
 *   - dtype: ERF_F32 (0) or ERF_BF16 (1) = element type of activations and packed weights;
 *     accumulation is always fp32; parameters (weights, scale, shift, bias) are fp32;
 *   - activations are NHWC (channels of a pixel contiguous) except where "NCHW" is stated:
 *     the network input image and the logits keep the reference's NCHW layout;
 *   - "packed" weights are MFMA operand fragments produced by the matching erf_*_pack call
 *     into a caller-owned buffer of erf_*_packed_bytes bytes (16-byte aligned).
 */
#ifndef ERFNET_MI355_H
#define ERFNET_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ERF_F32 0
#define ERF_BF16 1
/* image sources only (x_dtype of erf_stem_downsampler / erf_image_to_nhwc_padded): 8-bit images as decoders deliver them,
 * converted on load to value / 255 in fp32, i.e. ToTensor() of the reference's input transform (train/main.py:47,66;
 * eval/eval_iou.py:31).  ERF_U8_NCHW: planar [N,3,H,W]; ERF_U8_NHWC: interleaved [N,H,W,3]. */
#define ERF_U8_NCHW 2
#define ERF_U8_NHWC 3

#define ERF_AXIS_H 0 /* 3x1 kernel: taps along rows    */
#define ERF_AXIS_W 1 /* 1x3 kernel: taps along columns */

/* Library/ABI version (major * 100 + minor). */
int erf_version(void);

/* ---- factorized 3x1 / 1x3 (dilated) convolution, C -> C ------------------------------
 * Replaces nn.Conv2d(chann, chann, (3,1)|(1,3), padding=(d,0)|(0,d), dilation=(d,1)|(1,d))
 * of non_bottleneck_1d, train/erfnet.py:43-51, with the ops that follow it in
 * forward (:58-74) fused into the epilogue:
 *     y = act( scale[c] * conv(x)[c] + shift[c] (+ residual) )
 * scale == NULL means 1, shift carries the conv bias (folded with eval-mode BatchNorm when
 * one follows), residual == NULL means none, relu != 0 applies max(.,0) last.
 * w: fp32 [C][C][3] (= Conv2d weight (C,C,3,1) or (C,C,1,3), contiguous).
 * mode 0: forward weights.  mode 1: data-gradient weights (ci<->co swapped, taps reversed),
 * which makes the same kernel compute dL/dx from dL/dy. */
size_t erf_conv1d_packed_bytes(int dtype, int C);
int erf_conv1d_pack(const float* w, void* packed, int dtype, int C, int mode, void* stream);
/* erf_conv1d_pack for `count` weights in one launch (a training step re-packs all 68 factorized convolutions,
 * forward and data-gradient form, after every optimiser update).  table: device array of count x 4 int64
 * {fp32 weight address, packed buffer address, C, mode}; max_c = largest C in the table (16, 64 or 128). */
int erf_conv1d_pack_many(const long long* table, int count, int max_c, int dtype, void* stream);
int erf_conv1d_factorized(const void* x, void* y, const void* packed, const float* scale, const float* shift,
                          const void* residual, int dtype, int N, int H, int W, int C, int axis, int dilation,
                          int relu, void* stream);

/* One half of non_bottleneck_1d.forward in a single launch (inference; train/erfnet.py:58-62 or
 * :64-74 with BatchNorm in eval mode folded into scale_b/shift_b):
 *     y = act( scale_b * conv1x3_d( relu( scale_a * conv3x1_d(x) + shift_a ) ) + shift_b (+ residual) )
 * packed_a / packed_b: mode-0 packed weights of the 3x1 and the 1x3 convolution.  bf16 only.
 * Returns ERF_EUNSUPPORTED (-2) without launching for shapes it does not take (fp32, or a dilation
 * that leaves too few pixels per residue class -- except C = 128, dilation 16 on maps of at most 64 x 128, which a kernel
 * working on whole image rows takes): issue two erf_conv1d_factorized calls then. */
int erf_conv1d_pair(const void* x, void* y, const void* packed_a, const void* packed_b, const float* scale_a,
                    const float* shift_a, const float* scale_b, const float* shift_b, const void* residual, int dtype,
                    int N, int H, int W, int C, int dilation, int relu, void* stream);

/* The whole of non_bottleneck_1d.forward in one launch (inference; train/erfnet.py:58-74 with both BatchNorms in eval mode
 * folded into scale / shift, Dropout2d the identity):
 *     a1 = relu(conv3x1(x) + t1); a2 = relu(s2 * conv1x3(a1) + t2); a3 = relu(conv3x1_d(a2) + t3);
 *     y  = relu(s4 * conv1x3_d(a3) + t4 + x)
 * x is read once and y written once; the three intermediate maps stay on the chip.  Results are bit-identical to two
 * erf_conv1d_pair calls.  packed1..4: mode-0 packed weights of the four convolutions; table: fp32 [4][2][C] in device memory,
 * row k = {scale of convolution k (1 where it has none), shift of convolution k}.  x and y must not alias.
 * Takes bf16, C = 16 or C = 64, dilation 1; returns ERF_EUNSUPPORTED (-2) without launching otherwise: issue two
 * erf_conv1d_pair.  The result is bit-identical to those two launches.  On MI355X the C = 16 kernel is the faster way
 * (45 vs 70 us at 16 x 256 x 512); the C = 64 kernel currently takes longer than the two launches (one wave per SIMD cannot
 * hide its own epilogues), so prefer that one only where the number of launches matters. */
int erf_nb1d_block_infer(const void* x, void* y, const void* packed1, const void* packed2, const void* packed3, const void* packed4,
                         const float* table, int dtype, int N, int H, int W, int C, int dilation, void* stream);

/* ---- DownsamplerBlock, train/erfnet.py:11-22 ------------------------------------------
 * y[..., :Cout-Cin] = conv3x3 s2 p1 (x), y[..., Cout-Cin:] = maxpool2x2 (x), then
 * y = act(scale * y + shift) over all Cout channels (concat happens on write).
 * Cin in {16, 64}: x is NHWC [N,Hin,Win,Cin]; w fp32 [Cout-Cin][Cin][3][3]. */
size_t erf_downsampler_packed_bytes(int dtype, int Cin, int Cout);
int erf_downsampler_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream);
int erf_downsampler(const void* x, void* y, const void* packed, const float* scale, const float* shift, int dtype,
                    int N, int Hin, int Win, int Cin, int Cout, int relu, void* stream);
/* The 3 -> 16 initial_block (train/erfnet.py:80): x is the NCHW network input
 * [N,3,Hin,Win] of type x_dtype (ERF_F32 / ERF_BF16, or one of the two uint8 image sources above, for which the kernel
 * applies ToTensor()'s / 255 on load); w fp32 [13][3][3][3] unpacked; y NHWC [N,Hin/2,Win/2,16]. */
int erf_stem_downsampler(const void* x_nchw, int x_dtype, void* y, const float* w, const float* scale,
                         const float* shift, int dtype, int N, int Hin, int Win, int relu, void* stream);

/* ---- UpsamplerBlock, train/erfnet.py:146-155 ------------------------------------------
 * ConvTranspose2d(Cin, Cout, 3, stride 2, padding 1, output_padding 1) as four output-parity
 * phases; y = act(scale * convT(x) + shift), y NHWC [N,2Hin,2Win,Cout].
 * w fp32 [Cin][Cout][3][3] (torch ConvTranspose2d layout). */
size_t erf_upsampler_packed_bytes(int dtype, int Cin, int Cout);
int erf_upsampler_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream);
int erf_upsampler_convT3x3s2(const void* x, void* y, const void* packed, const float* scale, const float* shift,
                             int dtype, int N, int Hin, int Win, int Cin, int Cout, int relu, void* stream);

/* ---- Decoder.output_conv, train/erfnet.py:171,179 -------------------------------------
 * ConvTranspose2d(16, nc, 2, stride 2): each input pixel produces its own 2x2 output patch.
 * logits (may be NULL): NCHW [N,nc,2Hin,2Win], fp32 when logits_f32 != 0, else `dtype`.
 * labels (may be NULL): per-pixel argmax over the nc classes, first maximum wins
 * (outputs.max(1)[1], eval/eval_iou.py:93), [N,2Hin,2Win] as uint8 (label_bytes 1) or
 * int64 (label_bytes 8), taken from the fp32 accumulators -- the logits need not be written.
 * w fp32 [16][nc][2][2], bias fp32 [nc]; nc <= 32. */
size_t erf_output_conv_packed_bytes(int dtype, int nc);
int erf_output_conv_pack(const float* w, void* packed, int dtype, int nc, void* stream);
int erf_output_convT2x2s2(const void* x, const void* packed, const float* bias, void* logits, int logits_f32,
                          void* labels, int label_bytes, int dtype, int N, int Hin, int Win, int nc, void* stream);

/* ---- Encoder.output_conv (1x1, only_encode path), train/erfnet.py:98,106-107 ----------
 * y NCHW [N,nc,H,W] (fp32 when out_f32), x NHWC [N,H,W,Cin]; w fp32 [nc][Cin]; nc <= 32. */
size_t erf_pointwise_packed_bytes(int dtype, int Cin, int nc);
int erf_pointwise_pack(const float* w, void* packed, int dtype, int Cin, int nc, void* stream);
int erf_pointwise_conv(const void* x, void* y, const void* packed, const float* bias, int out_f32, int dtype,
                       int N, int H, int W, int Cin, int nc, void* stream);

/* ---- layout glue ------------------------------------------------------------------------
 * NCHW <-> NHWC copies with dtype conversion (module boundaries of the nn.Module API). */
int erf_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C, int H, int W, void* stream);
int erf_nhwc_to_nchw(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C, int H, int W, void* stream);

/* ---- metrics ----------------------------------------------------------------------------
 * erf_argmax_c: outputs.max(1)[1] on NCHW logits (eval/eval_iou.py:93, train/main.py:212,275).
 * erf_iou_hist_accumulate: the counting part of iouEval.addBatch (train/iouEval.py:21-69):
 *   hist[label * nc + pred] += 1 for every pixel, int64 [nc*nc + 1]; pixels whose label or
 *   prediction lies outside [0,nc) are counted in hist[nc*nc] instead.  tp/fp/fn follow as
 *   tp_c = hist[c][c], fn_c = sum_p hist[c][p] - tp_c, fp_c = sum_g hist[g][c] - tp_c - hist[ignore][c].
 *   pred_bytes / label_bytes: 1 (uint8) or 8 (int64). */
int erf_argmax_c(const void* logits, int dtype, void* labels, int label_bytes, int N, int C, long long HW, void* stream);
int erf_iou_hist_accumulate(const void* pred, int pred_bytes, const void* label, int label_bytes, long long count,
                            int nc, long long* hist, void* stream);
/* The label side of the loaders' target transform on the device (eval/eval_iou.py:33-37 ToLabel(), Relabel(255, 19);
 * train/main.py:65-66; train/transform.py:53-68), so that a loader can hand over the decoded 8-bit label images:
 * erf_iou_hist_accumulate_relabel counts as erf_iou_hist_accumulate does after `label[label == olabel] = nlabel`
 *   (0 <= olabel, 0 <= nlabel < nc), without writing the relabelled map;
 * erf_labels_from_u8 writes dst[i] = (src[i] == olabel ? nlabel : src[i]) as int64 -- ToLabel().long() + Relabel in one pass,
 *   the targets of the loss (olabel < 0: conversion only).  src and dst 16-byte aligned. */
int erf_iou_hist_accumulate_relabel(const void* pred, int pred_bytes, const void* label, int label_bytes, long long count,
                                    int nc, int olabel, int nlabel, long long* hist, void* stream);
int erf_labels_from_u8(const void* src, long long* dst, long long count, int olabel, int nlabel, void* stream);
/* iouEval.addBatch when an input already has nClasses planes (train/iouEval.py:32-46 `else: x_onehot = x.float()`): the
 * reference then multiplies the planes as they are (one-hot or soft scores), :60-65.  sums [3][classes] fp64 (tp, fp, fn;
 * classes = ignore_index when ignore_index >= 0, else nc) is ADDED to:
 *   tp_c += sum x_c*y_c, fp_c += sum x_c*(1 - y_c - y_ignore), fn_c += sum (1 - x_c)*y_c.
 * x_kind / y_kind: 0 = fp32 planes NCHW [N,nc,HW], 1 = int64 label map [N,HW], 2 = uint8 label map (expanded on the fly).
 * Products are formed in fp32 like the reference's, sums in fp64 in a fixed order (the reference sums in fp32: equal for
 * one-hot inputs, which give integers). workspace: erf_iou_planes_workspace_bytes(nc). */
size_t erf_iou_planes_workspace_bytes(int nc);
int erf_iou_planes_accumulate(const void* x, int x_kind, const void* y, int y_kind, int N, int nc, long long HW, int ignore_index,
                              double* sums, void* workspace, void* stream);

/* =========================== training step =============================================== */

/* ---- BatchNorm2d(C, eps=1e-3) in training mode, train/erfnet.py:17,47,53,150 ------------
 * erf_bn_stats_finalize: per-channel batch mean / biased variance of z (NHWC, `pixels` = N*H*W),
 *   mean/invstd for the backward pass, the folded affine scale = gamma*invstd,
 *   shift = beta - mean*scale, and the running-statistics update
 *   running = (1-momentum)*running + momentum*(mean, unbiased var) (skipped when running_mean NULL).
 * erf_bn_apply: y = act(((z - center)*scale + shift) * mask[n][c] + residual).  With center = mean,
 *   scale = gamma*invstd, shift = beta this is torch's (z - mean)*invstd*gamma + beta without the
 *   cancellation of the folded form; center NULL means 0.  mask (Dropout2d multipliers [N][C],
 *   train/erfnet.py:71-72) and residual (:74) may be NULL.
 * erf_bn_backward: with dy = g * [gate > 0] * mask (ReLU and Dropout2d backward folded in; gate and
 *   mask may be NULL): dgamma = sum dy*xhat, dbeta = sum dy,
 *   dz = gamma*invstd*(dy - dbeta/M - xhat*dgamma/M)  (batch_stats = 1), or gamma*invstd*dy (= 0,
 *   eval-mode statistics).  g_out (may be NULL) receives g * [gate > 0], the gradient that flows into
 *   the residual branch.
 * workspace: erf_bn_workspace_bytes(C) bytes, contents undefined on entry. */
size_t erf_bn_workspace_bytes(int C);
int erf_bn_stats_finalize(const void* z, int dtype, long long pixels, int C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                          float* scale, float* shift, void* workspace, void* stream);
int erf_bn_apply(const void* z, void* y, const float* center, const float* scale, const float* shift, const void* residual,
                 const float* mask, int dtype, int N, long long HW, int C, int relu, void* stream);
/* erf_bn_apply that also writes the ReLU gate of its output (what F.relu's backward needs of it, train/erfnet.py:63,74): one
 * byte per 16-byte chunk of y, bit e = [y value e > 0] (bf16: 8 channels per byte, fp32: 4; N*HW*C*sizeof(T)/16 bytes),
 * taken from the stored, rounded values.  C a multiple of the chunk.  erf_conv1d_dgrad_bn_backward_bits reads it in place of
 * the tensor: a sixteenth of the bytes. */
int erf_bn_apply_gates(const void* z, void* y, const float* center, const float* scale, const float* shift, const void* residual,
                       const float* mask, int dtype, int N, long long HW, int C, int relu, void* gate_out, void* stream);
int erf_bn_backward(const void* g, const void* gate, const float* mask, const void* z, const float* mean,
                    const float* invstd, const float* gamma, int batch_stats, void* dz, void* g_out, float* dgamma,
                    float* dbeta, int dtype, int N, long long HW, int C, void* workspace, void* stream);
/* out = g * [gate > 0]: F.relu backward between the two convs of a factorized pair (train/erfnet.py:61,67). */
int erf_relu_backward(const void* g, const void* gate, void* out, int dtype, long long count, void* stream);
/* sums[c] = sum_p z[p][c]: conv bias gradients.  workspace as for BatchNorm. */
int erf_channel_sum(const void* z, int dtype, long long pixels, int C, float* sums, void* workspace, void* stream);
/* nn.Dropout2d(p) multipliers (train/erfnet.py:55): mask[i] = Bernoulli(1-p)/(1-p), i < count = N*C,
 * a counter-based hash of (seed, i). */
int erf_dropout2d_mask(float* mask, int count, float p, unsigned long long seed, void* stream);
/* The masks of up to 32 Dropout2d layers in one launch (every non_bottleneck_1d of a forward pass): segment s of `mask`
 * has counts[s] elements and holds exactly what erf_dropout2d_mask(., counts[s], p[s], seeds[s]) writes.  counts / p /
 * seeds are host arrays. */
int erf_dropout2d_mask_many(float* mask, int segments, const int* counts, const float* p, const unsigned long long* seeds,
                            void* stream);
/* The same masks with the seeds kept in device memory, for steps replayed from a captured graph: state[0] = base,
 * state[1] = masks drawn so far; segment s gets seed = state[0] + (state[1] + s + 1) * 0xD1B54A32D192ED03 (mod 2^64) and
 * state[1] then advances by `segments` (a second, one-thread launch).  counts / p are host arrays. */
int erf_dropout2d_mask_many_counter(float* mask, int segments, const int* counts, const float* p, unsigned long long* state,
                                    void* stream);
/* MaxPool2d(2, stride 2) backward of the DownsamplerBlock (train/erfnet.py:16,20): routes
 * dz[..., zoff:zoff+Cin] (channel stride ldz) to the first maximum of each window of x;
 * accumulate != 0 adds to dx (after the conv part's data gradient), else overwrites. */
int erf_maxpool2x2_backward(const void* x, const void* dz, void* dx, int dtype, int N, int Hin, int Win, int Cin, int ldz,
                            int zoff, int accumulate, void* stream);

/* ---- loss: torch.nn.CrossEntropyLoss(weight), train/main.py:121,202 (= CrossEntropyLoss2d :72-80)
 * fwd: out3 = { sum_p w[t_p]*nll_p, sum_p w[t_p], their quotient }; a target of -100 (torch's ignore_index) carries no
 *   weight; any other target outside [0,C) -- on which torch raises -- gives NaN sums and NaN gradients (loud, and without a
 *   host round trip).  The two sums are returned separately so that data-parallel ranks can divide by the
 *   GLOBAL denominator, as DataParallel's loss over the gathered batch does.
 * bwd: dlogits = coef[0] * w[t] * (softmax - onehot), coef a DEVICE scalar (grad_output / denominator).
 * logits / dlogits NCHW of `dtype`; target int64 [N,H,W]; weight fp32 [C] or NULL. */
size_t erf_ce_workspace_bytes(void);
int erf_weighted_ce_fwd(const void* logits, int dtype, const long long* target, const float* weight, int N, int C,
                        long long HW, float* out3, void* workspace, void* stream);
int erf_weighted_ce_bwd(const void* logits, int dtype, const long long* target, const float* weight, int N, int C,
                        long long HW, const float* coef, void* dlogits, void* stream);

/* ---- optimiser: torch.optim.Adam(lr, betas, eps, weight_decay) of train/main.py:140 and
 * torch.optim.SGD(lr, momentum, weight_decay) of imagenet/main.py:83-85, over flat fp32 arenas.
 * g = grad * grad_scale + weight_decay * p (coupled L2); step is 1-based.  Hyper-parameters are
 * doubles (as Python floats are): 1 - beta and the bias corrections are formed in double first. */
int erf_adam_fused_arena(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, double lr,
                         double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream);
/* Adam with its step count and learning rate in device memory (a captured graph replays the step without host values):
 * erf_adam_advance: *step += 1; hyper[0] = lr / (1 - beta1^step), hyper[1] = sqrt(1 - beta2^step), formed in double.
 * erf_adam_fused_arena_hyper: erf_adam_fused_arena's update with those two factors read from hyper. */
int erf_adam_advance(int* step, const double* lr, float* hyper, double beta1, double beta2, void* stream);
int erf_adam_fused_arena_hyper(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, const float* hyper,
                               double beta1, double beta2, double eps, double weight_decay, double grad_scale, void* stream);
int erf_sgd_fused_arena(float* params, const float* grads, float* momentum_buf, long long n, double lr, double momentum,
                        double weight_decay, int first_step, double grad_scale, void* stream);
/* The same update with the learning rate and the weight decay read from device memory (hyper[0], hyper[1]): one captured
 * step serves every value of imagenet/main.py:276-286's schedule (lr * 0.95^(epoch-15), weight decay dropped to 0). */
int erf_sgd_fused_arena_hyper(float* params, const float* grads, float* momentum_buf, long long n, const float* hyper,
                              double momentum, int first_step, double grad_scale, void* stream);

/* ---- batched weight re-packing (no counterpart in the reference: torch re-reads the fp32 parameters in place) ----
 * A training step re-packs every weight after the optimiser update.  Between erf_pack_record_begin(host_table,
 * capacity) and erf_pack_record_end the *_pack calls of the calling thread launch nothing: each appends one or more
 * descriptors (erf_pack_desc_bytes() bytes each, holding the source and destination pointers passed) to host_table.
 * erf_pack_record_end returns the number of descriptors (or a negative status) and the grid extent / dtype to pass on.
 * The caller copies the table to device memory once; erf_pack_replay then redoes all those re-packs in one launch. */
size_t erf_pack_desc_bytes(void);
int erf_pack_record_begin(void* host_table, int capacity);
int erf_pack_record_end(long long* max_chunks, int* dtype);
int erf_pack_replay(const void* table_dev, int count, long long max_chunks, int dtype, void* stream);

/* ---- Multi_models (train/multi_models.py), inference -------------------------------------------------------
 * The multi-task network reuses non_bottleneck_1d and UpsamplerBlock (also 128 -> 16, multi_models.py:127) and adds:
 * erf_conv3x3: 3x3 convolution, padding 1, stride 1 or 2, y = act(scale * conv(x) + shift), NHWC in and out
 *   (DownsamplerBlock_simplified.conv, multi_models.py:27, stride 1; Classifier conv, :190, stride 2).  Weights in
 *   nn.Conv2d layout [Cout][Cin_valid][3][3]; x carries Cin >= Cin_valid channels (the image is widened to 16 by
 *   erf_image_to_nhwc_channels).  Shapes: (16,16) (16,64) (64,128) (128,64).
 * erf_maxpool2x2_affine: y = act(scale * maxpool2x2(x) + shift) -- pool, BatchNorm (folded) and ReLU of
 *   DownsamplerBlock_simplified.forward (multi_models.py:31-36) in one pass.
 * erf_convT2x2s2: ConvTranspose2d(32, Cout, 2, stride 2) + bias -> NCHW planes (the decoders' last layer,
 *   multi_models.py:134 / :165), weights [32][Cout][2][2].
 * erf_spatial_mean: feat[n][c] = mean over H x W (Classifier.forward `output.mean(3).mean(2)`, multi_models.py:200).
 * erf_concat_channels: out[n][h][w][:] = a[n][h][w][:] ++ b[n][h][w][:] -- `torch.cat([output, skip], 1)` of the decoders
 *   (multi_models.py:141,145) on NHWC maps; Ca and Cb multiples of 8 (bf16) / 4 (fp32), else ERF_EUNSUPPORTED. */
size_t erf_conv3x3_packed_bytes(int dtype, int Cin, int Cout);
int erf_conv3x3_pack(const float* w, void* packed, int dtype, int Cin, int Cin_valid, int Cout, void* stream);
int erf_conv3x3(const void* x, void* y, const void* packed, const float* scale, const float* shift, int dtype, int N, int Hin,
                int Win, int Cin, int Cout, int stride, int relu, void* stream);
int erf_maxpool2x2_affine(const void* x, void* y, const float* scale, const float* shift, int dtype, int N, int Hin, int Win,
                          int C, int relu, void* stream);
size_t erf_convT2x2s2_packed_bytes(int dtype, int Cin, int Cout);
int erf_convT2x2s2_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream);
int erf_convT2x2s2(const void* x, void* y, const void* packed, const float* bias, int out_f32, int dtype, int N, int Hin, int Win,
                   int Cin, int Cout, void* stream);
int erf_spatial_mean(const void* x, int dtype, float* feat, int N, int H, int W, int C, void* stream);
int erf_concat_channels(const void* a, int Ca, const void* b, int Cb, void* out, int dtype, int N, int H, int W, void* stream);
int erf_image_to_nhwc_channels(const void* x_nchw, int x_dtype, void* out, int dtype, int N, int H, int W, int cpad, void* stream);

/* ---- data gradients (autograd of the layers above) --------------------------------------
 * conv1d: dx = conv1d(dy; erf_conv1d_pack(mode 1)) * [gate > 0] + residual (gate, residual may be NULL).
 * downsampler: dx = convT(dz[..., :Cout-Cin]) (pool part: erf_maxpool2x2_backward with accumulate).
 * upsampler: dx = conv3x3 s2 (dz) * [gate > 0].
 * output_conv: dlogits NCHW -> patches [N,Hin,Win,ldp] with channel 4*class + (2a+b), ldp =
 *   erf_output_conv_patch_channels(nc); dx = patches x W^T * [gate > 0]. */
int erf_conv1d_dgrad(const void* dy, void* dx, const void* packed_t, const void* gate, const void* residual, int dtype,
                     int N, int H, int W, int C, int axis, int dilation, void* stream);
size_t erf_downsampler_dgrad_packed_bytes(int dtype, int Cin, int Cout);
int erf_downsampler_dgrad_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream);
int erf_downsampler_dgrad(const void* dz, void* dx, const void* packed_t, int dtype, int N, int Hin, int Win, int Cin,
                          int Cout, void* stream);
size_t erf_upsampler_dgrad_packed_bytes(int dtype, int Cin, int Cout);
int erf_upsampler_dgrad_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream);
int erf_upsampler_dgrad(const void* dz, void* dx, const void* packed_t, const void* gate, int dtype, int N, int Hin, int Win,
                        int Cin, int Cout, void* stream);
int erf_output_conv_patch_channels(int nc);
int erf_output_conv_patches(const void* dlogits, int src_dtype, void* patches, int dtype, int N, int Hin, int Win, int nc,
                            void* stream);
size_t erf_output_conv_dgrad_packed_bytes(int dtype, int nc);
int erf_output_conv_dgrad_pack(const float* w, void* packed, int dtype, int nc, void* stream);
int erf_output_conv_dgrad(const void* patches, void* dx, const void* packed_t, const void* gate, int dtype, int N, int Hin,
                          int Win, int nc, void* stream);

/* ---- Encoder.output_conv backward: the encoder-only training pass (train/main.py:203 `model(inputs, only_encode=enc)`,
 * train/erfnet.py:96-97 `if predict: output = self.output_conv(output)`; autograd of that 1x1 convolution).
 * dlogits NCHW [N,nc,H,W] -> g NHWC [N,H,W,ldg], ldg = erf_pointwise_grad_channels(nc), channels >= nc zero;
 * dx [N,H,W,Cin] = g x W (weights packed by erf_pointwise_dgrad_pack from the parameter's [nc,Cin,1,1] layout);
 * dW [nc,Cin] fp32 = sum over pixels of g^T x (workspace: erf_wgrad_workspace_bytes(nc, Cin, 1, N*H*W));
 * the bias gradient is erf_channel_sum(g)[:nc]. */
int erf_pointwise_grad_channels(int nc);
int erf_pointwise_grad_nhwc(const void* dlogits, int src_dtype, void* g, int dtype, int N, int H, int W, int nc, void* stream);
size_t erf_pointwise_dgrad_packed_bytes(int dtype, int Cin, int nc);
int erf_pointwise_dgrad_pack(const float* w, void* packed, int dtype, int Cin, int nc, void* stream);
int erf_pointwise_dgrad(const void* g, void* dx, const void* packed_t, int dtype, int N, int H, int W, int Cin, int nc,
                        void* stream);
int erf_pointwise_wgrad(const void* x, const void* g, float* dw, int dtype, int N, int H, int W, int Cin, int nc,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- weight gradients: dW in the parameter's own torch layout, fp32, overwritten -----------
 * Contraction over pixels on MFMA tiles; per-workgroup partial tiles in `workspace`
 * (erf_wgrad_workspace_bytes(rows = output channels, cin, taps per phase, base pixels)), summed in a
 * fixed order.  x = the layer's forward input, dz = gradient at its (pre-BatchNorm) output.
 * Stem: x_pad from erf_image_to_nhwc_padded (image as NHWC, channels zero-padded to 8 bf16 / 4 fp32). */
size_t erf_wgrad_workspace_bytes(int rows, int cin, int taps_per_phase, long long base_pixels);
int erf_conv1d_wgrad(const void* x, const void* dz, float* dw, int dtype, int N, int H, int W, int C, int axis, int dilation,
                     void* workspace, size_t workspace_bytes, void* stream);
int erf_downsampler_wgrad(const void* x, const void* dz, float* dw, int dtype, int N, int Hin, int Win, int Cin, int Cout,
                          void* workspace, size_t workspace_bytes, void* stream);
/* erf_conv1d_wgrad in two parts, so that all weight-gradient reductions of a backward pass run as one launch:
 * erf_conv1d_wgrad_partial leaves the partial tiles in `workspace` and the pending reduction's arguments in `desc`
 * (host memory, erf_wgrad_reduce_desc_bytes()); the caller gathers the descriptors into one host table and
 * erf_wgrad_reduce_many(table, count) writes every dW (the table travels as kernel arguments: nothing is uploaded, the
 * host memory is free again when the call returns). */
size_t erf_wgrad_reduce_desc_bytes(void);
int erf_conv1d_wgrad_partial(const void* x, const void* dz, float* dw, int dtype, int N, int H, int W, int C, int axis,
                             int dilation, void* workspace, size_t workspace_bytes, void* desc, void* stream);
int erf_wgrad_reduce_many(const void* table, int count, void* stream);
/* The same for up to 4 convolutions on tensors of one shape [N, H, W, C] -- the four factorized convolutions of a
 * non_bottleneck_1d block (train/erfnet.py:44-58) -- as ONE contraction launch: each layer gets a share of the chip and
 * proportionally fewer partial tiles.  Every job brings its own workspace of `workspace_bytes`
 * (erf_wgrad_workspace_bytes(C, C, 3, N*H*W)); descs: count * erf_wgrad_reduce_desc_bytes() of host memory.
 * ERF_EUNSUPPORTED (fp32, other channel counts): nothing was launched, use erf_conv1d_wgrad_partial per layer. */
/* The same deferral for the other weight gradients (down/up-samplers, stem, heads): between
 * erf_wgrad_reduce_record_begin(descs, capacity) and erf_wgrad_reduce_record_end() -- which returns the number of
 * descriptors written -- every erf_*_wgrad call of the calling thread leaves its partial-sum reductions pending in `descs`
 * for a later erf_wgrad_reduce_many; its workspace must stay untouched until then.  A transposed convolution has one
 * reduction per output phase: they are put off only when the workspace holds 4 x erf_wgrad_workspace_bytes(...). */
int erf_wgrad_reduce_record_begin(void* descs, int capacity);
int erf_wgrad_reduce_record_end(void);
typedef struct erf_wgrad_job { const void* x; const void* dz; float* dw; void* workspace; int axis; int dilation; } erf_wgrad_job;
int erf_conv1d_wgrad_partial_many(const erf_wgrad_job* jobs, int count, int dtype, int N, int H, int W, int C,
                                  size_t workspace_bytes, void* descs, void* stream);
int erf_image_to_nhwc_padded(const void* x_nchw, int x_dtype, void* out, int dtype, int N, int H, int W, void* stream);
int erf_stem_wgrad(const void* x_pad, const void* dz, float* dw, int dtype, int N, int Hin, int Win, void* workspace,
                   size_t workspace_bytes, void* stream);
int erf_upsampler_wgrad(const void* x, const void* dz, float* dw, int dtype, int N, int Hin, int Win, int Cin, int Cout,
                        void* workspace, size_t workspace_bytes, void* stream);
int erf_output_conv_wgrad(const void* x, const void* patches, float* dw, int dtype, int N, int Hin, int Win, int nc,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused producers of the training step (no counterpart call in the reference: torch launches the
 * convolution, the BatchNorm statistics and the bias-gradient reduction as separate kernels) -------
 * erf_conv1d_factorized_bn: z = conv(x) + bias (train/erfnet.py:60-61 / :68-69, the 1x3 convolution and the
 *   BatchNorm that follows it) -- the per-channel sum and sum of squares of z are accumulated in the
 *   convolution's epilogue and finalised exactly like erf_bn_stats_finalize (same outputs, same workspace
 *   size erf_bn_workspace_bytes(C)).
 * erf_conv1d_dgrad_bias: erf_conv1d_dgrad plus dbias[c] = sum_p dx[p][c], the bias gradient of the
 *   convolution whose output gradient dx is.
 * erf_conv1d_dgrad_bn_backward: dx = [gate > 0] * (erf_conv1d_dgrad(dy) + residual) followed by erf_bn_backward of
 *   the BatchNorm whose output gradient dx is (autograd of train/erfnet.py:61-62 `bn1 -> relu`, :69-74
 *   `bn2 -> dropout -> relu(output + input)`, :20-21 / :153-154 of the samplers): that layer's two reductions are
 *   taken in the convolution's epilogue, then dz, dgamma, dbeta as erf_bn_backward writes them (gate already
 *   applied, bn_mask = the Dropout2d multipliers [N][C] or NULL; with bn_mask the convolution must run along H).
 *   workspace: erf_bn_workspace_bytes(C).
 * All three return ERF_EUNSUPPORTED without launching when the shape needs the generic kernels: issue the
 * separate calls then. */
int erf_conv1d_factorized_bn(const void* x, void* z, const void* packed, const float* bias, int dtype, int N, int H, int W,
                             int C, int axis, int dilation, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                             float* shift, void* workspace, void* stream);
int erf_conv1d_dgrad_bias(const void* dy, void* dx, const void* packed_t, const void* gate, const void* residual, int dtype,
                          int N, int H, int W, int C, int axis, int dilation, float* dbias, void* workspace, void* stream);
/* ---- Decoder.output_conv fused with the loss (training; train/erfnet.py:171,179 + train/main.py:121,202) ----------
 * erf_output_conv_ce_fwd: out3 = (sum_p w[t_p] * nll_p, sum_p w[t_p], their ratio) of the logits
 *   output_conv(x) + bias against target [N, 2Hin, 2Win] (int64), the logits living in registers only (packed:
 *   erf_output_conv_pack; weight: [nc] or NULL; workspace: erf_output_conv_ce_workspace_bytes).
 * erf_output_conv_ce_bwd: the same logits again, d logits = coef[0] * w[t] * (softmax - onehot) written as the patch
 *   tensor [N, Hin, Win, erf_output_conv_patch_channels(nc)] that erf_output_conv_wgrad / erf_output_conv_dgrad take
 *   (coef: one float in device memory = upstream gradient / global denominator). */
size_t erf_output_conv_ce_workspace_bytes(int N, int Hin, int Win);
int erf_output_conv_ce_fwd(const void* x, const void* packed, const float* bias, const long long* target, const float* weight,
                           float* out3, void* workspace, int dtype, int N, int Hin, int Win, int nc, void* stream);
int erf_output_conv_ce_bwd(const void* x, const void* packed, const float* bias, const long long* target, const float* weight,
                           const float* coef, void* patches, int dtype, int N, int Hin, int Win, int nc, void* stream);

/* erf_conv1d_dgrad_bias in two parts, so that the bias gradients of a whole backward pass are finished by one launch:
 * erf_conv1d_dgrad_bias_partial leaves the partial rows in `workspace` (erf_bn_workspace_bytes(C)) and returns their
 * number (> 0; negative: status); erf_channel_sum_finalize_many(table, count) takes `count` rows of
 * {workspace pointer, partial rows, C, destination pointer} (int64, host memory) and writes every sum. */
int erf_conv1d_dgrad_bias_partial(const void* dy, void* dx, const void* packed_t, const void* gate, const void* residual,
                                  int dtype, int N, int H, int W, int C, int axis, int dilation, void* workspace, void* stream);
int erf_channel_sum_finalize_many(const long long* table, int count, void* stream);
int erf_conv1d_dgrad_bn_backward(const void* dy, void* dx, const void* packed_t, const void* gate, const void* residual,
                                 int dtype, int N, int H, int W, int C, int axis, int dilation, const void* bn_z,
                                 const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_mask,
                                 int batch_stats, void* dz, float* dgamma, float* dbeta, void* workspace, void* stream);
/* erf_conv1d_dgrad_bn_backward with the gate as erf_bn_apply_gates wrote it (one byte per 16-byte chunk of the gating tensor);
 * same results bit for bit. */
int erf_conv1d_dgrad_bn_backward_bits(const void* dy, void* dx, const void* packed_t, const void* gate_bits, const void* residual,
                                      int dtype, int N, int H, int W, int C, int axis, int dilation, const void* bn_z,
                                      const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_mask,
                                      int batch_stats, void* dz, float* dgamma, float* dbeta, void* workspace, void* stream);

/* ---- classifier head of the ImageNet pre-training variant, train/erfnet_imagenet.py:97-120 ----
 * Features.forward after the encoder: MaxPool2d(2, stride 2) -> AvgPool2d(14, 1, 0) on the 28x28 map
 * -> view(N, C); Classifier: Linear(C, K).  x: NHWC [N,H,W,C] (H = W = 28 in the reference; any even
 * H, W averages all pooled positions), feat / dfeat: fp32 [N,C]; dx: gradient at x (every element
 * written; the first maximum of a window takes the gradient, as torch's max-pool backward).
 * Linear: f [N,C], w [K,C], b [K] (may be NULL), y / dy [N,K], all fp32; bwd overwrites df, dw, db. */
int erf_maxpool2_avgpool_fwd(const void* x, int dtype, float* feat, int N, int H, int W, int C, void* stream);
int erf_maxpool2_avgpool_bwd(const void* x, const float* dfeat, void* dx, int dtype, int N, int H, int W, int C, void* stream);
int erf_linear_fwd(const float* f, const float* w, const float* b, float* y, int N, int C, int K, void* stream);
int erf_linear_bwd(const float* f, const float* w, const float* dy, float* df, float* dw, float* db, int N, int C, int K,
                   void* stream);

/* ---- peak probes (no counterpart in the reference, which times whole forward passes only: eval/eval_forwardTime.py:40-59) ----
 * What this part sustains on the two resources the roofline fractions are priced against.  erf_probe_mfma: a
 * matrix-instruction-only loop (dtype 1: v_mfma_f32_16x16x32_bf16 on constant operands, 2: the same on operands that differ
 * in every lane and register, 0: v_mfma_f32_16x16x4_f32; `workgroups` of 8 waves, `iters` trips of 8 independent chains per
 * wave; `sink`: 4 floats -- [1] and [2] receive how far s_memtime and the 100 MHz s_memrealtime moved over workgroup 0's loop);
 * erf_probe_mfma_flop: the flop such a launch executes.  erf_probe_copy: streaming copy of `bytes` (multiple of 16) by `workgroups` x 256 lanes. */
long long erf_probe_mfma_flop(int dtype, int workgroups, int iters);
int erf_probe_mfma(int dtype, int workgroups, int iters, float* sink, void* stream);
int erf_probe_copy(const void* src, void* dst, long long bytes, int workgroups, void* stream);

/* ---- the one entry point that allocates nothing and takes no workspace ------------------------------------------
 * No counterpart in the reference.  It is for callers that cannot allocate device memory: code under HIP-graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.
 * erf_bn_stats_finalize_static = erf_bn_stats_finalize (same kernels, bit-identical outputs) with the per-workgroup
 * partial sums kept in a fixed array inside the library's own device image (about 1 MiB per GPU the library runs on,
 * held until the process ends) instead of a caller workspace.  Limit: C <= erf_static_scratch_max_channels() (128);
 * larger C returns -4 without launching.  All calls on one GPU share that GPU's array, so calls that overlap on one GPU
 * (different streams or threads) race; calls on one stream are ordered and safe. */
int erf_static_scratch_max_channels(void);
int erf_bn_stats_finalize_static(const void* z, int dtype, long long pixels, int C, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                                 float* scale, float* shift, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ERFNET_MI355_H */
