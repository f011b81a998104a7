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
 *   - return value: 0 ok, -1 invalid argument, -2 unsupported shape/dtype, -3 launch failure;
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
int erf_conv1d_factorized(const void* x, void* y, const void* packed, const float* scale, const float* shift,
                          const void* residual, int dtype, int N, int H, int W, int C, int axis, int dilation,
                          int relu, void* stream);

/* ---- DownsamplerBlock, train/erfnet.py:11-22 ------------------------------------------
 * y[..., :Cout-Cin] = conv3x3 s2 p1 (x), y[..., Cout-Cin:] = maxpool2x2 (x), then
 * y = act(scale * y + shift) over all Cout channels (concat happens on write).
 * Cin in {16, 64}: x is NHWC [N,Hin,Win,Cin]; w fp32 [Cout-Cin][Cin][3][3]. */
size_t erf_downsampler_packed_bytes(int dtype, int Cin, int Cout);
int erf_downsampler_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream);
int erf_downsampler(const void* x, void* y, const void* packed, const float* scale, const float* shift, int dtype,
                    int N, int Hin, int Win, int Cin, int Cout, int relu, void* stream);
/* The 3 -> 16 initial_block (train/erfnet.py:80): x is the NCHW network input
 * [N,3,Hin,Win] of type x_dtype; w fp32 [13][3][3][3] unpacked; y NHWC [N,Hin/2,Win/2,16]. */
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

#ifdef __cplusplus
}
#endif
#endif /* ERFNET_MI355_H */
