"""The engine's operators as PyTorch custom ops: namespace `erfnet_mi355`, registered with torch.library.

    torch.ops.erfnet_mi355.<op>(...)

Every op has
  * a device kernel ("CUDA" dispatch key, which is what ROCm devices use): shape checks, output allocation from torch's
    caching allocator, one call through the C ABI (include/erfnet_mi355.h) on the calling thread's current stream;
  * a fake kernel (torch.library.register_fake): output shapes / dtypes / devices only, so FakeTensorMode, meta tensors and
    torch.export's shape propagation see through the ops without touching a GPU.
There is no CPU kernel on purpose: a CPU tensor raises NotImplementedError from the dispatcher.

Activations are NHWC tensors [N,H,W,C] (fp32 or bf16, contiguous); `packed*` arguments are the weight images written by
the erf_*_pack entry points (ops.pack_*).  The ops carry no autograd formulas: training goes through the autograd.Function
classes of autograd.py, whose forward / backward bodies call these ops (and, for the pieces that hand workspaces and deferred
reductions from one launch to the next, the C ABI directly).

The nn.Module forwards (erfnet.py, multi_models.py, erfnet_imagenet.py), loss.py and iouEval.py reach the kernels through
ops.py, whose functional entry points dispatch here; `out=` variants of ops.py write into caller-owned memory and call the
C ABI directly.
"""
from __future__ import annotations

import torch

from ._lib import AXIS_H, AXIS_W, BF16, F32, call, query, size_query

NS = "erfnet_mi355"
LIB = torch.library.Library(NS, "DEF")
OPS = []                      # names, in registration order (tests walk this list)


def _stream() -> int:
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"erfnet_pytorch_amd computes in float32 or bfloat16, got {t.dtype}")


def _p(t):
    return 0 if t is None else t.data_ptr()


def _nhwc(t: torch.Tensor, what: str):
    if t.dim() != 4 or not t.is_contiguous():
        raise ValueError(f"{what}: expected a contiguous NHWC tensor [N,H,W,C], got shape {tuple(t.shape)} strides {t.stride()}")


def _label_dtype(nbytes: int) -> torch.dtype:
    if nbytes == 1:
        return torch.uint8
    if nbytes == 8:
        return torch.int64
    raise ValueError("label maps are uint8 (1 byte) or int64 (8 bytes)")


def _register(schema: str, impl, fake):
    name = schema.split("(", 1)[0]
    LIB.define(schema)
    LIB.impl(name, impl, "CUDA")
    torch.library.register_fake(f"{NS}::{name}", fake, lib=LIB)
    OPS.append(name)


# ------------------------------------------------------------------------------------------------ factorized convolutions
def _conv1d(x, packed, scale, shift, residual, axis, dilation, relu):
    _nhwc(x, "conv1d")
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    call("erf_conv1d_factorized", x.data_ptr(), y.data_ptr(), packed.data_ptr(), _p(scale), _p(shift), _p(residual),
         _dt(x), n, h, w, c, axis, dilation, int(relu), _stream())
    return y


def _like_x(x, *a):
    return torch.empty_like(x)


_register("conv1d(Tensor x, Tensor packed, Tensor? scale, Tensor? shift, Tensor? residual, int axis, int dilation, bool relu) -> Tensor",
          _conv1d, _like_x)


def _conv1d_pair(x, packed_a, shift_a, packed_b, scale_b, shift_b, residual, dilation, relu):
    """conv3x1 -> +shift_a -> ReLU -> conv1x3 (dilated) -> scale_b / shift_b (+ residual) -> ReLU.  One launch where the
    library takes the shape (bf16, the tile utilisation rule of erf_conv1d_pair), else the two single-convolution launches."""
    _nhwc(x, "conv1d_pair")
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    rc = query("erf_conv1d_pair", x.data_ptr(), y.data_ptr(), packed_a.data_ptr(), packed_b.data_ptr(), 0, _p(shift_a),
               _p(scale_b), _p(shift_b), _p(residual), _dt(x), n, h, w, c, dilation, int(relu), _stream())
    if rc == 0:
        return y
    if rc != -2:
        raise RuntimeError(f"erf_conv1d_pair failed with status {rc}")
    mid = _conv1d(x, packed_a, None, shift_a, None, AXIS_H, dilation, True)
    return _conv1d(mid, packed_b, scale_b, shift_b, residual, AXIS_W, dilation, relu)


_register("conv1d_pair(Tensor x, Tensor packed_a, Tensor shift_a, Tensor packed_b, Tensor? scale_b, Tensor shift_b, Tensor? residual, "
          "int dilation, bool relu) -> Tensor", _conv1d_pair, _like_x)


def _nb1d_block(x, w1, w2, w3, w4, table, dilation):
    _nhwc(x, "nb1d_block")
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    rc = query("erf_nb1d_block_infer", x.data_ptr(), y.data_ptr(), w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), w4.data_ptr(),
               table.data_ptr(), _dt(x), n, h, w, c, dilation, _stream())
    if rc == -2:
        # a shape the one-launch kernel declines (other channel counts / dtypes, very large maps): the two pair launches, which
        # decline to single launches themselves; table rows = [1, b1, s2, t2, 1, b3, s4, t4]
        mid = _conv1d_pair(x, w1, table[1], w2, table[2], table[3], None, 1, True)
        return _conv1d_pair(mid, w3, table[5], w4, table[6], table[7], x, dilation, True)
    if rc != 0:
        raise RuntimeError(f"erf_nb1d_block_infer failed with status {rc}")
    return y


_register("nb1d_block(Tensor x, Tensor w1, Tensor w2, Tensor w3, Tensor w4, Tensor table, int dilation) -> Tensor",
          _nb1d_block, _like_x)


def _conv1d_dgrad(dy, packed_t, gate, residual, axis, dilation):
    _nhwc(dy, "conv1d_dgrad")
    n, h, w, c = dy.shape
    dx = torch.empty_like(dy)
    call("erf_conv1d_dgrad", dy.data_ptr(), dx.data_ptr(), packed_t.data_ptr(), _p(gate), _p(residual), _dt(dy), n, h, w,
         c, axis, dilation, _stream())
    return dx


_register("conv1d_dgrad(Tensor dy, Tensor packed_t, Tensor? gate, Tensor? residual, int axis, int dilation) -> Tensor",
          _conv1d_dgrad, _like_x)


def _conv1d_wgrad(x, dz, axis, dilation):
    _nhwc(x, "conv1d_wgrad")
    n, h, w, c = x.shape
    dw = torch.empty((c, c, 3, 1) if axis == 0 else (c, c, 1, 3), dtype=torch.float32, device=x.device)
    nb = size_query("erf_wgrad_workspace_bytes", c, c, 3, n * h * w)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    call("erf_conv1d_wgrad", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), _dt(x), n, h, w, c, axis, dilation, ws.data_ptr(), nb, _stream())
    return dw


def _conv1d_wgrad_fake(x, dz, axis, dilation):
    c = x.shape[3]
    return x.new_empty((c, c, 3, 1) if axis == 0 else (c, c, 1, 3), dtype=torch.float32)


_register("conv1d_wgrad(Tensor x, Tensor dz, int axis, int dilation) -> Tensor", _conv1d_wgrad, _conv1d_wgrad_fake)


# ------------------------------------------------------------------------------------------------ samplers and heads
def _downsampler(x, packed, scale, shift, cout, relu):
    _nhwc(x, "downsampler")
    n, h, w, cin = x.shape
    y = torch.empty((n, h // 2, w // 2, cout), dtype=x.dtype, device=x.device)
    call("erf_downsampler", x.data_ptr(), y.data_ptr(), packed.data_ptr(), _p(scale), _p(shift), _dt(x), n, h, w, cin, cout,
         int(relu), _stream())
    return y


_register("downsampler(Tensor x, Tensor packed, Tensor? scale, Tensor? shift, int cout, bool relu) -> Tensor", _downsampler,
          lambda x, packed, scale, shift, cout, relu: x.new_empty((x.shape[0], x.shape[1] // 2, x.shape[2] // 2, cout)))


def _stem_geometry(x):
    """(source code, N, H, W): fp32 / bf16 / uint8 [N,3,H,W] or uint8 [N,H,W,3] (interleaved, as image decoders deliver)."""
    if x.dim() != 4:
        raise ValueError("the initial block takes a batch of 3-channel images")
    if x.dtype == torch.uint8 and x.shape[1] != 3 and x.shape[3] == 3:
        return 3, x.shape[0], x.shape[1], x.shape[2]
    if x.shape[1] != 3:
        raise ValueError("the initial block takes 3-channel images: [N,3,H,W], or uint8 [N,H,W,3]")
    return (2 if x.dtype == torch.uint8 else _dt(x)), x.shape[0], x.shape[2], x.shape[3]


def _stem(x_nchw, w, scale, shift, code, relu):
    src, n, h, w_ = _stem_geometry(x_nchw)
    if not x_nchw.is_contiguous():
        raise ValueError("the initial block takes contiguous images")
    y = torch.empty((n, h // 2, w_ // 2, 16), dtype=torch.bfloat16 if code == BF16 else torch.float32, device=x_nchw.device)
    call("erf_stem_downsampler", x_nchw.data_ptr(), src, y.data_ptr(), w.data_ptr(), _p(scale), _p(shift), code,
         n, h, w_, int(relu), _stream())
    return y


def _stem_fake(x, w, scale, shift, code, relu):
    _, n, h, w_ = _stem_geometry(x)
    return x.new_empty((n, h // 2, w_ // 2, 16), dtype=torch.bfloat16 if code == BF16 else torch.float32)


_register("stem(Tensor x_nchw, Tensor w, Tensor? scale, Tensor? shift, int code, bool relu) -> Tensor", _stem, _stem_fake)


def _upsampler(x, packed, scale, shift, cout, relu):
    _nhwc(x, "upsampler")
    n, h, w, cin = x.shape
    y = torch.empty((n, 2 * h, 2 * w, cout), dtype=x.dtype, device=x.device)
    call("erf_upsampler_convT3x3s2", x.data_ptr(), y.data_ptr(), packed.data_ptr(), _p(scale), _p(shift), _dt(x), n, h, w,
         cin, cout, int(relu), _stream())
    return y


_register("upsampler(Tensor x, Tensor packed, Tensor? scale, Tensor? shift, int cout, bool relu) -> Tensor", _upsampler,
          lambda x, packed, scale, shift, cout, relu: x.new_empty((x.shape[0], 2 * x.shape[1], 2 * x.shape[2], cout)))


def _output_conv(x, packed, bias, nc, want_logits, logits_f32, label_bytes):
    """Decoder.output_conv (ConvTranspose2d 2x2 stride 2): NCHW logits and / or the arg-max label map [N,2H,2W]
    (label_bytes 0: none, 1: uint8, 8: int64).  An output that was not asked for comes back as an empty tensor."""
    _nhwc(x, "output_conv")
    n, h, w, cin = x.shape
    logits = torch.empty((n, nc, 2 * h, 2 * w) if want_logits else (0,), dtype=torch.float32 if logits_f32 else x.dtype, device=x.device)
    labels = torch.empty((n, 2 * h, 2 * w) if label_bytes else (0,), dtype=_label_dtype(label_bytes or 8), device=x.device)
    call("erf_output_convT2x2s2", x.data_ptr(), packed.data_ptr(), bias.data_ptr(), logits.data_ptr() if want_logits else 0,
         int(logits_f32), labels.data_ptr() if label_bytes else 0, label_bytes, _dt(x), n, h, w, nc, _stream())
    return logits, labels


def _output_conv_fake(x, packed, bias, nc, want_logits, logits_f32, label_bytes):
    n, h, w, _ = x.shape
    return (x.new_empty((n, nc, 2 * h, 2 * w) if want_logits else (0,), dtype=torch.float32 if logits_f32 else x.dtype),
            x.new_empty((n, 2 * h, 2 * w) if label_bytes else (0,), dtype=_label_dtype(label_bytes or 8)))


_register("output_conv(Tensor x, Tensor packed, Tensor bias, int nc, bool want_logits, bool logits_f32, int label_bytes) -> (Tensor, Tensor)",
          _output_conv, _output_conv_fake)


def _pointwise(x, packed, bias, nc, out_f32):
    _nhwc(x, "pointwise")
    n, h, w, cin = x.shape
    y = torch.empty((n, nc, h, w), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    call("erf_pointwise_conv", x.data_ptr(), y.data_ptr(), packed.data_ptr(), _p(bias), int(out_f32), _dt(x), n, h, w, cin, nc, _stream())
    return y


_register("pointwise(Tensor x, Tensor packed, Tensor? bias, int nc, bool out_f32) -> Tensor", _pointwise,
          lambda x, packed, bias, nc, out_f32: x.new_empty((x.shape[0], nc, x.shape[1], x.shape[2]),
                                                           dtype=torch.float32 if out_f32 else x.dtype))


# ------------------------------------------------------------------------------------------------ multi-task variant (train/multi_models.py)
def _conv3x3(x, packed, scale, shift, cout, stride, relu):
    """3x3 convolution, padding 1, stride 1 or 2, + per-channel scale / shift (+ ReLU): DownsamplerBlock_simplified's and the
    classifier's convolution (train/multi_models.py:23-36, :181-189)."""
    _nhwc(x, "conv3x3")
    n, h, w, cin = x.shape
    y = torch.empty((n, (h + stride - 1) // stride, (w + stride - 1) // stride, cout), dtype=x.dtype, device=x.device)
    call("erf_conv3x3", x.data_ptr(), y.data_ptr(), packed.data_ptr(), _p(scale), _p(shift), _dt(x), n, h, w, cin, cout, stride,
         int(relu), _stream())
    return y


_register("conv3x3(Tensor x, Tensor packed, Tensor? scale, Tensor? shift, int cout, int stride, bool relu) -> Tensor", _conv3x3,
          lambda x, packed, scale, shift, cout, stride, relu: x.new_empty(
              (x.shape[0], (x.shape[1] + stride - 1) // stride, (x.shape[2] + stride - 1) // stride, cout)))


def _maxpool2x2_affine(z, scale, shift, relu):
    """maxpool 2x2 -> BatchNorm (folded to scale / shift) -> ReLU (train/multi_models.py:32-35)."""
    _nhwc(z, "maxpool2x2_affine")
    n, h, w, c = z.shape
    if (h | w) & 1:
        raise ValueError("maxpool2x2_affine needs even H and W")
    y = torch.empty((n, h // 2, w // 2, c), dtype=z.dtype, device=z.device)
    call("erf_maxpool2x2_affine", z.data_ptr(), y.data_ptr(), scale.data_ptr(), shift.data_ptr(), _dt(z), n, h, w, c, int(relu), _stream())
    return y


_register("maxpool2x2_affine(Tensor z, Tensor scale, Tensor shift, bool relu) -> Tensor", _maxpool2x2_affine,
          lambda z, scale, shift, relu: z.new_empty((z.shape[0], z.shape[1] // 2, z.shape[2] // 2, z.shape[3])))


def _concat_channels(a, b):
    """torch.cat([a, b], 1) of the reference (train/multi_models.py:141,145) on NHWC maps of one dtype."""
    _nhwc(a, "concat_channels")
    _nhwc(b, "concat_channels")
    if a.shape[:3] != b.shape[:3] or a.dtype != b.dtype:
        raise ValueError("concat_channels: maps of one size and dtype")
    n, h, w, ca = a.shape
    cb = b.shape[3]
    out = torch.empty((n, h, w, ca + cb), dtype=a.dtype, device=a.device)
    call("erf_concat_channels", a.data_ptr(), ca, b.data_ptr(), cb, out.data_ptr(), _dt(a), n, h, w, _stream())
    return out


_register("concat_channels(Tensor a, Tensor b) -> Tensor", _concat_channels,
          lambda a, b: a.new_empty((a.shape[0], a.shape[1], a.shape[2], a.shape[3] + b.shape[3])))


def _convT2x2s2(x, packed, bias, nc, out_f32):
    """ConvTranspose2d(Cin, nc, 2, stride 2) + bias -> NCHW maps (the regression decoders' head, train/multi_models.py:131,147)."""
    _nhwc(x, "convT2x2s2")
    n, h, w, cin = x.shape
    y = torch.empty((n, nc, 2 * h, 2 * w), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    call("erf_convT2x2s2", x.data_ptr(), y.data_ptr(), packed.data_ptr(), bias.data_ptr(), int(out_f32), _dt(x), n, h, w, cin, nc, _stream())
    return y


_register("convT2x2s2(Tensor x, Tensor packed, Tensor bias, int nc, bool out_f32) -> Tensor", _convT2x2s2,
          lambda x, packed, bias, nc, out_f32: x.new_empty((x.shape[0], nc, 2 * x.shape[1], 2 * x.shape[2]),
                                                           dtype=torch.float32 if out_f32 else x.dtype))


def _spatial_mean(y):
    """Mean over H and W per image and channel -> fp32 [N, C] (train/multi_models.py:198-199)."""
    _nhwc(y, "spatial_mean")
    n, h, w, c = y.shape
    feat = torch.empty((n, c), dtype=torch.float32, device=y.device)
    call("erf_spatial_mean", y.data_ptr(), _dt(y), feat.data_ptr(), n, h, w, c, _stream())
    return feat


_register("spatial_mean(Tensor y) -> Tensor", _spatial_mean, lambda y: y.new_empty((y.shape[0], y.shape[3]), dtype=torch.float32))


def _linear(feat, weight, bias):
    """nn.Linear on fp32 features [N, C] -> [N, K] (train/multi_models.py:200, train/erfnet_imagenet.py:116)."""
    if feat.dtype != torch.float32 or weight.dtype != torch.float32 or feat.dim() != 2 or weight.shape[1] != feat.shape[1]:
        raise ValueError("linear: fp32 features [N, C] and weight [K, C]")
    n, c = feat.shape
    out = torch.empty((n, weight.shape[0]), dtype=torch.float32, device=feat.device)
    call("erf_linear_fwd", feat.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), n, c, weight.shape[0], _stream())
    return out


_register("linear(Tensor feat, Tensor weight, Tensor bias) -> Tensor", _linear,
          lambda feat, weight, bias: feat.new_empty((feat.shape[0], weight.shape[0])))


def _image_to_nhwc_channels(image, code, channels):
    """NCHW fp32 / bf16 image -> NHWC with `channels` zero-padded channels in the compute dtype."""
    if image.dim() != 4 or not image.is_contiguous():
        raise ValueError("image_to_nhwc_channels: contiguous NCHW image")
    n, c, h, w = image.shape
    x = torch.empty((n, h, w, channels), dtype=torch.bfloat16 if code == BF16 else torch.float32, device=image.device)
    call("erf_image_to_nhwc_channels", image.data_ptr(), _dt(image), x.data_ptr(), code, n, h, w, channels, _stream())
    return x


_register("image_to_nhwc_channels(Tensor image, int code, int channels) -> Tensor", _image_to_nhwc_channels,
          lambda image, code, channels: image.new_empty((image.shape[0], image.shape[2], image.shape[3], channels),
                                                        dtype=torch.bfloat16 if code == BF16 else torch.float32))


# ------------------------------------------------------------------------------------------------ BatchNorm / dropout pieces
def _bn_apply(z, center, scale, shift, residual, mask, relu):
    """y = act(((z - center) * scale + shift) * mask[n][c] + residual)"""
    _nhwc(z, "bn_apply")
    n, h, w, c = z.shape
    y = torch.empty_like(z)
    call("erf_bn_apply", z.data_ptr(), y.data_ptr(), _p(center), scale.data_ptr(), shift.data_ptr(), _p(residual), _p(mask),
         _dt(z), n, h * w, c, int(relu), _stream())
    return y


_register("bn_apply(Tensor z, Tensor? center, Tensor scale, Tensor shift, Tensor? residual, Tensor? mask, bool relu) -> Tensor",
          _bn_apply, _like_x)


def _channel_sum(z):
    c = z.shape[-1]
    out = torch.empty(c, dtype=torch.float32, device=z.device)
    ws = torch.empty(size_query("erf_bn_workspace_bytes", c), dtype=torch.uint8, device=z.device)
    call("erf_channel_sum", z.data_ptr(), _dt(z), z.numel() // c, c, out.data_ptr(), ws.data_ptr(), _stream())
    return out


_register("channel_sum(Tensor z) -> Tensor", _channel_sum, lambda z: z.new_empty((z.shape[-1],), dtype=torch.float32))


def _dropout2d_mask(like, n, c, p, seed):
    mask = torch.empty((n, c), dtype=torch.float32, device=like.device)
    call("erf_dropout2d_mask", mask.data_ptr(), n * c, float(p), seed & 0xFFFFFFFFFFFFFFFF, _stream())
    return mask


_register("dropout2d_mask(Tensor like, int n, int c, float p, int seed) -> Tensor", _dropout2d_mask,
          lambda like, n, c, p, seed: like.new_empty((n, c), dtype=torch.float32))


# ------------------------------------------------------------------------------------------------ loss and metrics
def _weighted_ce_fwd(logits, target, weight):
    """[sum w*nll, sum w, their ratio] of NCHW logits against int64 targets [N,H,W] (CrossEntropyLoss2d, train/main.py:58-66)."""
    n, c, h, w = logits.shape
    if not logits.is_contiguous() or not target.is_contiguous() or target.dtype != torch.int64:
        raise ValueError("weighted_ce_fwd: contiguous NCHW logits and contiguous int64 targets")
    out = torch.empty(3, dtype=torch.float32, device=logits.device)
    ws = torch.empty(query("erf_ce_workspace_bytes"), dtype=torch.uint8, device=logits.device)
    call("erf_weighted_ce_fwd", logits.data_ptr(), _dt(logits), target.data_ptr(), _p(weight), n, c, h * w, out.data_ptr(),
         ws.data_ptr(), _stream())
    return out


_register("weighted_ce_fwd(Tensor logits, Tensor target, Tensor? weight) -> Tensor", _weighted_ce_fwd,
          lambda logits, target, weight: logits.new_empty((3,), dtype=torch.float32))


def _weighted_ce_bwd(logits, target, weight, coef):
    n, c, h, w = logits.shape
    dz = torch.empty_like(logits)
    call("erf_weighted_ce_bwd", logits.data_ptr(), _dt(logits), target.data_ptr(), _p(weight), n, c, h * w, coef.data_ptr(),
         dz.data_ptr(), _stream())
    return dz


_register("weighted_ce_bwd(Tensor logits, Tensor target, Tensor? weight, Tensor coef) -> Tensor", _weighted_ce_bwd,
          lambda logits, target, weight, coef: torch.empty_like(logits))


def _argmax_c(logits, label_bytes):
    """outputs.max(1)[1] on NCHW-contiguous logits -> [N,H,W] (first maximum wins, as torch's)."""
    if not logits.is_contiguous():
        raise ValueError("argmax_c: contiguous NCHW logits")
    n, c, h, w = logits.shape
    lab = torch.empty((n, h, w), dtype=_label_dtype(label_bytes), device=logits.device)
    call("erf_argmax_c", logits.data_ptr(), _dt(logits), lab.data_ptr(), label_bytes, n, c, h * w, _stream())
    return lab


_register("argmax_c(Tensor logits, int label_bytes) -> Tensor", _argmax_c,
          lambda logits, label_bytes: logits.new_empty((logits.shape[0], logits.shape[2], logits.shape[3]), dtype=_label_dtype(label_bytes)))


def _iou_hist_accumulate(pred, label, nc, hist, olabel=-1, nlabel=0):
    """hist (int64 [nc*nc+1]) += confusion counts of (label, pred); eval/iouEval.py:18-60 as a histogram.
    olabel >= 0: counted as after Relabel(olabel, nlabel) on the label map (eval/eval_iou.py:37)."""
    if pred.numel() != label.numel():
        raise ValueError("prediction and label maps differ in size")
    for t in (pred, label):
        if t.dtype not in (torch.uint8, torch.int64) or not t.is_contiguous():
            raise TypeError("label maps must be contiguous uint8 or int64 tensors")
    if hist.dtype != torch.int64 or hist.numel() < nc * nc + 1:
        raise ValueError("hist: int64 [nc*nc+1]")
    if olabel >= 0:
        call("erf_iou_hist_accumulate_relabel", pred.data_ptr(), pred.element_size(), label.data_ptr(), label.element_size(),
             pred.numel(), nc, int(olabel), int(nlabel), hist.data_ptr(), _stream())
    else:
        call("erf_iou_hist_accumulate", pred.data_ptr(), pred.element_size(), label.data_ptr(), label.element_size(),
             pred.numel(), nc, hist.data_ptr(), _stream())


_register("iou_hist_accumulate(Tensor pred, Tensor label, int nc, Tensor(a!) hist, int olabel=-1, int nlabel=0) -> ()",
          _iou_hist_accumulate, lambda pred, label, nc, hist, olabel=-1, nlabel=0: None)


def _labels_from_u8(src, olabel, nlabel):
    """ToLabel() + Relabel(olabel, nlabel) (train/transform.py:53-68): uint8 label maps -> int64, one value mapped to another
    (olabel < 0: conversion only)."""
    if src.dtype != torch.uint8 or not src.is_contiguous():
        raise TypeError("labels_from_u8: a contiguous uint8 tensor")
    if src.data_ptr() % 16:                       # a slice that starts inside a 16-byte group: the kernel reads whole groups
        src = src.clone()
    dst = torch.empty(src.shape, dtype=torch.int64, device=src.device)
    call("erf_labels_from_u8", src.data_ptr(), dst.data_ptr(), src.numel(), int(olabel), int(nlabel), _stream())
    return dst


_register("labels_from_u8(Tensor src, int olabel, int nlabel) -> Tensor", _labels_from_u8,
          lambda src, olabel, nlabel: src.new_empty(src.shape, dtype=torch.int64))


# ------------------------------------------------------------------------------------------------ optimiser
def _adam_fused_arena(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale):
    for t in (params, grads, exp_avg, exp_avg_sq):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != params.numel():
            raise ValueError("adam_fused_arena_: four contiguous fp32 tensors of one size")
    call("erf_adam_fused_arena", params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), params.numel(),
         float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), _stream())


_register("adam_fused_arena_(Tensor(a!) params, Tensor grads, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, float lr, float beta1, "
          "float beta2, float eps, float weight_decay, int step, float grad_scale) -> ()", _adam_fused_arena,
          lambda *a: None)
