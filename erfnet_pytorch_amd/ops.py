"""Tensor-level wrappers over the C ABI (libermi355.so).

Inside the package activations are NHWC tensors of shape [N,H,W,C] (contiguous, fp32 or
bf16).  At the nn.Module boundary they appear as the logical [N,C,H,W] tensors the reference
produces, backed by channels_last memory (a zero-copy permute of the NHWC tensor).
All launches go to torch's current HIP stream, so they compose with torch ops, events and
HIP-graph capture (torch.cuda.graph).
"""
from __future__ import annotations

import torch

from . import _lib
from . import torch_ops  # noqa: F401  (registers torch.ops.erfnet_mi355.*)
from ._lib import AXIS_H, AXIS_W, BF16, F32, call, query

T = torch.ops.erfnet_mi355


def _stream() -> int:
    """Raw handle of torch's current stream on the current device (the Stream-object route costs several microseconds per
    call, which adds up over the few hundred launches of a training step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"erfnet_pytorch_amd computes in float32 or bfloat16, got {t.dtype}")


# 8-bit images as decoders deliver them (the reference's loaders hand PIL images to ToTensor(), train/main.py:47,66): the
# initial block converts them on load to value / 255 in fp32, so a uint8 batch gives the bits of `images.float() / 255`.
U8_NCHW, U8_NHWC = 2, 3


def image_source(t: torch.Tensor):
    """(source code for the initial block, N, H, W) of a network input: fp32 / bf16 [N,3,H,W], uint8 [N,3,H,W] or uint8 [N,H,W,3]."""
    if t.dim() != 4:
        raise ValueError(f"expected a batch of images [N,3,H,W] (or uint8 [N,H,W,3]), got shape {tuple(t.shape)}")
    if t.dtype == torch.uint8:
        if t.shape[1] == 3:
            return U8_NCHW, t.shape[0], t.shape[2], t.shape[3]
        if t.shape[3] == 3:
            return U8_NHWC, t.shape[0], t.shape[1], t.shape[2]
        raise ValueError(f"uint8 images are [N,3,H,W] or [N,H,W,3], got shape {tuple(t.shape)}")
    if t.shape[1] != 3:
        raise ValueError("the initial block takes 3-channel images")
    return dt(t), t.shape[0], t.shape[2], t.shape[3]


def image_code(t: torch.Tensor) -> int:
    """Compute dtype code an image batch implies when the model sets none: uint8 images are fp32 after ToTensor()."""
    return F32 if t.dtype == torch.uint8 else dt(t)


def image_is_f32(t: torch.Tensor) -> bool:
    return t.dtype in (torch.float32, torch.uint8)


def tdtype(code: int) -> torch.dtype:
    return torch.bfloat16 if code == BF16 else torch.float32


def _p(t):
    return 0 if t is None else t.data_ptr()


def require_gpu(t: torch.Tensor, who: str):
    if not t.is_cuda:
        raise RuntimeError(f"{who}: erfnet_pytorch_amd runs on MI355X (gfx950) device tensors only; "
                           "got a CPU tensor and there is no CPU path in this package")


# --------------------------------------------------------------------------- layout glue
def to_nhwc(x: torch.Tensor, dtype: torch.dtype | None = None) -> torch.Tensor:
    """Logical [N,C,H,W] -> NHWC tensor [N,H,W,C]; zero-copy when x is channels_last already."""
    dtype = dtype or x.dtype
    n, c, h, w = x.shape
    if x.dtype == dtype and x.permute(0, 2, 3, 1).is_contiguous():
        return x.permute(0, 2, 3, 1)
    x = x.contiguous()
    out = torch.empty((n, h, w, c), dtype=dtype, device=x.device)
    call("erf_nchw_to_nhwc", x.data_ptr(), dt(x), out.data_ptr(), dt(out), n, c, h, w, _stream())
    return out


def to_nchw_view(y: torch.Tensor) -> torch.Tensor:
    """NHWC tensor [N,H,W,C] -> logical [N,C,H,W] (channels_last strides), zero-copy."""
    return y.permute(0, 3, 1, 2)


def nhwc_to_nchw_contiguous(y: torch.Tensor, dtype: torch.dtype | None = None) -> torch.Tensor:
    n, h, w, c = y.shape
    out = torch.empty((n, c, h, w), dtype=dtype or y.dtype, device=y.device)
    call("erf_nhwc_to_nchw", y.data_ptr(), dt(y), out.data_ptr(), dt(out), n, c, h, w, _stream())
    return out


# --------------------------------------------------------------------------- weight packing
# Optimisers that update parameters through raw pointers (the fused arena kernels) do not bump torch's
# per-tensor version counters; they bump this epoch instead, and every weight-derived cache keys on it.
PARAM_EPOCH = [0]


def bump_param_epoch():
    PARAM_EPOCH[0] += 1


class PackedBank:
    """Persistent packed buffers (forward and data-gradient form) of all factorized convolutions of a model,
    refreshed by ONE launch (erf_conv1d_pack_many).  `refresh()` must run after every parameter update and
    before the next forward; blocks look their weights up through `lookup` and fall back to packing on the
    fly when the bank is stale or was built for another dtype."""

    def __init__(self, model, code: int):
        from . import erfnet as E
        self.code, self.epoch = code, -1
        self.max_c = 16
        self._entries = []                                   # (weight parameter, packed buffer, C, mode)
        dev = next(model.parameters()).device
        for mod in model.modules():
            if not isinstance(mod, E.non_bottleneck_1d):
                continue
            bank = {}
            for name in ("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2"):
                w = getattr(mod, name).weight
                c = w.shape[0]
                self.max_c = max(self.max_c, c)
                for mode in (0, 1):
                    buf = _packed(query("erf_conv1d_packed_bytes", code, c), dev)
                    bank[(name, mode)] = buf
                    self._entries.append((w, buf, c, mode))
            mod.__dict__["_erf_packed_bank"] = (self, bank)
        self.count = len(self._entries)
        self._sig, self.table = None, None
        # the other layers (samplers, the two heads): (weight, buffer, pack entry point, its shape arguments);
        # their re-packs are recorded once and replayed by one launch per refresh (erf_pack_replay)
        self._generic = []
        for mod in model.modules():
            kinds = None
            if isinstance(mod, E.DownsamplerBlock) and mod.ninput != 3:
                kinds = {"fwd": ("erf_downsampler", (mod.ninput, mod.noutput)), "dgrad": ("erf_downsampler_dgrad", (mod.ninput, mod.noutput))}
                w = mod.conv.weight
            elif isinstance(mod, E.UpsamplerBlock):
                kinds = {"fwd": ("erf_upsampler", (mod.ninput, mod.noutput)), "dgrad": ("erf_upsampler_dgrad", (mod.ninput, mod.noutput))}
                w = mod.conv.weight
            elif isinstance(mod, E.Decoder):
                oc = mod.output_conv
                kinds = {"fwd": ("erf_output_conv", (oc.out_channels,)), "dgrad": ("erf_output_conv_dgrad", (oc.out_channels,))}
                w, mod = oc.weight, oc
            elif isinstance(mod, E.Encoder):
                oc = mod.output_conv
                kinds = {"fwd": ("erf_pointwise", (oc.in_channels, oc.out_channels)), "dgrad": ("erf_pointwise_dgrad", (oc.in_channels, oc.out_channels))}
                w, mod = oc.weight, oc
            if kinds is None:
                continue
            bufs = {}
            for kind, (fn, shape) in kinds.items():
                nbytes = query(fn + "_packed_bytes", code, *shape)
                if nbytes <= 0:
                    continue
                buf = _packed(nbytes, dev)
                bufs[kind] = buf
                self._generic.append((w, buf, fn + "_pack", shape))
            mod.__dict__["_erf_packed_generic"] = (self, bufs, w)
        self._gsig, self._gtable, self._gcount, self._gmax = None, None, 0, 0

    def _record_generic(self):
        import ctypes
        desc = query("erf_pack_desc_bytes")
        cap = 8 * len(self._generic)                         # an entry point may issue several re-packs (one per phase)
        host = torch.zeros(cap * desc, dtype=torch.uint8)
        out = torch.zeros(2, dtype=torch.int64)
        dt_out = torch.zeros(1, dtype=torch.int32)
        call("erf_pack_record_begin", host.data_ptr(), cap)
        try:
            for (w, buf, fn, shape) in self._generic:
                call(fn, w.data_ptr(), buf.data_ptr(), self.code, *shape, _stream())
        finally:
            n = query("erf_pack_record_end", out.data_ptr(), dt_out.data_ptr())
        if n < 0:
            raise _lib.ErfError(f"recording the weight re-packs failed with status {n}")
        self._gcount, self._gmax = int(n), int(out[0])
        self._gtable = host[: n * desc].to(self._generic[0][1].device) if n else None

    def refresh(self):
        sig = tuple(w.data_ptr() for (w, _b, _c, _m) in self._entries)
        if sig != self._sig:                                 # parameters moved (first use, .to(), arena re-binding)
            rows = [[w.data_ptr(), b.data_ptr(), c, m] for (w, b, c, m) in self._entries]
            self.table = torch.tensor(rows, dtype=torch.int64).to(self._entries[0][1].device) if rows else None
            self._sig = sig
        if self.count:
            call("erf_conv1d_pack_many", self.table.data_ptr(), self.count, self.max_c, self.code, _stream())
        gsig = tuple(w.data_ptr() for (w, _b, _f, _s) in self._generic)
        if gsig != self._gsig and self._generic:
            self._record_generic()
            self._gsig = gsig
        if self._gcount:
            call("erf_pack_replay", self._gtable.data_ptr(), self._gcount, self._gmax, self.code, _stream())
        self.epoch = PARAM_EPOCH[0]


def packed_generic(mod, kind: str, w: torch.Tensor, code: int, pack_now):
    """Packed weights of a sampler / head layer: from the model's PackedBank when it is current, else `pack_now()`."""
    ent = mod.__dict__.get("_erf_packed_generic") if mod is not None else None
    if ent is not None:
        bank, bufs, wp = ent
        if bank.code == code and bank.epoch == PARAM_EPOCH[0] and kind in bufs and wp.data_ptr() == w.data_ptr() \
                and w.dtype == torch.float32:
            return bufs[kind]
    return pack_now()


def packed_conv1d(mod, name: str, w: torch.Tensor, code: int, mode: int = 0) -> torch.Tensor:
    """Packed weights of `mod.<name>`: from the model's PackedBank when it is current, else packed now."""
    ent = mod.__dict__.get("_erf_packed_bank")
    if ent is not None:
        bank, bufs = ent
        if bank.code == code and bank.epoch == PARAM_EPOCH[0] and getattr(mod, name).weight.data_ptr() == w.data_ptr():
            return bufs[(name, mode)]
    return pack_conv1d(w, code, mode)


def _packed(nbytes: int, device) -> torch.Tensor:
    if nbytes <= 0:
        raise _lib.ErfError("unsupported layer shape for packing")
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def pack_conv1d(w: torch.Tensor, code: int, mode: int = 0) -> torch.Tensor:
    """w: Conv2d weight (C,C,3,1) or (C,C,1,3), fp32."""
    c = w.shape[0]
    w = w.detach().float().contiguous()
    buf = _packed(query("erf_conv1d_packed_bytes", code, c), w.device)
    call("erf_conv1d_pack", w.data_ptr(), buf.data_ptr(), code, c, mode, _stream())
    return buf


def pack_downsampler(w: torch.Tensor, code: int) -> torch.Tensor:
    cc, cin = w.shape[0], w.shape[1]
    w = w.detach().float().contiguous()
    buf = _packed(query("erf_downsampler_packed_bytes", code, cin, cc + cin), w.device)
    call("erf_downsampler_pack", w.data_ptr(), buf.data_ptr(), code, cin, cc + cin, _stream())
    return buf


def pack_upsampler(w: torch.Tensor, code: int) -> torch.Tensor:
    cin, cout = w.shape[0], w.shape[1]
    w = w.detach().float().contiguous()
    buf = _packed(query("erf_upsampler_packed_bytes", code, cin, cout), w.device)
    call("erf_upsampler_pack", w.data_ptr(), buf.data_ptr(), code, cin, cout, _stream())
    return buf


def pack_output_conv(w: torch.Tensor, code: int) -> torch.Tensor:
    nc = w.shape[1]
    w = w.detach().float().contiguous()
    buf = _packed(query("erf_output_conv_packed_bytes", code, nc), w.device)
    call("erf_output_conv_pack", w.data_ptr(), buf.data_ptr(), code, nc, _stream())
    return buf


def pack_pointwise(w: torch.Tensor, code: int) -> torch.Tensor:
    nc, cin = w.shape[0], w.shape[1]
    w = w.detach().float().contiguous()
    buf = _packed(query("erf_pointwise_packed_bytes", code, cin, nc), w.device)
    call("erf_pointwise_pack", w.data_ptr(), buf.data_ptr(), code, cin, nc, _stream())
    return buf


# --------------------------------------------------------------------------- forward ops (NHWC in, NHWC out)
def conv1d(x, packed, scale, shift, residual, axis: int, dilation: int, relu: bool, out=None):
    if out is None:
        return T.conv1d(x, packed, scale, shift, residual, axis, dilation, relu)
    n, h, w, c = x.shape
    y = out
    call("erf_conv1d_factorized", x.data_ptr(), y.data_ptr(), packed.data_ptr(), _p(scale), _p(shift), _p(residual),
         dt(x), n, h, w, c, axis, dilation, int(relu), _stream())
    return y


def conv1d_pair(x, packed_a, shift_a, packed_b, scale_b, shift_b, residual, dilation: int, relu: bool = True, out=None):
    """conv3x1 -> +shift_a -> ReLU -> conv1x3 -> scale_b/shift_b (+ residual) -> ReLU in one launch.
    Returns None when the library does not take the shape (the caller then issues the two convolutions)."""
    n, h, w, c = x.shape
    y = torch.empty_like(x) if out is None else out
    rc = query("erf_conv1d_pair", x.data_ptr(), y.data_ptr(), packed_a.data_ptr(), packed_b.data_ptr(), 0, _p(shift_a),
               _p(scale_b), _p(shift_b), _p(residual), dt(x), n, h, w, c, dilation, int(relu), _stream())
    if rc == -2:
        return None
    if rc != 0:
        raise RuntimeError(f"erf_conv1d_pair failed with status {rc}")
    return y


def nb1d_block(x, packed4, table, dilation: int, out=None):
    """A whole non_bottleneck_1d block (eval mode, BatchNorms folded) in one launch.  Returns None when the library does not
    take the shape (the caller then issues the two conv1d_pair launches).  torch.ops.erfnet_mi355.nb1d_block is the same
    launch as a registered op (it issues the two pair launches itself instead of returning None)."""
    n, h, w, c = x.shape
    y = torch.empty_like(x) if out is None else out
    rc = query("erf_nb1d_block_infer", x.data_ptr(), y.data_ptr(), packed4[0].data_ptr(), packed4[1].data_ptr(), packed4[2].data_ptr(),
               packed4[3].data_ptr(), table.data_ptr(), dt(x), n, h, w, c, dilation, _stream())
    if rc == -2:
        return None
    if rc != 0:
        raise RuntimeError(f"erf_nb1d_block_infer failed with status {rc}")
    return y


def downsampler(x, packed, scale, shift, cout: int, relu: bool = True, out=None):
    if out is None:
        return T.downsampler(x, packed, scale, shift, cout, relu)
    n, h, w, cin = x.shape
    y = out
    call("erf_downsampler", x.data_ptr(), y.data_ptr(), packed.data_ptr(), _p(scale), _p(shift), dt(x), n, h, w, cin, cout,
         int(relu), _stream())
    return y


def stem(x_nchw, w, scale, shift, code: int, relu: bool = True, out=None):
    src, n, h, w_ = image_source(x_nchw)
    if out is None:
        return T.stem(x_nchw, w, scale, shift, code, relu)
    y = out
    call("erf_stem_downsampler", x_nchw.data_ptr(), src, y.data_ptr(), w.data_ptr(), _p(scale), _p(shift), code,
         n, h, w_, int(relu), _stream())
    return y


def upsampler(x, packed, scale, shift, cout: int, relu: bool = True, out=None):
    if out is None:
        return T.upsampler(x, packed, scale, shift, cout, relu)
    n, h, w, cin = x.shape
    y = out
    call("erf_upsampler_convT3x3s2", x.data_ptr(), y.data_ptr(), packed.data_ptr(), _p(scale), _p(shift), dt(x), n, h, w,
         cin, cout, int(relu), _stream())
    return y


def output_conv(x, packed, bias, nc: int, want_logits: bool = True, logits_f32: bool = True, want_labels: bool = False,
                label_dtype=torch.int64, logits_out=None, labels_out=None):
    """Returns (logits NCHW or None, labels [N,2H,2W] or None)."""
    if logits_out is None and labels_out is None and (want_logits or want_labels):
        lo, la = T.output_conv(x, packed, bias, nc, want_logits, logits_f32,
                               (1 if label_dtype == torch.uint8 else 8) if want_labels else 0)
        return (lo if want_logits else None), (la if want_labels else None)
    n, h, w, cin = x.shape
    logits = labels = None
    if want_logits:
        logits = logits_out if logits_out is not None else torch.empty(
            (n, nc, 2 * h, 2 * w), dtype=torch.float32 if logits_f32 else x.dtype, device=x.device)
    if want_labels:
        labels = labels_out if labels_out is not None else torch.empty((n, 2 * h, 2 * w), dtype=label_dtype, device=x.device)
    call("erf_output_convT2x2s2", x.data_ptr(), packed.data_ptr(), bias.data_ptr(), _p(logits),
         int(logits is not None and logits.dtype == torch.float32), _p(labels),
         0 if labels is None else labels.element_size(), dt(x), n, h, w, nc, _stream())
    return logits, labels


def pointwise(x, packed, bias, nc: int, out_f32: bool = True):
    return T.pointwise(x, packed, bias, nc, out_f32)


# --------------------------------------------------------------------------- metrics
def argmax_c(logits: torch.Tensor, label_dtype=torch.int64) -> torch.Tensor:
    """outputs.max(1)[1] on NCHW-contiguous logits -> [N,H,W]."""
    if label_dtype not in (torch.uint8, torch.int64):
        raise TypeError("label maps are uint8 or int64")
    return T.argmax_c(logits.contiguous(), 1 if label_dtype == torch.uint8 else 8)


def iou_hist_accumulate(pred: torch.Tensor, label: torch.Tensor, nc: int, hist: torch.Tensor, relabel=None):
    """hist (int64 [nc*nc+1], device) += confusion counts of (label, pred); relabel = (olabel, nlabel): label values equal to
    olabel count as nlabel."""
    olabel, nlabel = (-1, 0) if relabel is None else relabel
    T.iou_hist_accumulate(pred.contiguous(), label.contiguous(), nc, hist, int(olabel), int(nlabel))
    return hist


def labels_from_u8(t: torch.Tensor, relabel=None) -> torch.Tensor:
    """uint8 label maps -> int64 targets, relabel = (olabel, nlabel) applied on the way (ToLabel() + Relabel of the loaders)."""
    olabel, nlabel = (-1, 0) if relabel is None else relabel
    return T.labels_from_u8(t.contiguous(), int(olabel), int(nlabel))
