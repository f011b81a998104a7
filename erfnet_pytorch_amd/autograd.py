"""Differentiable (training) path: one torch.autograd.Function per ERFNet block, every forward
and backward computed by libermi355.so kernels.

Semantics follow the reference modules in training mode (/root/reference/train/erfnet.py):
BatchNorm2d with batch statistics and running-stat updates (eps 1e-3, momentum 0.1), Dropout2d
masks per (image, channel), ReLU, residual add; in eval mode with gradients enabled the running
statistics are used instead (frozen-BN fine-tuning).  Parameter gradients are fp32 tensors in
the parameters' own torch layouts whatever the activation dtype (fp32 or bf16).
"""
from __future__ import annotations

import os
import threading

import torch

from . import ops
from ._lib import AXIS_H, AXIS_W, BF16, F32, call, query, size_query

_mask_counter = [0]


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _p(t):
    return 0 if t is None else t.data_ptr()


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


def _f32(t):
    return t.detach().float().contiguous()


def _count_batch(bn):
    """`num_batches_tracked += 1` of a BatchNorm forward in training mode; a trainer that owns the model may take the
    39 one-element updates over (it sets `_erf_batch_counter` to a list the layer is appended to, and adds 1 to all of
    them with one multi-tensor call per step)."""
    pending = bn.__dict__.get("_erf_batch_counter")
    if pending is not None:
        pending.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


def _pack_now(fn, w, code, dev, *shape):
    """Pack a sampler / head weight on the spot (no current PackedBank): `fn`_packed_bytes + `fn`_pack."""
    pk = torch.empty(query(fn + "_packed_bytes", code, *shape), dtype=torch.uint8, device=dev)
    call(fn + "_pack", _f32(w).data_ptr(), pk.data_ptr(), code, *shape, _stream())
    return pk


# --------------------------------------------------------------------------- primitive wrappers
def bn_forward_stats(z, bn, training):
    """Returns (mean, invstd, scale, beta) for y = (z - mean)*scale + beta, scale = gamma*invstd; updates the
    running statistics when training."""
    c = z.shape[-1]
    dev = z.device
    gamma, beta = _f32(bn.weight), _f32(bn.bias)
    if training:
        mean = torch.empty(c, device=dev)
        invstd = torch.empty(c, device=dev)
        scale = torch.empty(c, device=dev)
        shift = torch.empty(c, device=dev)
        ws = _ws(size_query("erf_bn_workspace_bytes", c), dev)
        track = bn.track_running_stats and bn.running_mean is not None
        momentum = 0.1 if bn.momentum is None else bn.momentum
        call("erf_bn_stats_finalize", z.data_ptr(), ops.dt(z), z.numel() // c, c, gamma.data_ptr(), beta.data_ptr(),
             _p(bn.running_mean if track else None), _p(bn.running_var if track else None), momentum, bn.eps,
             mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), ws.data_ptr(), _stream())
        if track:
            _count_batch(bn)
        return mean, invstd, scale, beta
    invstd = torch.rsqrt(bn.running_var.float() + bn.eps)
    scale = gamma * invstd
    return bn.running_mean.float().contiguous(), invstd.contiguous(), scale.contiguous(), beta


def conv1d_bn_stats(x, packed, bias, axis, dilation, bn, training):
    """z = conv1d(x) + bias followed by bn_forward_stats(z): with batch statistics the sums come out of the
    convolution's epilogue (erf_conv1d_factorized_bn) instead of a second pass over z."""
    n, h, w, c = x.shape
    if training:
        dev = x.device
        z = torch.empty_like(x)
        gamma, beta = _f32(bn.weight), _f32(bn.bias)
        mean, invstd = torch.empty(c, device=dev), torch.empty(c, device=dev)
        scale, shift = torch.empty(c, device=dev), torch.empty(c, device=dev)
        ws = _ws(size_query("erf_bn_workspace_bytes", c), dev)
        track = bn.track_running_stats and bn.running_mean is not None
        momentum = 0.1 if bn.momentum is None else bn.momentum
        rc = query("erf_conv1d_factorized_bn", x.data_ptr(), z.data_ptr(), packed.data_ptr(), bias.data_ptr(), ops.dt(x), n, h, w, c,
                   axis, dilation, gamma.data_ptr(), beta.data_ptr(), _p(bn.running_mean if track else None),
                   _p(bn.running_var if track else None), momentum, bn.eps, mean.data_ptr(), invstd.data_ptr(),
                   scale.data_ptr(), shift.data_ptr(), ws.data_ptr(), _stream())
        if rc == 0:
            if track:
                _count_batch(bn)
            return z, mean, invstd, scale, beta
        if rc != -2:
            raise RuntimeError(f"erf_conv1d_factorized_bn failed with status {rc}")
    z = ops.conv1d(x, packed, None, bias, None, axis, dilation, False)
    return (z,) + tuple(bn_forward_stats(z, bn, training))


# Bias and weight gradients are leaves of the backward pass: their last step (summing per-workgroup partial results) is
# put off and done for all of them by one launch each -- when a block hands gradients back to autograd, before a
# gradient bucket is all-reduced, or when the trainer calls flush_bias_sums() after backward.
# (The backward pass runs on autograd's device thread, the trainer on the caller's: the lists are shared, per device.)
_PENDING_SUMS = {}
_PENDING_WGRAD = {}
_PENDING_LOCK = threading.Lock()


def flush_bias_sums():
    with _PENDING_LOCK:
        todo = [(dev, ent) for dev, ent in _PENDING_SUMS.items() if ent[0]]
        for dev, _ in todo:
            _PENDING_SUMS[dev] = ([], [])
        wtodo = [(dev, ent) for dev, ent in _PENDING_WGRAD.items() if ent]
        for dev, _ in wtodo:
            _PENDING_WGRAD[dev] = []
    for dev, (rows, _keep) in todo:
        table = torch.tensor(rows, dtype=torch.int64)
        with torch.cuda.device(dev):
            call("erf_channel_sum_finalize_many", table.data_ptr(), len(rows), _stream())
    for dev, ents in wtodo:
        with torch.cuda.device(dev):
            table = torch.cat([e[0] for e in ents])
            call("erf_wgrad_reduce_many", table.data_ptr(), len(ents), _stream())


def conv1d_dgrad_bias(dy, packed_t, gate, residual, axis, dilation, dbias_out=None, defer=False):
    """dx = conv1d_dgrad(...) and the bias gradient sum_p dx[p][c] of the convolution that produced dx's tensor.
    defer: the sum is complete only after flush_bias_sums()."""
    n, h, w, c = dy.shape
    dx = torch.empty_like(dy)
    db = torch.empty(c, device=dy.device) if dbias_out is None else dbias_out
    ws = _ws(size_query("erf_bn_workspace_bytes", c), dy.device)
    if defer:
        rc = query("erf_conv1d_dgrad_bias_partial", dy.data_ptr(), dx.data_ptr(), packed_t.data_ptr(), _p(gate), _p(residual),
                   ops.dt(dy), n, h, w, c, axis, dilation, ws.data_ptr(), _stream())
        if rc > 0:
            with _PENDING_LOCK:
                rows, keep = _PENDING_SUMS.setdefault(dy.device.index, ([], []))
                rows.append([ws.data_ptr(), rc, c, db.data_ptr()])
                keep.append((ws, db))                    # alive until the flush has enqueued the kernel that reads them
            return dx, db
        if rc != -2:
            raise RuntimeError(f"erf_conv1d_dgrad_bias_partial failed with status {rc}")
        dx = conv1d_dgrad(dy, packed_t, gate, residual, axis, dilation)
        return dx, channel_sum(dx, db)
    rc = query("erf_conv1d_dgrad_bias", dy.data_ptr(), dx.data_ptr(), packed_t.data_ptr(), _p(gate), _p(residual), ops.dt(dy), n, h, w,
               c, axis, dilation, db.data_ptr(), ws.data_ptr(), _stream())
    if rc == 0:
        return dx, db
    if rc != -2:
        raise RuntimeError(f"erf_conv1d_dgrad_bias failed with status {rc}")
    dx = conv1d_dgrad(dy, packed_t, gate, residual, axis, dilation)
    return dx, channel_sum(dx, db)


# ---- BatchNorm backward riding on the data-gradient convolution that produces its incoming gradient -------------------
# A block's last BatchNorm (bn2 of non_bottleneck_1d, bn of the samplers) receives its gradient from the block that
# follows.  When that block is a non_bottleneck_1d, its final data-gradient convolution applies the producer's ReLU gate
# and takes the producer's two BatchNorm-backward sums in its epilogue; the producer's backward then finds dz / dgamma /
# dbeta ready.  Forward: the producer hangs a BnTail on the tensor it returns; the consumer keeps it for its backward.
# Backward: the consumer leaves a one-slot hand-off that the very next block backward takes (or discards).
FUSE_BN_BACKWARD = True                  # module switch for tests; the environment is never consulted
FUSED_BN_BACKWARD_CALLS = [0]             # diagnostic counter (tests check that the fused path is the one that runs)
_HANDOFF = threading.local()             # one slot per backward thread (autograd runs one thread per device)


class BnTail:
    __slots__ = ("mod", "prefix", "out_ptr", "z", "mean", "invstd", "gamma", "mask", "training", "gate_bits")

    def __init__(self, mod, prefix, out, z, mean, invstd, gamma, mask, training, gate_bits=None):
        self.mod, self.prefix, self.out_ptr = mod, prefix, out.data_ptr()
        self.z, self.mean, self.invstd, self.gamma, self.mask, self.training = z, mean, invstd, gamma, mask, training
        self.gate_bits = gate_bits            # ReLU gate of `out`, one byte per 16-byte chunk (bn_apply_gates), or None


def _with_tail(out, tail):
    y = ops.to_nchw_view(out)
    if FUSE_BN_BACKWARD:
        y._erf_bn_tail = tail
    return y


def _tail_of(x, xh):
    t = getattr(x, "_erf_bn_tail", None) if FUSE_BN_BACKWARD else None
    return t if t is not None and t.out_ptr == xh.data_ptr() and t.z.shape == xh.shape and t.z.dtype == xh.dtype else None


def _take_handoff(g):
    h = getattr(_HANDOFF, "slot", None)
    _HANDOFF.slot = None
    if h is not None and h[0] == g.data_ptr() and h[1] == tuple(g.shape):
        return h[2]
    return None           # nothing handed over (or not for this tensor): the caller runs its BatchNorm backward itself,
                          # which gives the same result on an already-gated gradient


def conv1d_dgrad_bn_backward(dy, packed_t, gate, residual, axis, dilation, z, mean, invstd, gamma, mask, batch_stats,
                             dgamma=None, dbeta=None, gate_bits=None):
    """dx = [gate > 0] * (conv1d_dgrad(dy) + residual), then the backward of the BatchNorm (z, mean, invstd, gamma, mask)
    whose output gradient dx is.  Returns (dx, dz, dgamma, dbeta), or None when the fused kernel does not take the shape.
    gate_bits: the gate as bn_apply_gates wrote it (read in place of the tensor when the kernel takes that form)."""
    n, h, w, c = dy.shape
    dev = dy.device
    dx, dz = torch.empty_like(dy), torch.empty_like(z)
    dgamma = torch.empty(c, device=dev) if dgamma is None else dgamma
    dbeta = torch.empty(c, device=dev) if dbeta is None else dbeta
    ws = _ws(size_query("erf_bn_workspace_bytes", c), dev)
    rc = -2
    if gate_bits is not None and GATE_BITS:
        rc = query("erf_conv1d_dgrad_bn_backward_bits", dy.data_ptr(), dx.data_ptr(), packed_t.data_ptr(), gate_bits.data_ptr(),
                   _p(residual), ops.dt(dy), n, h, w, c, axis, dilation, z.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                   gamma.data_ptr(), _p(mask), int(batch_stats), dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(),
                   _stream())
        if rc == 0:
            GATE_BITS_CALLS[0] += 1
    if rc == -2:
        rc = query("erf_conv1d_dgrad_bn_backward", dy.data_ptr(), dx.data_ptr(), packed_t.data_ptr(), gate.data_ptr(), _p(residual),
                   ops.dt(dy), n, h, w, c, axis, dilation, z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                   _p(mask), int(batch_stats), dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), _stream())
    if rc == 0:
        FUSED_BN_BACKWARD_CALLS[0] += 1
        return dx, dz, dgamma, dbeta
    if rc != -2:
        raise RuntimeError(f"erf_conv1d_dgrad_bn_backward failed with status {rc}")
    return None


def bn_apply(z, center, scale, shift, residual=None, mask=None, relu=True):
    """y = act(((z - center) * scale + shift) * mask + residual)"""
    return ops.T.bn_apply(z, center, scale, shift, residual, mask, relu)


# The ReLU gate of a BatchNorm-apply output as one byte per 16-byte chunk (bit e = [value e > 0]), written by the pass that
# produces the tensor: the data-gradient launch that applies that ReLU's backward reads the bytes instead of the tensor.
GATE_BITS = True                         # module switch for tests; the environment is never consulted
GATE_BITS_CALLS = [0]                    # diagnostic counter: fused launches that read a gate as bytes


def bn_apply_gates(z, center, scale, shift, residual=None, mask=None):
    """(y, gate bytes) with y = relu(((z - center) * scale + shift) * mask + residual); gate bytes None when switched off or the
    kernel declines the shape (then y comes from bn_apply)."""
    if GATE_BITS and z.dtype == torch.bfloat16:
        n, h, w, c = z.shape
        y = torch.empty_like(z)
        bits = torch.empty(z.numel() * z.element_size() // 16, dtype=torch.uint8, device=z.device)
        rc = query("erf_bn_apply_gates", z.data_ptr(), y.data_ptr(), _p(center), scale.data_ptr(), shift.data_ptr(), _p(residual),
                   _p(mask), ops.dt(z), n, h * w, c, 1, bits.data_ptr(), _stream())
        if rc == 0:
            return y, bits
        if rc != -2:
            raise RuntimeError(f"erf_bn_apply_gates failed with status {rc}")
    return bn_apply(z, center, scale, shift, residual, mask, True), None


def bn_backward(g, gate, mask, z, mean, invstd, gamma, batch_stats, want_g_out=False, dgamma=None, dbeta=None):
    n, h, w, c = z.shape
    dev = z.device
    dz = torch.empty_like(z)
    g_out = torch.empty_like(z) if want_g_out else None
    dgamma = torch.empty(c, device=dev) if dgamma is None else dgamma
    dbeta = torch.empty(c, device=dev) if dbeta is None else dbeta
    ws = _ws(size_query("erf_bn_workspace_bytes", c), dev)
    call("erf_bn_backward", g.data_ptr(), _p(gate), _p(mask), z.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
         gamma.data_ptr(), int(batch_stats), dz.data_ptr(), _p(g_out), dgamma.data_ptr(), dbeta.data_ptr(), ops.dt(z), n,
         h * w, c, ws.data_ptr(), _stream())
    return dz, g_out, dgamma, dbeta


def channel_sum(z, out=None):
    if out is None:
        return ops.T.channel_sum(z)
    c = z.shape[-1]
    ws = _ws(size_query("erf_bn_workspace_bytes", c), z.device)
    call("erf_channel_sum", z.data_ptr(), ops.dt(z), z.numel() // c, c, out.data_ptr(), ws.data_ptr(), _stream())
    return out


def pre_bn_bias_grad(dz, batch_stats, out=None, out_is_zero=False):
    """Gradient of a conv bias that feeds a BatchNorm directly.  With batch statistics the BatchNorm backward
    removes the per-channel mean of its output gradient, so sum_p dz[p][c] is exactly zero in exact arithmetic
    (the reference computes rounding noise around zero): write the analytic value and skip the pass over dz.
    With frozen statistics (eval mode with gradients) the sum is a real gradient."""
    if not batch_stats:
        return channel_sum(dz, out)
    c = dz.shape[-1]
    if out is None:
        return torch.zeros(c, device=dz.device)
    return out if out_is_zero else out.zero_()


def conv1d_dgrad(dy, packed_t, gate, residual, axis, dilation):
    return ops.T.conv1d_dgrad(dy, packed_t, gate, residual, axis, dilation)


def conv1d_wgrad(x, dz, wshape, axis, dilation, out=None, defer=False):
    """dW of a factorized convolution.  defer: only the contraction is launched; dW is complete after flush_bias_sums()."""
    n, h, w, c = x.shape
    if defer:
        dw = torch.empty(wshape, device=x.device) if out is None else out
        nb = size_query("erf_wgrad_workspace_bytes", c, c, 3, n * h * w)
        ws = _ws(nb, x.device)
        desc = torch.zeros(size_query("erf_wgrad_reduce_desc_bytes"), dtype=torch.uint8)
        call("erf_conv1d_wgrad_partial", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ops.dt(x), n, h, w, c, axis, dilation,
             ws.data_ptr(), nb, desc.data_ptr(), _stream())
        with _PENDING_LOCK:
            _PENDING_WGRAD.setdefault(x.device.index, []).append((desc, 3 * c * c, ws, dw))
        return dw
    if out is None:
        return ops.T.conv1d_wgrad(x, dz, axis, dilation).view(wshape)
    dw = out
    nb = size_query("erf_wgrad_workspace_bytes", c, c, 3, n * h * w)
    ws = _ws(nb, x.device)
    call("erf_conv1d_wgrad", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ops.dt(x), n, h, w, c, axis, dilation,
         ws.data_ptr(), nb, _stream())
    return dw


def wgrad_deferred(dev, keep, fn, *args):
    """call(fn, *args) for a weight-gradient entry point, with its partial-sum reductions left pending: dW is complete
    after flush_bias_sums().  `keep`: what must stay alive until then (workspace, dW)."""
    dsz = size_query("erf_wgrad_reduce_desc_bytes")
    descs = torch.zeros(4 * dsz, dtype=torch.uint8)
    call("erf_wgrad_reduce_record_begin", descs.data_ptr(), 4)
    try:
        call(fn, *args)
    finally:
        n = query("erf_wgrad_reduce_record_end")
    if n:
        with _PENDING_LOCK:
            pend = _PENDING_WGRAD.setdefault(dev.index, [])
            for i in range(n):
                pend.append((descs[i * dsz:(i + 1) * dsz], 0, keep))


def conv1d_wgrad_block(jobs):
    """dW of several factorized convolutions on tensors of one shape -- jobs: (x, dz, dw_out, axis, dilation) -- by one
    contraction launch where the library takes the shape (bf16, C in {16, 64, 128}), else layer by layer.  Complete after
    flush_bias_sums()."""
    x0 = jobs[0][0]
    n, h, w, c = x0.shape
    dev = x0.device
    nb = size_query("erf_wgrad_workspace_bytes", c, c, 3, n * h * w)
    dsz = size_query("erf_wgrad_reduce_desc_bytes")
    wss = [_ws(nb, dev) for _ in jobs]
    table = torch.tensor([[x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), axis | (dil << 32)]
                          for (x, dz, dw, axis, dil), ws in zip(jobs, wss)], dtype=torch.int64)
    descs = torch.zeros(len(jobs) * dsz, dtype=torch.uint8)
    rc = query("erf_conv1d_wgrad_partial_many", table.data_ptr(), len(jobs), ops.dt(x0), n, h, w, c, nb, descs.data_ptr(), _stream())
    if rc == 0:
        with _PENDING_LOCK:
            pend = _PENDING_WGRAD.setdefault(dev.index, [])
            for i, ((x, dz, dw, _, _), ws) in enumerate(zip(jobs, wss)):
                pend.append((descs[i * dsz:(i + 1) * dsz], 3 * c * c, ws, dw))
        return
    if rc != -2:
        raise RuntimeError(f"erf_conv1d_wgrad_partial_many failed with status {rc}")
    for x, dz, dw, axis, dil in jobs:
        conv1d_wgrad(x, dz, tuple(dw.shape), axis, dil, dw, defer=True)


def _next_seed():
    _mask_counter[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _mask_counter[0] * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF


def new_seed_state(dev):
    """Device-resident seed state for prepare_dropout_masks(state=...): [base, masks drawn so far] as int64 bit patterns.
    The base follows torch's seed like the host-side generator of dropout_mask()."""
    base = (torch.initial_seed() * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    signed = base - (1 << 64) if base >= (1 << 63) else base
    return torch.tensor([signed, 0], dtype=torch.int64, device=dev)


def prepare_dropout_masks(mods, n, dev, state=None):
    """The masks of a forward pass for `mods` (the non_bottleneck_1d blocks with p > 0, in forward order) by ONE launch.
    Each block finds its mask in `_erf_mask_next` and uses it once; seeds advance exactly as if the blocks had drawn their
    masks one by one, so the values are the same either way.
    `state` (new_seed_state): the seeds are derived on the device from a counter that lives there and moves on with every
    call -- nothing in the launch changes from step to step on the host, which is what a captured graph needs."""
    import numpy as np
    todo = [m for m in mods if m.__dict__.get("_erf_forced_mask") is None]
    if not todo or len(todo) > 32:
        return
    counts = np.array([n * m.conv3x1_1.out_channels for m in todo], dtype=np.int32)
    ps = np.array([m.dropout.p for m in todo], dtype=np.float32)
    buf = torch.empty(int(counts.sum()), device=dev)
    if state is not None:
        call("erf_dropout2d_mask_many_counter", buf.data_ptr(), len(todo), counts.ctypes.data, ps.ctypes.data, state.data_ptr(), _stream())
    else:
        seeds = np.array([_next_seed() for _ in todo], dtype=np.uint64)
        call("erf_dropout2d_mask_many", buf.data_ptr(), len(todo), counts.ctypes.data, ps.ctypes.data, seeds.ctypes.data, _stream())
    off = 0
    for m, cnt in zip(todo, counts.tolist()):
        m.__dict__["_erf_mask_next"] = (buf[off:off + cnt].view(n, -1), n)
        off += cnt


def dropout_mask(n, c, p, dev, forced=None, mod=None):
    if forced is not None:
        return forced.to(dev, torch.float32).reshape(n, c).contiguous()
    ready = mod.__dict__.pop("_erf_mask_next", None) if mod is not None else None
    if ready is not None and ready[1] == n and ready[0].shape[1] == c and ready[0].device == dev:
        return ready[0]
    seed = _next_seed()
    return ops.T.dropout2d_mask(torch.empty(0, device=dev), n, c, float(p), seed - (1 << 64) if seed >= (1 << 63) else seed)


class _Sink:
    """Where a block's parameter gradients go.  Default: fresh tensors returned to autograd (plain
    torch optimisers work).  When a trainer attached `_erf_grad_sink` (parameter name -> view of its
    flat fp32 gradient arena) to the module, the kernels write straight into the arena, autograd gets
    None for those parameters, and `_erf_after_backward` (if set) is called once the block's gradients
    are enqueued -- the hook gradient-bucket all-reduces hang on.  An owner of a sink calls flush_bias_sums()
    after backward(): the convolution-bias gradients are complete only then."""

    def __init__(self, mod, claim=True):
        """claim=False: only look the views up (a block writing a NEIGHBOUR's BatchNorm gradients on its behalf)."""
        self.views = mod.__dict__.get("_erf_grad_sink") if mod is not None else None
        if self.views is not None and claim:
            # The kernels OVERWRITE the arena views (torch would accumulate into .grad).  The arena's owner counts its
            # zero_grad() calls in `_erf_grad_sink_epoch`; a second backward pass through the same block before the next
            # zero_grad() would silently replace the first one's gradients, so it is refused.
            epoch = mod.__dict__.get("_erf_grad_sink_epoch")
            if epoch is not None:
                if mod.__dict__.get("_erf_grad_sink_written") == epoch[0]:
                    raise RuntimeError(
                        f"{type(mod).__name__}: second backward pass into a trainer-owned gradient arena before zero_grad(); "
                        "gradient accumulation (several backward calls per step, auxiliary losses, a block applied twice in one "
                        "graph) is not supported on a model owned by Trainer / ArenaSGD -- train it through .step() only")
                mod.__dict__["_erf_grad_sink_written"] = epoch[0]
        # the owner of the arena clears it before every backward pass: analytic zeros need no fill of their own
        self.zeroed = self.views is not None and bool(mod.__dict__.get("_erf_grad_sink_zeroed"))
        self.after = mod.__dict__.get("_erf_after_backward") if mod is not None else None

    def out(self, name, shape, dev):
        if self.views is not None:
            return self.views[name]
        return torch.empty(shape, device=dev)

    def ret(self, *grads):
        if self.views is not None:
            if self.after is not None:
                flush_bias_sums()                        # the bucket all-reduce reads the bias gradients
                self.after()
            return tuple(None for _ in grads)
        flush_bias_sums()                                # autograd consumes the gradients
        return grads


def _grad_nhwc(g, like):
    """Incoming gradient (logical NCHW, any strides / dtype) -> NHWC tensor of the activations' dtype."""
    return ops.to_nhwc(g, like.dtype)


# --------------------------------------------------------------------------- non_bottleneck_1d
class _NB1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, w1, b1, w2, b2, g1, be1, w3, b3, w4, b4, g2, be2):
        xh = ops.to_nhwc(x)
        code = ops.dt(xh)
        d = mod.dilated
        training = mod.training
        a1 = ops.conv1d(xh, ops.packed_conv1d(mod, "conv3x1_1", w1, code), None, _f32(b1), None, AXIS_H, 1, True)
        z2, mean1, invstd1, s1, t1 = conv1d_bn_stats(a1, ops.packed_conv1d(mod, "conv1x3_1", w2, code), _f32(b2), AXIS_W, 1,
                                                     mod.bn1, training)
        a2, a2_bits = bn_apply_gates(z2, mean1, s1, t1)
        a3 = ops.conv1d(a2, ops.packed_conv1d(mod, "conv3x1_2", w3, code), None, _f32(b3), None, AXIS_H, d, True)
        z4, mean2, invstd2, s2, t2 = conv1d_bn_stats(a3, ops.packed_conv1d(mod, "conv1x3_2", w4, code), _f32(b4), AXIS_W, d,
                                                     mod.bn2, training)
        mask = None
        if training and mod.dropout.p != 0:
            mask = dropout_mask(xh.shape[0], xh.shape[3], mod.dropout.p, xh.device, mod.__dict__.get("_erf_forced_mask"), mod)
        out, out_bits = bn_apply_gates(z4, mean2, s2, t2, residual=xh, mask=mask)
        ctx.save_for_backward(xh, a1, z2, a2, a3, z4, out, mean1, invstd1, mean2, invstd2, w1, w2, w3, w4, g1, g2)
        ctx.mask, ctx.d, ctx.training, ctx.mod = mask, d, training, mod
        ctx.a2_bits = a2_bits
        ctx.tail = _tail_of(x, xh)
        return _with_tail(out, BnTail(mod, "bn2", out, z4, mean2, invstd2, g2, mask, training, out_bits))

    @staticmethod
    def backward(ctx, gout):
        xh, a1, z2, a2, a3, z4, out, mean1, invstd1, mean2, invstd2, w1, w2, w3, w4, g1, g2 = ctx.saved_tensors
        code, d = ops.dt(xh), ctx.d
        g = _grad_nhwc(gout, xh)
        sk, dev, c = _Sink(ctx.mod), xh.device, xh.shape[3]
        o = lambda name, shape: sk.out(name, shape, dev)
        ready = _take_handoff(g)
        if ready is not None:                 # the block after this one gated g and ran this bn2's backward already
            dz4, dg2, dbe2 = ready
            g_res = g
        else:
            dz4, g_res, dg2, dbe2 = bn_backward(g, out, ctx.mask, z4, mean2, invstd2, _f32(g2), ctx.training, want_g_out=True,
                                                dgamma=o("bn2.weight", (c,)), dbeta=o("bn2.bias", (c,)))
        db4 = pre_bn_bias_grad(dz4, ctx.training, o("conv1x3_2.bias", (c,)), sk.zeroed)
        dw4, dw3 = o("conv1x3_2.weight", w4.shape), o("conv3x1_2.weight", w3.shape)
        dw2, dw1 = o("conv1x3_1.weight", w2.shape), o("conv3x1_1.weight", w1.shape)
        dz3, db3 = conv1d_dgrad_bias(dz4, ops.packed_conv1d(ctx.mod, "conv1x3_2", w4, code, 1), a3, None, AXIS_W, d,
                                     o("conv3x1_2.bias", (c,)), defer=True)
        fused = conv1d_dgrad_bn_backward(dz3, ops.packed_conv1d(ctx.mod, "conv3x1_2", w3, code, 1), a2, None, AXIS_H, d, z2,
                                         mean1, invstd1, _f32(g1), None, ctx.training, o("bn1.weight", (c,)),
                                         o("bn1.bias", (c,)), gate_bits=ctx.a2_bits) if FUSE_BN_BACKWARD else None
        if fused is not None:
            _, dz2, dg1, dbe1 = fused
        else:
            da2 = conv1d_dgrad(dz3, ops.packed_conv1d(ctx.mod, "conv3x1_2", w3, code, 1), None, None, AXIS_H, d)
            dz2, _, dg1, dbe1 = bn_backward(da2, a2, None, z2, mean1, invstd1, _f32(g1), ctx.training,
                                            dgamma=o("bn1.weight", (c,)), dbeta=o("bn1.bias", (c,)))
        db2 = pre_bn_bias_grad(dz2, ctx.training, o("conv1x3_1.bias", (c,)), sk.zeroed)
        dz1, db1 = conv1d_dgrad_bias(dz2, ops.packed_conv1d(ctx.mod, "conv1x3_1", w2, code, 1), a1, None, AXIS_W, 1,
                                     o("conv3x1_1.bias", (c,)), defer=True)
        t, fused = ctx.tail, None
        if t is not None:                     # this block's input is the ReLU output of a BatchNorm: run that layer's backward here
            psk = _Sink(t.mod, claim=False)
            fused = conv1d_dgrad_bn_backward(dz1, ops.packed_conv1d(ctx.mod, "conv3x1_1", w1, code, 1), xh, g_res, AXIS_H, 1, t.z,
                                             t.mean, t.invstd, _f32(t.gamma), t.mask, t.training,
                                             psk.out(t.prefix + ".weight", (c,), dev), psk.out(t.prefix + ".bias", (c,), dev),
                                             gate_bits=t.gate_bits)
        if fused is not None:
            dx = fused[0]
            _HANDOFF.slot = (dx.data_ptr(), tuple(dx.shape), fused[1:])
        else:
            dx = conv1d_dgrad(dz1, ops.packed_conv1d(ctx.mod, "conv3x1_1", w1, code, 1), None, g_res, AXIS_H, 1)
        # the block's four weight gradients are leaves: one launch for all of them, behind the data path
        conv1d_wgrad_block([(a3, dz4, dw4, AXIS_W, d), (a2, dz3, dw3, AXIS_H, d), (a1, dz2, dw2, AXIS_W, 1), (xh, dz1, dw1, AXIS_H, 1)])
        return (ops.to_nchw_view(dx), None) + sk.ret(dw1, db1, dw2, db2, dg1, dbe1, dw3, db3, dw4, db4, dg2, dbe2)


# --------------------------------------------------------------------------- DownsamplerBlock
class _DownFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, w, b, gamma, beta):
        if mod.ninput != 3:
            code = ops.dt(x)
        else:
            code = mod.__dict__.get("_erf_code")
            if code is None:
                code = ops.image_code(x)
        cin, cout = mod.ninput, mod.noutput
        cc = cout - cin
        shift = torch.zeros(cout, device=x.device)
        shift[:cc] = b.detach().float()
        if cin == 3:
            img = x.contiguous()
            z = ops.stem(img, _f32(w), None, shift, code, relu=False)
            xh = img
        else:
            xh = ops.to_nhwc(x)
            z = ops.downsampler(xh, ops.packed_generic(mod, "fwd", w, code, lambda: ops.pack_downsampler(w, code)), None, shift, cout,
                                relu=False)
        mean, invstd, s, t = bn_forward_stats(z, mod.bn, mod.training)
        out = bn_apply(z, mean, s, t)
        ctx.save_for_backward(xh, z, out, mean, invstd, w, gamma)
        ctx.cin, ctx.cout, ctx.training, ctx.code, ctx.mod = cin, cout, mod.training, code, mod
        ctx.need_dx = x.requires_grad
        return _with_tail(out, BnTail(mod, "bn", out, z, mean, invstd, gamma, None, mod.training))

    @staticmethod
    def backward(ctx, gout):
        xh, z, out, mean, invstd, w, gamma = ctx.saved_tensors
        cin, cout, code = ctx.cin, ctx.cout, ctx.code
        cc = cout - cin
        dev = z.device
        g = _grad_nhwc(gout, z)
        sk = _Sink(ctx.mod)
        ready = _take_handoff(g)
        if ready is not None:
            dz, dgamma, dbeta = ready
        else:
            dz, _, dgamma, dbeta = bn_backward(g, out, None, z, mean, invstd, _f32(gamma), ctx.training,
                                               dgamma=sk.out("bn.weight", (cout,), dev), dbeta=sk.out("bn.bias", (cout,), dev))
        db = sk.out("conv.bias", (cc,), dev)
        if ctx.training:
            if not sk.zeroed:
                db.zero_()                               # analytic value, see pre_bn_bias_grad
        else:
            db.copy_(channel_sum(dz)[:cc])
        dw = sk.out("conv.weight", w.shape, dev)
        n, ho, wo, _ = z.shape
        if cin == 3:
            cpad = 8 if code == BF16 else 4
            xpad = torch.empty((n, 2 * ho, 2 * wo, cpad), dtype=z.dtype, device=dev)
            call("erf_image_to_nhwc_padded", xh.data_ptr(), ops.image_source(xh)[0], xpad.data_ptr(), code, n, 2 * ho, 2 * wo, _stream())
            nb = size_query("erf_wgrad_workspace_bytes", 13, cpad, 9, n * ho * wo)
            ws = _ws(nb, dev)
            wgrad_deferred(dev, (ws, dw, xpad), "erf_stem_wgrad", xpad.data_ptr(), dz.data_ptr(), dw.data_ptr(), code, n, 2 * ho,
                           2 * wo, ws.data_ptr(), nb, _stream())
            return (None, None) + sk.ret(dw, db, dgamma, dbeta)
        nb = size_query("erf_wgrad_workspace_bytes", cc, cin, 9, n * ho * wo)
        ws = _ws(nb, dev)
        wgrad_deferred(dev, (ws, dw), "erf_downsampler_wgrad", xh.data_ptr(), dz.data_ptr(), dw.data_ptr(), code, n, 2 * ho,
                       2 * wo, cin, cout, ws.data_ptr(), nb, _stream())
        dx = None
        if ctx.need_dx:
            pk = ops.packed_generic(ctx.mod, "dgrad", w, code,
                                    lambda: _pack_now("erf_downsampler_dgrad", w, code, dev, cin, cout))
            dxh = torch.empty_like(xh)
            call("erf_downsampler_dgrad", dz.data_ptr(), dxh.data_ptr(), pk.data_ptr(), code, n, 2 * ho, 2 * wo, cin, cout,
                 _stream())
            call("erf_maxpool2x2_backward", xh.data_ptr(), dz.data_ptr(), dxh.data_ptr(), code, n, 2 * ho, 2 * wo, cin, cout,
                 cc, 1, _stream())
            dx = ops.to_nchw_view(dxh)
        return (dx, None) + sk.ret(dw, db, dgamma, dbeta)


# --------------------------------------------------------------------------- UpsamplerBlock
class _UpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, w, b, gamma, beta):
        xh = ops.to_nhwc(x)
        code = ops.dt(xh)
        z = ops.upsampler(xh, ops.packed_generic(mod, "fwd", w, code, lambda: ops.pack_upsampler(w, code)), None, _f32(b), mod.noutput,
                          relu=False)
        mean, invstd, s, t = bn_forward_stats(z, mod.bn, mod.training)
        out = bn_apply(z, mean, s, t)
        ctx.save_for_backward(xh, z, out, mean, invstd, w, gamma)
        ctx.training, ctx.mod = mod.training, mod
        return _with_tail(out, BnTail(mod, "bn", out, z, mean, invstd, gamma, None, mod.training))

    @staticmethod
    def backward(ctx, gout):
        xh, z, out, mean, invstd, w, gamma = ctx.saved_tensors
        code, dev = ops.dt(xh), xh.device
        n, h, wd, cin = xh.shape
        cout = z.shape[3]
        g = _grad_nhwc(gout, z)
        sk = _Sink(ctx.mod)
        ready = _take_handoff(g)
        if ready is not None:
            dz, dgamma, dbeta = ready
        else:
            dz, _, dgamma, dbeta = bn_backward(g, out, None, z, mean, invstd, _f32(gamma), ctx.training,
                                               dgamma=sk.out("bn.weight", (cout,), dev), dbeta=sk.out("bn.bias", (cout,), dev))
        db = pre_bn_bias_grad(dz, ctx.training, sk.out("conv.bias", (cout,), dev), sk.zeroed)
        dw = sk.out("conv.weight", w.shape, dev)
        nb = 4 * size_query("erf_wgrad_workspace_bytes", cout, cin, 4, n * h * wd)      # a slice per output phase
        ws = _ws(nb, dev)
        wgrad_deferred(dev, (ws, dw), "erf_upsampler_wgrad", xh.data_ptr(), dz.data_ptr(), dw.data_ptr(), code, n, h, wd, cin, cout,
                       ws.data_ptr(), nb, _stream())
        pk = ops.packed_generic(ctx.mod, "dgrad", w, code, lambda: _pack_now("erf_upsampler_dgrad", w, code, dev, cin, cout))
        dxh = torch.empty_like(xh)
        call("erf_upsampler_dgrad", dz.data_ptr(), dxh.data_ptr(), pk.data_ptr(), 0, code, n, h, wd, cin, cout, _stream())
        return (ops.to_nchw_view(dxh), None) + sk.ret(dw, db, dgamma, dbeta)


# --------------------------------------------------------------------------- Decoder.output_conv
class _OutConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, logits_f32, mod=None):
        xh = ops.to_nhwc(x)
        code = ops.dt(xh)
        nc = w.shape[1]
        logits, _ = ops.output_conv(xh, ops.packed_generic(mod, "fwd", w, code, lambda: ops.pack_output_conv(w, code)), _f32(b), nc,
                                    True, logits_f32, False)
        ctx.save_for_backward(xh, w)
        ctx.mod = mod
        return logits

    @staticmethod
    def backward(ctx, glogits):
        xh, w = ctx.saved_tensors
        code, dev = ops.dt(xh), xh.device
        n, h, wd, _ = xh.shape
        nc = w.shape[1]
        glogits = glogits.contiguous()
        ld = size_query("erf_output_conv_patch_channels", nc)
        patches = torch.empty((n, h, wd, ld), dtype=xh.dtype, device=dev)
        call("erf_output_conv_patches", glogits.data_ptr(), ops.dt(glogits), patches.data_ptr(), code, n, h, wd, nc, _stream())
        return _outconv_backward(ctx.mod, xh, w, patches) + (None, None)


def _outconv_backward(mod, xh, w, patches):
    """Bias / weight / data gradients of Decoder.output_conv from the patch form of d logits.  Returns (dx, dw, db)."""
    code, dev = ops.dt(xh), xh.device
    n, h, wd, _ = xh.shape
    nc = w.shape[1]
    sk = _Sink(mod)
    db = sk.out("bias", (nc,), dev)
    db.copy_(channel_sum(patches)[: 4 * nc].reshape(nc, 4).sum(1))
    dw = sk.out("weight", w.shape, dev)
    nb = size_query("erf_wgrad_workspace_bytes", 4 * nc, 16, 1, n * h * wd)
    ws = _ws(nb, dev)
    wgrad_deferred(dev, (ws, dw), "erf_output_conv_wgrad", xh.data_ptr(), patches.data_ptr(), dw.data_ptr(), code, n, h, wd, nc,
                   ws.data_ptr(), nb, _stream())
    pk = ops.packed_generic(mod, "dgrad", w, code, lambda: _pack_now("erf_output_conv_dgrad", w, code, dev, nc))
    dxh = torch.empty_like(xh)
    call("erf_output_conv_dgrad", patches.data_ptr(), dxh.data_ptr(), pk.data_ptr(), 0, code, n, h, wd, nc, _stream())
    return (ops.to_nchw_view(dxh),) + sk.ret(dw, db)


class _HeadLossFn(torch.autograd.Function):
    """Decoder.output_conv and the weighted cross entropy in one pass each way: the logits exist in registers only
    (train/erfnet.py:171,179 + train/main.py:121,202).  Returns the loss  sum_p w[t_p] nll_p / sum_p w[t_p]  with both
    sums taken over the process group when one is given, exactly as loss.weighted_cross_entropy does on stored logits."""

    @staticmethod
    def forward(ctx, x, w, b, target, weight, group, mod):
        from .dp import global_weighted_mean
        xh = ops.to_nhwc(x)
        code, dev = ops.dt(xh), xh.device
        n, h, wd, _ = xh.shape
        nc = w.shape[1]
        target = target.contiguous()
        if target.dtype != torch.int64 or tuple(target.shape) != (n, 2 * h, 2 * wd):
            raise ValueError(f"expected int64 targets of shape {(n, 2 * h, 2 * wd)}, got {target.dtype} {tuple(target.shape)}")
        wt = None if weight is None else weight.detach().to(dev, torch.float32).contiguous()
        packed = ops.packed_generic(mod, "fwd", w, code, lambda: ops.pack_output_conv(w, code))
        bias = _f32(b)
        out3 = torch.empty(3, device=dev)
        ws = _ws(size_query("erf_output_conv_ce_workspace_bytes", n, h, wd), dev)
        call("erf_output_conv_ce_fwd", xh.data_ptr(), packed.data_ptr(), bias.data_ptr(), target.data_ptr(), _p(wt), out3.data_ptr(),
             ws.data_ptr(), code, n, h, wd, nc, _stream())
        num, den = global_weighted_mean(out3[0], out3[1], group)
        ctx.save_for_backward(xh, w, bias, target, den.reshape(1), packed)
        ctx.weight, ctx.mod = wt, mod
        return num / den

    @staticmethod
    def backward(ctx, gout):
        xh, w, bias, target, den, packed = ctx.saved_tensors
        code, dev = ops.dt(xh), xh.device
        n, h, wd, _ = xh.shape
        nc = w.shape[1]
        coef = (gout.reshape(1).float() / den).contiguous()
        patches = torch.empty((n, h, wd, size_query("erf_output_conv_patch_channels", nc)), dtype=xh.dtype, device=dev)
        call("erf_output_conv_ce_bwd", xh.data_ptr(), packed.data_ptr(), bias.data_ptr(), target.data_ptr(), _p(ctx.weight),
             coef.data_ptr(), patches.data_ptr(), code, n, h, wd, nc, _stream())
        return _outconv_backward(ctx.mod, xh, w, patches) + (None, None, None, None)


# --------------------------------------------------------------------------- Encoder.output_conv (encoder-only training)
class _PointwiseFn(torch.autograd.Function):
    """1x1 convolution 128 -> num_classes on the encoder's features (reference train/erfnet.py:96-97)."""

    @staticmethod
    def forward(ctx, x, w, b, logits_f32, mod=None):
        xh = ops.to_nhwc(x)
        code = ops.dt(xh)
        nc = w.shape[0]
        logits = ops.pointwise(xh, ops.packed_generic(mod, "fwd", w, code, lambda: ops.pack_pointwise(w.flatten(1), code)), _f32(b), nc,
                               out_f32=logits_f32)
        ctx.save_for_backward(xh, w)
        ctx.mod = mod
        return logits

    @staticmethod
    def backward(ctx, glogits):
        xh, w = ctx.saved_tensors
        code, dev = ops.dt(xh), xh.device
        n, h, wd, cin = xh.shape
        nc = w.shape[0]
        glogits = glogits.contiguous()
        ld = size_query("erf_pointwise_grad_channels", nc)
        g = torch.empty((n, h, wd, ld), dtype=xh.dtype, device=dev)
        call("erf_pointwise_grad_nhwc", glogits.data_ptr(), ops.dt(glogits), g.data_ptr(), code, n, h, wd, nc, _stream())
        sk = _Sink(ctx.mod)
        db = sk.out("bias", (nc,), dev)
        db.copy_(channel_sum(g)[:nc])
        dw = sk.out("weight", w.shape, dev)
        nb = size_query("erf_wgrad_workspace_bytes", nc, cin, 1, n * h * wd)
        ws = _ws(nb, dev)
        wgrad_deferred(dev, (ws, dw), "erf_pointwise_wgrad", xh.data_ptr(), g.data_ptr(), dw.data_ptr(), code, n, h, wd, cin, nc,
                       ws.data_ptr(), nb, _stream())
        pk = ops.packed_generic(ctx.mod, "dgrad", w, code, lambda: _pack_now("erf_pointwise_dgrad", w, code, dev, cin, nc))
        dxh = torch.empty_like(xh)
        call("erf_pointwise_dgrad", g.data_ptr(), dxh.data_ptr(), pk.data_ptr(), code, n, h, wd, cin, nc, _stream())
        return (ops.to_nchw_view(dxh),) + sk.ret(dw, db) + (None, None)


# --------------------------------------------------------------------------- dispatch from the modules
def block_forward(mod, x):
    from . import erfnet as E
    if isinstance(mod, E.non_bottleneck_1d):
        return _NB1dFn.apply(x, mod, mod.conv3x1_1.weight, mod.conv3x1_1.bias, mod.conv1x3_1.weight, mod.conv1x3_1.bias,
                             mod.bn1.weight, mod.bn1.bias, mod.conv3x1_2.weight, mod.conv3x1_2.bias, mod.conv1x3_2.weight,
                             mod.conv1x3_2.bias, mod.bn2.weight, mod.bn2.bias)
    if isinstance(mod, E.DownsamplerBlock):
        return _DownFn.apply(x, mod, mod.conv.weight, mod.conv.bias, mod.bn.weight, mod.bn.bias)
    if isinstance(mod, E.UpsamplerBlock):
        return _UpFn.apply(x, mod, mod.conv.weight, mod.conv.bias, mod.bn.weight, mod.bn.bias)
    raise TypeError(type(mod))


def _compute_code(owner, x):
    cd = getattr(owner, "compute_dtype", None)
    return ops.image_code(x) if cd is None else (BF16 if cd == torch.bfloat16 else F32)


def encoder_forward(enc, x, predict=False, code=None):
    enc.initial_block.__dict__["_erf_code"] = _compute_code(enc, x) if code is None else code
    y = block_forward(enc.initial_block, x)
    for layer in enc.layers:
        y = block_forward(layer, y)
    if predict:
        y = _PointwiseFn.apply(y, enc.output_conv.weight, enc.output_conv.bias, ops.image_is_f32(x), enc.output_conv)
    return y


def decoder_forward(dec, x, logits_f32=True):
    y = x
    for layer in dec.layers:
        y = block_forward(layer, y)
    return _OutConvFn.apply(y, dec.output_conv.weight, dec.output_conv.bias, logits_f32, dec.output_conv)


def erfnet_forward_loss(net, x, target, weight=None, group=None):
    """criterion(model(x), target) of train/main.py:196-202 with the last layer and the loss fused: the full-resolution
    logits tensor is never materialised.  target: [N, H, W] int64."""
    feats = encoder_forward(net.encoder, x, code=_compute_code(net, x))
    y = feats
    for layer in net.decoder.layers:
        y = block_forward(layer, y)
    oc = net.decoder.output_conv
    return _HeadLossFn.apply(y, oc.weight, oc.bias, target, weight, group, oc)


# ---- the same step cut at its exchange points (a step replayed from HIP graphs with the collectives issued between them)
def _cut(y):
    """A leaf that carries y's values (and the BatchNorm tail the next block's backward fuses): the autograd graph ends here,
    so the backward pass can be run -- and captured -- one section at a time."""
    d = y.detach().requires_grad_(True)
    tail = getattr(y, "_erf_bn_tail", None)
    if tail is not None:
        d._erf_bn_tail = tail
    return d


def erfnet_forward_sums(net, x, target, weight=None, cuts=False):
    """First part of erfnet_forward_loss: the decoder's features (autograd graph attached) and this rank's
    (sum w*nll, sum w, ratio) of the fused output layer + loss, with NO exchange: the caller all-reduces the first two numbers.
    Returns (features, sums, state for erfnet_backward_from_sums).
    cuts: the autograd graph is cut where the gradient buckets of dp.bucket_ranges end -- in front of the decoder and in
    front of the 128-channel encoder stage (encoder.layers[6]) -- and state carries the (output, leaf) pairs, last section
    first, for erfnet_backward_sections."""
    enc, dec = net.encoder, net.decoder
    enc.initial_block.__dict__["_erf_code"] = _compute_code(net, x)
    y = block_forward(enc.initial_block, x)
    joints = []
    for i, layer in enumerate(enc.layers):
        if cuts and i == 6:
            leaf = _cut(y)
            joints.append((y, leaf))
            y = leaf
        y = block_forward(layer, y)
    if cuts:
        leaf = _cut(y)
        joints.append((y, leaf))
        y = leaf
    for layer in dec.layers:
        y = block_forward(layer, y)
    oc = dec.output_conv
    w, b = oc.weight, oc.bias
    xh = ops.to_nhwc(y.detach())
    code, dev = ops.dt(xh), xh.device
    n, h, wd, _ = xh.shape
    nc = w.shape[1]
    target = target.contiguous()
    if target.dtype != torch.int64 or tuple(target.shape) != (n, 2 * h, 2 * wd):
        raise ValueError(f"expected int64 targets of shape {(n, 2 * h, 2 * wd)}, got {target.dtype} {tuple(target.shape)}")
    wt = None if weight is None else weight.detach().to(dev, torch.float32).contiguous()
    packed = ops.packed_generic(oc, "fwd", w, code, lambda: ops.pack_output_conv(w, code))
    bias = _f32(b)
    out3 = torch.empty(3, device=dev)
    ws = _ws(size_query("erf_output_conv_ce_workspace_bytes", n, h, wd), dev)
    call("erf_output_conv_ce_fwd", xh.data_ptr(), packed.data_ptr(), bias.data_ptr(), target.data_ptr(), _p(wt), out3.data_ptr(),
         ws.data_ptr(), code, n, h, wd, nc, _stream())
    return y, out3, (xh, w, bias, target, wt, packed, oc, joints[::-1])


def _head_backward(sums, state):
    """loss = sums[0] / sums[1] (the caller has summed both over the ranks) and the output layer's backward by hand, with the
    coefficient 1 / sums[1] that _HeadLossFn.backward derives from a unit loss gradient.  Returns (loss, d features)."""
    xh, w, bias, target, wt, packed, oc = state[:7]
    code, dev = ops.dt(xh), xh.device
    n, h, wd, _ = xh.shape
    nc = w.shape[1]
    den = sums[1].reshape(1)
    loss = sums[0] / sums[1]
    coef = (torch.ones(1, device=dev) / den).contiguous()
    patches = torch.empty((n, h, wd, size_query("erf_output_conv_patch_channels", nc)), dtype=xh.dtype, device=dev)
    call("erf_output_conv_ce_bwd", xh.data_ptr(), packed.data_ptr(), bias.data_ptr(), target.data_ptr(), _p(wt),
         coef.data_ptr(), patches.data_ptr(), code, n, h, wd, nc, _stream())
    grads = _outconv_backward(oc, xh, w, patches)
    if grads[1] is not None:                             # no gradient arena attached: hand the output layer's gradients over
        oc.weight.grad, oc.bias.grad = grads[1], grads[2]
    return loss, grads[0]


def erfnet_backward_from_sums(y, sums, state):
    """Second part: the loss and the whole backward pass in one go (state from erfnet_forward_sums(cuts=False))."""
    loss, g = _head_backward(sums, state)
    y.backward(g)
    return loss


def erfnet_backward_sections(y, sums, state):
    """Second part, section by section (state from erfnet_forward_sums(cuts=True)): a generator that runs the head and the
    decoder's backward, yields, runs the 128-channel encoder stage, yields, runs the rest and returns; every section ends
    with its bias / weight-gradient reductions flushed, so the gradient bucket it completes may leave at the yield.  The
    gradient crosses a cut as the very tensor the section's last kernel wrote (torch.autograd.grad hands it out without a
    copy), which keeps the BatchNorm-backward hand-off between neighbouring blocks alive across the cut.  The first value
    yielded is the loss tensor."""
    loss, g = _head_backward(sums, state)
    out = y
    for k, (produced, leaf) in enumerate(state[7]):
        (g,) = torch.autograd.grad([out], [leaf], [g])
        flush_bias_sums()
        yield loss if k == 0 else None
        out = produced
    out.backward(g)
    flush_bias_sums()


def erfnet_forward(net, x):
    feats = encoder_forward(net.encoder, x, code=_compute_code(net, x))
    return decoder_forward(net.decoder, feats, logits_f32=ops.image_is_f32(x))
