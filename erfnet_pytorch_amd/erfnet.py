"""ERFNet behind the reference's nn.Module surface, computed by libermi355.so on MI355X.

Class names, constructor signatures, attribute names and therefore state_dict keys/shapes
mirror /root/reference/train/erfnet.py (DownsamplerBlock :11, non_bottleneck_1d :39,
Encoder :77, UpsamplerBlock :146, Decoder :157, ERFNet :271; `Net` is the upstream alias still
used at train/train_for_traversability.py:373).  The torch.nn layers held by each block are
PARAMETER CONTAINERS only (same registration order and default initialisation as the
reference); their own forward is never called.  Every forward pass runs the hand-written
gfx950 kernels: the inference path below when autograd is off, the autograd path of
erfnet_pytorch_amd.autograd when gradients are required.

Tensors at module boundaries are logical [N,C,H,W] like the reference's; feature maps are
backed by channels_last memory, logits are NCHW-contiguous.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import AXIS_H, AXIS_W, BF16, F32

BN_EPS = 1e-3
FUSE_PAIRS = True     # inference: one launch per conv3x1 -> conv1x3 pair where the library supports it
# inference: channel counts whose dilation-1 non_bottleneck_1d blocks run as ONE launch (bit-identical to the pair path):
# 16 (one workgroup per tile, csrc/nb1d_block16.hip) and 64 (a stream of rows through four wave roles, csrc/nb1d_block64s.hip).
FUSE_BLOCKS = (16, 64)


_SIDE_STREAMS = {}       # device index -> side streams of ERFNet.predict_labels(ways=k)


def _versions(mod: nn.Module):
    return tuple((t.data_ptr(), t._version) for t in list(mod.parameters(recurse=False)) + list(mod.buffers(recurse=False)))


def _bn_affine(bn: nn.BatchNorm2d, conv_bias: Optional[torch.Tensor] = None, lo: int = 0, hi: Optional[int] = None):
    """Eval-mode BatchNorm as y = s * x + t, with an optional preceding conv bias folded in."""
    s = (bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps))
    t = bn.bias.detach().float() - bn.running_mean.float() * s
    s, t = s[lo:hi], t[lo:hi]
    if conv_bias is not None:
        t = t + conv_bias.detach().float() * s
    return s.contiguous(), t.contiguous()


def _needs_autograd(mod: nn.Module, x: torch.Tensor) -> bool:
    if not torch.is_grad_enabled():
        return False
    return x.requires_grad or any(p.requires_grad for p in mod.parameters())


class _Block(nn.Module):
    """Shared plumbing: per-(dtype, parameter-version) cache of packed weights and folded affines."""

    def _cached(self, code: int):
        key = (code, self.training, ops.PARAM_EPOCH[0]) + tuple(v for m in self.modules() for v in _versions(m))
        cache = self.__dict__.get("_erf_cache")
        if cache is None or cache[0] != key:
            cache = (key, self._prepare(code))
            self.__dict__["_erf_cache"] = cache
        return cache[1]

    def forward(self, input):
        ops.require_gpu(input, type(self).__name__)
        if self.training or _needs_autograd(self, input):
            from . import autograd as ag
            return ag.block_forward(self, input)
        y = self.forward_nhwc(ops.to_nhwc(input))
        return ops.to_nchw_view(y)


class DownsamplerBlock(_Block):
    def __init__(self, ninput, noutput):
        super().__init__()
        self.conv = nn.Conv2d(ninput, noutput - ninput, (3, 3), stride=2, padding=1, bias=True)
        self.pool = nn.MaxPool2d(2, stride=2, ceil_mode=False)
        self.bn = nn.BatchNorm2d(noutput, eps=BN_EPS)
        self.ninput, self.noutput = ninput, noutput

    def _prepare(self, code):
        cc = self.noutput - self.ninput
        s, t = _bn_affine(self.bn)
        t = t.clone()
        t[:cc] += self.conv.bias.detach().float() * s[:cc]
        if self.ninput == 3:
            w = self.conv.weight.detach().float().contiguous()
        else:
            w = ops.pack_downsampler(self.conv.weight, code)
        return {"w": w, "s": s, "t": t}

    def forward(self, input):
        if self.ninput != 3:
            return super().forward(input)
        ops.require_gpu(input, "DownsamplerBlock")
        if self.training or _needs_autograd(self, input):
            from . import autograd as ag
            return ag.block_forward(self, input)
        return ops.to_nchw_view(self.forward_image(input, ops.image_code(input)))

    def forward_image(self, image_nchw, code):
        """3 -> 16 initial block: reads the NCHW image directly (fp32 / bf16, or 8-bit images [N,3,H,W] / [N,H,W,3], which
        it turns into value / 255 on load: ToTensor() of train/main.py:47,66)."""
        c = self._cached(code)
        return ops.stem(image_nchw.contiguous(), c["w"], c["s"], c["t"], code)

    def forward_nhwc(self, x):
        c = self._cached(ops.dt(x))
        return ops.downsampler(x, c["w"], c["s"], c["t"], self.noutput)


class non_bottleneck_1d(_Block):
    def __init__(self, chann, dropprob, dilated):
        super().__init__()
        self.conv3x1_1 = nn.Conv2d(chann, chann, (3, 1), stride=1, padding=(1, 0), bias=True)
        self.conv1x3_1 = nn.Conv2d(chann, chann, (1, 3), stride=1, padding=(0, 1), bias=True)
        self.bn1 = nn.BatchNorm2d(chann, eps=BN_EPS)
        self.conv3x1_2 = nn.Conv2d(chann, chann, (3, 1), stride=1, padding=(1 * dilated, 0), bias=True, dilation=(dilated, 1))
        self.conv1x3_2 = nn.Conv2d(chann, chann, (1, 3), stride=1, padding=(0, 1 * dilated), bias=True, dilation=(1, dilated))
        self.bn2 = nn.BatchNorm2d(chann, eps=BN_EPS)
        self.dropout = nn.Dropout2d(dropprob)
        self.chann, self.dilated = chann, dilated

    def _prepare(self, code):
        s2, t2 = _bn_affine(self.bn1, self.conv1x3_1.bias)
        s4, t4 = _bn_affine(self.bn2, self.conv1x3_2.bias)
        b1, b3 = self.conv3x1_1.bias.detach().float(), self.conv3x1_2.bias.detach().float()
        one = torch.ones_like(b1)
        return {
            # [4 convolutions][scale | shift][C] for the one-launch block kernel
            "table": torch.stack([one, b1, s2, t2, one, b3, s4, t4]).contiguous(),
            "w1": ops.pack_conv1d(self.conv3x1_1.weight, code), "t1": self.conv3x1_1.bias.detach().float().contiguous(),
            "w2": ops.pack_conv1d(self.conv1x3_1.weight, code), "s2": s2, "t2": t2,
            "w3": ops.pack_conv1d(self.conv3x1_2.weight, code), "t3": self.conv3x1_2.bias.detach().float().contiguous(),
            "w4": ops.pack_conv1d(self.conv1x3_2.weight, code), "s4": s4, "t4": t4,
        }

    def forward_nhwc(self, x):
        c = self._cached(ops.dt(x))
        d = self.dilated
        if d == 1 and x.dtype == torch.bfloat16 and self.chann in FUSE_BLOCKS:
            return ops.T.nb1d_block(x, c["w1"], c["w2"], c["w3"], c["w4"], c["table"], d)
        if FUSE_PAIRS:
            # one launch per conv3x1 -> conv1x3 pair where the library takes the shape, else two (decided inside the op)
            y = ops.T.conv1d_pair(x, c["w1"], c["t1"], c["w2"], c["s2"], c["t2"], None, 1, True)
            return ops.T.conv1d_pair(y, c["w3"], c["t3"], c["w4"], c["s4"], c["t4"], x, d, True)
        y = ops.conv1d(x, c["w1"], None, c["t1"], None, AXIS_H, 1, True)
        y = ops.conv1d(y, c["w2"], c["s2"], c["t2"], None, AXIS_W, 1, True)
        z = ops.conv1d(y, c["w3"], None, c["t3"], None, AXIS_H, d, True)
        return ops.conv1d(z, c["w4"], c["s4"], c["t4"], x, AXIS_W, d, True)


class UpsamplerBlock(_Block):
    def __init__(self, ninput, noutput):
        super().__init__()
        self.conv = nn.ConvTranspose2d(ninput, noutput, 3, stride=2, padding=1, output_padding=1, bias=True)
        self.bn = nn.BatchNorm2d(noutput, eps=BN_EPS)
        self.ninput, self.noutput = ninput, noutput

    def _prepare(self, code):
        s, t = _bn_affine(self.bn, self.conv.bias)
        return {"w": ops.pack_upsampler(self.conv.weight, code), "s": s, "t": t}

    def forward_nhwc(self, x):
        c = self._cached(ops.dt(x))
        return ops.upsampler(x, c["w"], c["s"], c["t"], self.noutput)


class Encoder(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.initial_block = DownsamplerBlock(3, 16)
        self.layers = nn.ModuleList()
        self.layers.append(DownsamplerBlock(16, 64))
        for _ in range(0, 5):
            self.layers.append(non_bottleneck_1d(64, 0.03, 1))
        self.layers.append(DownsamplerBlock(64, 128))
        for _ in range(0, 2):
            self.layers.append(non_bottleneck_1d(128, 0.3, 2))
            self.layers.append(non_bottleneck_1d(128, 0.3, 4))
            self.layers.append(non_bottleneck_1d(128, 0.3, 8))
            self.layers.append(non_bottleneck_1d(128, 0.3, 16))
        # only used in encoder-only mode (forward(..., predict=True))
        self.output_conv = nn.Conv2d(128, num_classes, 1, stride=1, padding=0, bias=True)
        self.compute_dtype: Optional[torch.dtype] = None

    def _head(self, code):
        key = (code, ops.PARAM_EPOCH[0]) + _versions(self.output_conv)
        cache = self.__dict__.get("_erf_head")
        if cache is None or cache[0] != key:
            cache = (key, ops.pack_pointwise(self.output_conv.weight.flatten(1), code),
                     self.output_conv.bias.detach().float().contiguous())
            self.__dict__["_erf_head"] = cache
        return cache[1], cache[2]

    def forward_features_nhwc(self, image, code):
        x = self.initial_block.forward_image(image, code)
        for layer in self.layers:
            x = layer.forward_nhwc(x)
        return x

    def forward(self, input, predict=False):
        ops.require_gpu(input, "Encoder")
        if self.training or _needs_autograd(self, input):
            from . import autograd as ag
            return ag.encoder_forward(self, input, predict)
        code = ops.image_code(input) if self.compute_dtype is None else (BF16 if self.compute_dtype == torch.bfloat16 else F32)
        x = self.forward_features_nhwc(input, code)
        if predict:
            w, b = self._head(code)
            return ops.pointwise(x, w, b, self.output_conv.out_channels, out_f32=ops.image_is_f32(input))
        return ops.to_nchw_view(x)


class Decoder(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.layers = nn.ModuleList()
        self.layers.append(UpsamplerBlock(128, 64))
        self.layers.append(non_bottleneck_1d(64, 0, 1))
        self.layers.append(non_bottleneck_1d(64, 0, 1))
        self.layers.append(UpsamplerBlock(64, 16))
        self.layers.append(non_bottleneck_1d(16, 0, 1))
        self.layers.append(non_bottleneck_1d(16, 0, 1))
        self.output_conv = nn.ConvTranspose2d(16, num_classes, 2, stride=2, padding=0, output_padding=0, bias=True)

    def _head(self, code):
        key = (code, ops.PARAM_EPOCH[0]) + _versions(self.output_conv)
        cache = self.__dict__.get("_erf_head")
        if cache is None or cache[0] != key:
            cache = (key, ops.pack_output_conv(self.output_conv.weight, code),
                     self.output_conv.bias.detach().float().contiguous())
            self.__dict__["_erf_head"] = cache
        return cache[1], cache[2]

    def forward_nhwc(self, x, logits_f32=True, want_logits=True, want_labels=False, label_dtype=torch.int64, labels_out=None):
        for layer in self.layers:
            x = layer.forward_nhwc(x)
        w, b = self._head(ops.dt(x))
        return ops.output_conv(x, w, b, self.output_conv.out_channels, want_logits, logits_f32, want_labels, label_dtype,
                               labels_out=labels_out)

    def forward(self, input):
        ops.require_gpu(input, "Decoder")
        if self.training or _needs_autograd(self, input):
            from . import autograd as ag
            return ag.decoder_forward(self, input)
        logits, _ = self.forward_nhwc(ops.to_nhwc(input), logits_f32=input.dtype == torch.float32)
        return logits


class ERFNet(nn.Module):
    def __init__(self, num_classes, encoder=None):  # `encoder` passes a pretrained Encoder in
        super().__init__()
        self.encoder = Encoder(num_classes) if encoder is None else encoder
        self.decoder = Decoder(num_classes)
        # None: compute in the input's dtype.  torch.bfloat16: bf16 activations/weights with fp32
        # accumulation even for fp32 images (logits stay fp32).
        self.compute_dtype: Optional[torch.dtype] = None

    def _code(self, input):
        cd = self.compute_dtype
        return ops.image_code(input) if cd is None else (BF16 if cd == torch.bfloat16 else F32)

    def forward(self, input, only_encode=False):
        ops.require_gpu(input, "ERFNet")
        if only_encode:
            if self.compute_dtype is not None:
                self.encoder.compute_dtype = self.compute_dtype
            return self.encoder.forward(input, predict=True)
        if self.training or _needs_autograd(self, input):
            from . import autograd as ag
            return ag.erfnet_forward(self, input)
        x = self.encoder.forward_features_nhwc(input, self._code(input))
        logits, _ = self.decoder.forward_nhwc(x, logits_f32=ops.image_is_f32(input))
        return logits

    @torch.no_grad()
    def predict_labels(self, input, label_dtype=torch.int64, with_logits=False, ways=1):
        """outputs.max(1)[1] without materialising the logits: the arg-max is taken from the fp32
        accumulators inside the output_conv kernel.  Returns labels [N,H,W] (and logits if asked).

        ways > 1 (labels only): the batch is cut into `ways` runs of images, each a chain of launches on its own HIP stream
        (forked from and joined to the caller's stream, so the call composes with graph capture).  Images are independent,
        so the labels are bit-identical; what changes is that one chain's launch prologues (every workgroup first loads its
        weights) and tails (the last, partly filled round of workgroups) run under the other chain's tiles."""
        ops.require_gpu(input, "ERFNet")
        if ways > 1 and not with_logits and input.shape[0] >= ways:
            return self._predict_labels_split(input, label_dtype, int(ways))
        x = self.encoder.forward_features_nhwc(input, self._code(input))
        logits, labels = self.decoder.forward_nhwc(x, logits_f32=True, want_logits=with_logits, want_labels=True,
                                                   label_dtype=label_dtype)
        return (labels, logits) if with_logits else labels

    @torch.no_grad()
    def evaluate_batch(self, input, target, evaluator, label_dtype=torch.uint8, ways=1):
        """One step of the evaluation loop (eval/eval_iou.py:88-93): labels = model(input).max(1)[1], then
        evaluator.addBatch(labels.unsqueeze(1), target).  With ways > 1 every chain of predict_labels(ways=...) counts its own
        images as soon as its labels exist (the counters are integer atomics, so the order does not matter) instead of one
        count over the whole batch after the chains have joined.  Returns the labels."""
        if ways > 1 and input.shape[0] >= ways:
            return self._predict_labels_split(input, label_dtype, int(ways), evaluator, target)
        labels = self.predict_labels(input, label_dtype)
        evaluator.addBatch(labels.unsqueeze(1), target)
        return labels

    def _predict_labels_split(self, input, label_dtype, ways, evaluator=None, target=None):
        code = self._code(input)
        # packed weights / folded affines are (re)built on the caller's stream BEFORE the chains fork: built inside one
        # chain they would be read by the others without any ordering
        for m in self.modules():
            if isinstance(m, _Block):
                m._cached(code)
        self.decoder._head(code)
        _, n, h, w = ops.image_source(input)
        labels = torch.empty((n, h, w), dtype=label_dtype, device=input.device)
        cur = torch.cuda.current_stream(input.device)
        side = _SIDE_STREAMS.setdefault(input.device.index, [])      # per device, not on the module: a module must stay copyable
        while len(side) < ways - 1:
            side.append(torch.cuda.Stream(input.device))
        per = (n + ways - 1) // ways
        runs = [(a, min(a + per, n)) for a in range(0, n, per)]

        if evaluator is not None:
            target = target.to(input.device)
            evaluator.reserve(input.device)                              # counters exist before the chains fork

        def chain(a, b):
            x = self.encoder.forward_features_nhwc(input[a:b], code)
            self.decoder.forward_nhwc(x, want_logits=False, want_labels=True, label_dtype=label_dtype, labels_out=labels[a:b])
            if evaluator is not None:
                evaluator.addBatch(labels[a:b].unsqueeze(1), target[a:b])

        for (a, b), s in zip(runs[1:], side):
            s.wait_stream(cur)                 # the input, the label buffer and the packed weights are ready
            with torch.cuda.stream(s):
                chain(a, b)
        chain(*runs[0])
        for _, s in zip(runs[1:], side):
            cur.wait_stream(s)
        return labels


Net = ERFNet
