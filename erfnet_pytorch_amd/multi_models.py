"""Multi-task ERFNet variant behind the reference's nn.Module surface (inference), computed by libermi355.so.

Class names, constructor signatures, attribute names and therefore the state_dict keys/shapes mirror
/root/reference/train/multi_models.py: DownsamplerBlock_simplified :23, Encoder :75 (three stages, two skip
outputs), Decoder_traversability :116 / Decoder_depth :148 (channel-concatenated skips, ConvTranspose2d(32, n, 2, 2)
head), Classifier :181 (conv 3x3 stride 2 + BatchNorm + Dropout2d + ReLU, global mean, Linear 64 -> 14) and
Multi_models :205.  The torch.nn layers are parameter containers only; eval-mode forward passes run the gfx950
kernels (the factorized-convolution and upsampler kernels of the segmentation network plus the generic 3x3 / 2x2
pieces of the C ABI).  Training this variant is not built: forward() in training mode or with gradients raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from ._lib import BF16, F32, call, query
from .erfnet import BN_EPS, UpsamplerBlock, _Block, _bn_affine, _needs_autograd, non_bottleneck_1d

__all__ = ["DownsamplerBlock_simplified", "non_bottleneck_1d", "Encoder", "UpsamplerBlock", "Decoder_traversability",
           "Decoder_depth", "Classifier", "Multi_models"]


T = torch.ops.erfnet_mi355          # registered by torch_ops (imported by ops)


def _concat(a, b):
    """torch.cat([a, b], 1) of the reference on NHWC maps of one dtype."""
    return T.concat_channels(a, b)


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _inference_only(mod, x):
    ops.require_gpu(x, type(mod).__name__)
    if mod.training or _needs_autograd(mod, x):
        raise NotImplementedError("erfnet_pytorch_amd.multi_models runs inference only: call .eval() and use torch.no_grad()")


def _pack_conv3x3(w, code, cin_padded):
    cout, cin = w.shape[0], w.shape[1]
    w = w.detach().float().contiguous()
    nbytes = query("erf_conv3x3_packed_bytes", code, cin_padded, cout)
    if nbytes <= 0:
        raise ValueError(f"3x3 convolution {cin}->{cout} is not a shape of the multi-task network")
    buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    call("erf_conv3x3_pack", w.data_ptr(), buf.data_ptr(), code, cin_padded, cin, cout, _stream())
    return buf


def _conv3x3(x, packed, scale, shift, cout, stride, relu):
    return T.conv3x3(x, packed, scale, shift, cout, stride, relu)


class DownsamplerBlock_simplified(_Block):
    """conv3x3 (stride 1) -> maxpool 2x2 -> BatchNorm -> ReLU (multi_models.py:23-36)."""

    def __init__(self, ninput, noutput):
        super().__init__()
        self.conv = nn.Conv2d(ninput, noutput, (3, 3), stride=1, padding=1, bias=True)
        self.pool = nn.MaxPool2d(2, stride=2, ceil_mode=False)
        self.bn = nn.BatchNorm2d(noutput, eps=BN_EPS)
        self.ninput, self.noutput = ninput, noutput

    def _prepare(self, code):
        s, t = _bn_affine(self.bn)
        cin_p = 16 if self.ninput == 3 else self.ninput
        return {"w": _pack_conv3x3(self.conv.weight, code, cin_p), "b": self.conv.bias.detach().float().contiguous(), "s": s, "t": t}

    def forward(self, input):
        _inference_only(self, input)
        if self.ninput == 3:
            return ops.to_nchw_view(self.forward_image(input, ops.dt(input)))
        return ops.to_nchw_view(self.forward_nhwc(ops.to_nhwc(input)))

    def forward_image(self, image_nchw, code):
        """3 -> 16 block: the NCHW image is widened to 16 zero-padded NHWC channels for the matrix-core convolution."""
        n, c, h, w = image_nchw.shape
        if c != 3:
            raise ValueError("the first block takes 3-channel images")
        return self.forward_nhwc(T.image_to_nhwc_channels(image_nchw.contiguous(), code, 16))

    def forward_nhwc(self, x):
        n, h, w, _ = x.shape
        if (h | w) & 1:
            raise ValueError("DownsamplerBlock_simplified needs even H and W")
        c = self._cached(ops.dt(x))
        z = _conv3x3(x, c["w"], None, c["b"], self.noutput, 1, False)
        return T.maxpool2x2_affine(z, c["s"], c["t"], True)


class Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.downblock_1 = DownsamplerBlock_simplified(3, 16)
        self.downblock_2 = nn.ModuleList()
        self.downblock_2.append(DownsamplerBlock_simplified(16, 64))
        for _ in range(0, 3):
            self.downblock_2.append(non_bottleneck_1d(64, 0.03, 1))
        self.downblock_3 = nn.ModuleList()
        self.downblock_3.append(DownsamplerBlock_simplified(64, 128))
        for _ in range(0, 1):
            self.downblock_3.append(non_bottleneck_1d(128, 0.3, 2))
            self.downblock_3.append(non_bottleneck_1d(128, 0.3, 4))
            self.downblock_3.append(non_bottleneck_1d(128, 0.3, 8))
            self.downblock_3.append(non_bottleneck_1d(128, 0.3, 16))
        self.compute_dtype = None

    def forward_nhwc(self, image, code):
        s1 = self.downblock_1.forward_image(image, code)
        x = s1
        for layer in self.downblock_2:
            x = layer.forward_nhwc(x)
        s2 = x
        for layer in self.downblock_3:
            x = layer.forward_nhwc(x)
        return x, s1, s2

    def forward(self, input, predict=False):
        _inference_only(self, input)
        code = ops.dt(input) if self.compute_dtype is None else (BF16 if self.compute_dtype == torch.bfloat16 else F32)
        return tuple(ops.to_nchw_view(t) for t in self.forward_nhwc(input, code))


class _SkipDecoder(nn.Module):
    """Shared body of the two regression decoders (identical graphs in the reference, multi_models.py:116-178)."""

    def __init__(self, num_classes=1):
        super().__init__()
        self.upblock_1 = nn.ModuleList()
        self.upblock_1.append(UpsamplerBlock(128, 64))
        self.upblock_1.append(non_bottleneck_1d(64, 0, 1))
        self.upblock_1.append(non_bottleneck_1d(64, 0, 1))
        self.upblock_2 = nn.ModuleList()
        self.upblock_2.append(UpsamplerBlock(128, 16))
        self.upblock_2.append(non_bottleneck_1d(16, 0, 1))
        self.upblock_2.append(non_bottleneck_1d(16, 0, 1))
        self.upblock_3 = nn.ModuleList()
        self.upblock_3.append(nn.ConvTranspose2d(32, num_classes, 2, stride=2, padding=0, output_padding=0, bias=True))

    def _head(self, code):
        conv = self.upblock_3[0]
        key = (code, ops.PARAM_EPOCH[0], conv.weight.data_ptr(), conv.weight._version, conv.bias.data_ptr(), conv.bias._version)
        cache = self.__dict__.get("_erf_head")
        if cache is None or cache[0] != key:
            w = conv.weight.detach().float().contiguous()
            nbytes = query("erf_convT2x2s2_packed_bytes", code, 32, conv.out_channels)
            if nbytes <= 0:
                raise ValueError("the decoder head takes at most 16 output maps")
            buf = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
            call("erf_convT2x2s2_pack", w.data_ptr(), buf.data_ptr(), code, 32, conv.out_channels, _stream())
            cache = (key, buf, conv.bias.detach().float().contiguous())
            self.__dict__["_erf_head"] = cache
        return cache[1], cache[2]

    def forward_nhwc(self, x, s1, s2, out_f32=True):
        for layer in self.upblock_1:
            x = layer.forward_nhwc(x)
        x = _concat(x, s2)                                  # channel concat of NHWC maps (multi_models.py:141)
        for layer in self.upblock_2:
            x = layer.forward_nhwc(x)
        x = _concat(x, s1)                                  # multi_models.py:145
        n, h, w, _ = x.shape
        nc = self.upblock_3[0].out_channels
        packed, bias = self._head(ops.dt(x))
        return T.convT2x2s2(x, packed, bias, nc, out_f32)

    def forward(self, input, shortcut_1, shortcut_2):
        _inference_only(self, input)
        return self.forward_nhwc(ops.to_nhwc(input), ops.to_nhwc(shortcut_1), ops.to_nhwc(shortcut_2),
                                 out_f32=input.dtype == torch.float32)


class Decoder_traversability(_SkipDecoder):
    pass


class Decoder_depth(_SkipDecoder):
    pass


class Classifier(nn.Module):
    def __init__(self, num_classes=14):
        super().__init__()
        layers = nn.ModuleList()
        layers.append(nn.Conv2d(128, 64, kernel_size=3, stride=2, padding=1))
        layers.append(nn.BatchNorm2d(64, eps=BN_EPS))
        layers.append(nn.Dropout2d(0.3))
        layers.append(nn.ReLU())
        self.classifier = nn.Sequential(*layers)
        self.linear = nn.Linear(64, num_classes)

    def _prep(self, code):
        conv, bn = self.classifier[0], self.classifier[1]
        key = (code, ops.PARAM_EPOCH[0]) + tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))
        cache = self.__dict__.get("_erf_cache")
        if cache is None or cache[0] != key:
            s, t = _bn_affine(bn, conv.bias)
            cache = (key, _pack_conv3x3(conv.weight, code, 128), s, t, self.linear.weight.detach().float().contiguous(),
                     self.linear.bias.detach().float().contiguous())
            self.__dict__["_erf_cache"] = cache
        return cache[1:]

    def forward_nhwc(self, x):
        packed, s, t, lw, lb = self._prep(ops.dt(x))
        y = _conv3x3(x, packed, s, t, 64, 2, True)           # conv + BatchNorm (folded) + ReLU; Dropout2d is the identity in eval
        return T.linear(T.spatial_mean(y), lw, lb)

    def forward(self, input):
        _inference_only(self, input)
        return self.forward_nhwc(ops.to_nhwc(input))


class Multi_models(nn.Module):
    def __init__(self, num_classes=14):
        super().__init__()
        self.encoder = Encoder()
        self.decoder_traversability = Decoder_traversability(1)
        self.decoder_depth = Decoder_depth(1)
        self.classifier = Classifier(num_classes)
        self.compute_dtype = None       # torch.bfloat16: bf16 activations from fp32 images (outputs stay fp32)

    def forward(self, input):
        _inference_only(self, input)
        code = ops.dt(input) if self.compute_dtype is None else (BF16 if self.compute_dtype == torch.bfloat16 else F32)
        feat, s1, s2 = self.encoder.forward_nhwc(input, code)
        f32 = input.dtype == torch.float32
        return (self.decoder_traversability.forward_nhwc(feat, s1, s2, out_f32=f32),
                self.decoder_depth.forward_nhwc(feat, s1, s2, out_f32=f32),
                self.classifier.forward_nhwc(feat))
