"""ERFNet encoder + classifier used for ImageNet pre-training, on the MI355X kernels.

Mirror of /root/reference/train/erfnet_imagenet.py: the same class names, constructor signatures,
attribute names and registration order, hence the same state_dict keys in the same order
(`features.encoder.initial_block.conv.weight`, ..., `classifier.linear.bias`):

    ERFNet(num_classes).forward(x):   x [N,3,224,224]
        features.encoder     (:64-93)   DownsamplerBlock(3,16), (16,64), 5 x nb1d(64, 0.1, 1),
                                        DownsamplerBlock(64,128), 2 x nb1d(128, 0.1, {2,4,8,16}); no output_conv
        features.extralayer1 (:100)     MaxPool2d(2, stride=2)          28x28 -> 14x14
        features.extralayer2 (:101)     AvgPool2d(14, 1, 0)             14x14 -> 1x1
        classifier.linear    (:112-120) view(N,128) -> Linear(128, num_classes)

The encoder blocks are the kernels of erfnet.py; the two pools are one kernel and the Linear layer
another (erf_maxpool2_avgpool_*, erf_linear_*).  Training recipe of imagenet/main.py:81-85
(CrossEntropyLoss, SGD(0.1, momentum 0.9, weight_decay 1e-4)): `loss.CrossEntropyLoss2d()` on
logits viewed as [N,K,1,1], and either torch.optim.SGD on `parameters()` or trainer.ArenaSGD.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import erfnet as E
from . import ops
from ._lib import BF16, F32, call

DownsamplerBlock = E.DownsamplerBlock


class non_bottleneck_1d(E.non_bottleneck_1d):
    """Same block; the reference file registers both BatchNorms after the four convolutions (:28-40)."""

    def __init__(self, chann, dropprob, dilated):
        nn.Module.__init__(self)
        self.conv3x1_1 = nn.Conv2d(chann, chann, (3, 1), stride=1, padding=(1, 0), bias=True)
        self.conv1x3_1 = nn.Conv2d(chann, chann, (1, 3), stride=1, padding=(0, 1), bias=True)
        self.conv3x1_2 = nn.Conv2d(chann, chann, (3, 1), stride=1, padding=(1 * dilated, 0), bias=True, dilation=(dilated, 1))
        self.conv1x3_2 = nn.Conv2d(chann, chann, (1, 3), stride=1, padding=(0, 1 * dilated), bias=True, dilation=(1, dilated))
        self.bn1 = nn.BatchNorm2d(chann, eps=E.BN_EPS)
        self.bn2 = nn.BatchNorm2d(chann, eps=E.BN_EPS)
        self.dropout = nn.Dropout2d(dropprob)
        self.chann, self.dilated = chann, dilated


def _stream():
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


class _PoolMeanFn(torch.autograd.Function):
    """MaxPool2d(2,2) -> AvgPool2d over all pooled positions: NHWC features -> fp32 [N, C, 1, 1]."""

    @staticmethod
    def forward(ctx, x):
        xh = ops.to_nhwc(x)
        n, h, w, c = xh.shape
        feat = torch.empty(n, c, device=xh.device)
        call("erf_maxpool2_avgpool_fwd", xh.data_ptr(), ops.dt(xh), feat.data_ptr(), n, h, w, c, _stream())
        ctx.save_for_backward(xh)
        return feat.view(n, c, 1, 1)

    @staticmethod
    def backward(ctx, g):
        (xh,) = ctx.saved_tensors
        n, h, w, c = xh.shape
        g = g.reshape(n, c).float().contiguous()
        dx = torch.empty_like(xh)
        call("erf_maxpool2_avgpool_bwd", xh.data_ptr(), g.data_ptr(), dx.data_ptr(), ops.dt(xh), n, h, w, c, _stream())
        return ops.to_nchw_view(dx)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, weight, bias):
        f = f.float().contiguous()
        w = weight.detach().float().contiguous()
        n, c = f.shape
        k = w.shape[0]
        y = torch.empty(n, k, device=f.device)
        call("erf_linear_fwd", f.data_ptr(), w.data_ptr(), 0 if bias is None else bias.detach().float().contiguous().data_ptr(),
             y.data_ptr(), n, c, k, _stream())
        ctx.save_for_backward(f, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        f, w = ctx.saved_tensors
        n, c = f.shape
        k = w.shape[0]
        gy = gy.float().contiguous()
        df, dw, db = torch.empty_like(f), torch.empty_like(w), torch.empty(k, device=f.device)
        call("erf_linear_bwd", f.data_ptr(), w.data_ptr(), gy.data_ptr(), df.data_ptr(), dw.data_ptr(), db.data_ptr(), n, c, k,
             _stream())
        return df, dw, (db if ctx.has_bias else None)


class Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.initial_block = DownsamplerBlock(3, 16)
        self.layers = nn.ModuleList()
        self.layers.append(DownsamplerBlock(16, 64))
        for _ in range(0, 5):
            self.layers.append(non_bottleneck_1d(64, 0.1, 1))
        self.layers.append(DownsamplerBlock(64, 128))
        for _ in range(0, 2):
            self.layers.append(non_bottleneck_1d(128, 0.1, 2))
            self.layers.append(non_bottleneck_1d(128, 0.1, 4))
            self.layers.append(non_bottleneck_1d(128, 0.1, 8))
            self.layers.append(non_bottleneck_1d(128, 0.1, 16))
        self.compute_dtype: Optional[torch.dtype] = None

    def _code(self, input):
        cd = self.compute_dtype
        return ops.dt(input) if cd is None else (BF16 if cd == torch.bfloat16 else F32)

    def forward_features_nhwc(self, image, code):
        x = self.initial_block.forward_image(image, code)
        for layer in self.layers:
            x = layer.forward_nhwc(x)
        return x

    def forward(self, input):
        ops.require_gpu(input, "Encoder")
        if self.training or E._needs_autograd(self, input):
            from . import autograd as ag
            return ag.encoder_forward(self, input, code=self._code(input))
        return ops.to_nchw_view(self.forward_features_nhwc(input, self._code(input)))


class Features(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = Encoder()
        self.extralayer1 = nn.MaxPool2d(2, stride=2)
        self.extralayer2 = nn.AvgPool2d(14, 1, 0)

    def forward(self, input):
        output = self.encoder(input)
        n, c, h, w = output.shape
        if h != 28 or w != 28:
            # the reference's AvgPool2d(14,1,0) + view(N,128) only works on the 28x28 map of a 224x224 image
            raise ValueError(f"Features expects 224x224 images (28x28 encoder map), got a {h}x{w} map")
        return _PoolMeanFn.apply(output)


class Classifier(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.linear = nn.Linear(128, num_classes)

    def forward(self, input):
        output = input.view(input.size(0), 128)
        return _LinearFn.apply(output, self.linear.weight, self.linear.bias)


class ERFNet(nn.Module):
    def __init__(self, num_classes):
        super().__init__()
        self.features = Features()
        self.classifier = Classifier(num_classes)

    @property
    def compute_dtype(self):
        return self.features.encoder.compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, value):
        self.features.encoder.compute_dtype = value

    def forward(self, input):
        ops.require_gpu(input, "ERFNet")
        output = self.features(input)
        return self.classifier(output)
