"""IoU metric with the interface of /root/reference/train/iouEval.py:8-75, counted on the GPU.

The reference builds two N x nClasses x H x W one-hot tensors per batch and sums their
products; here one pass over the two label maps fills an nClasses x nClasses confusion
histogram with LDS atomics (erf_iou_hist_accumulate), kept as int64 on the device.  tp/fp/fn
(float64 CPU tensors, as in the reference) are derived from it on access:

    tp_c = H[c][c]
    fn_c = sum_p H[c][p] - tp_c                       (prediction != c, label == c)
    fp_c = sum_{g != ignore} H[g][c] - tp_c           (prediction == c, label != c, label not ignored)

All counts are exact integers (the reference's float32 sums are exact only up to 2^24 per call).
"""
from __future__ import annotations

import torch

from . import ops


class iouEval:

    def __init__(self, nClasses, ignoreIndex=19):
        self.nClasses = nClasses
        # an ignoreIndex >= nClasses means "no ignore class" (train/iouEval.py:12)
        self.ignoreIndex = ignoreIndex if nClasses > ignoreIndex else -1
        if nClasses > 32:
            raise ValueError("iouEval: at most 32 classes are supported by the histogram kernel")
        self._hist = None
        self.reset()

    def reset(self):
        self._hist = None
        self._host = torch.zeros(self.nClasses * self.nClasses + 1, dtype=torch.int64)

    # -- counting ---------------------------------------------------------------------------
    def addBatch(self, x, y):  # x = predictions, y = targets; both [N,1,H,W] (or [N,H,W]) label maps
        if x.dim() == 4 and x.size(1) != 1 or y.dim() == 4 and y.size(1) != 1:
            raise NotImplementedError("iouEval.addBatch takes label maps (size(1) == 1); one-hot input is not supported")
        ops.require_gpu(x if x.is_cuda else y, "iouEval.addBatch")
        dev = x.device if x.is_cuda else y.device
        x, y = x.to(dev), y.to(dev)           # the reference moves both to the GPU when either is there (:27-29)
        if self._hist is None or self._hist.device != dev:
            self._flush()
            self._hist = torch.zeros(self.nClasses * self.nClasses + 1, dtype=torch.int64, device=dev)
        ops.iou_hist_accumulate(x, y, self.nClasses, self._hist)

    def _flush(self):
        if self._hist is not None:
            self._host += self._hist.cpu()
            self._hist = None

    def confusion(self) -> torch.Tensor:
        """int64 [nClasses, nClasses] confusion matrix, rows = label, columns = prediction."""
        total = self._host.clone()
        if self._hist is not None:
            total += self._hist.cpu()
        n = self.nClasses
        if int(total[n * n]) != 0:
            raise ValueError(f"iouEval: {int(total[n * n])} pixels carried a label or prediction outside [0,{n})")
        return total[: n * n].reshape(n, n)

    def _counts(self):
        h = self.confusion().double()
        classes = self.nClasses if self.ignoreIndex == -1 else self.nClasses - 1
        diag = h.diagonal()[:classes]
        fn = h.sum(dim=1)[:classes] - diag
        col = h.sum(dim=0)[:classes] - diag
        if self.ignoreIndex != -1:
            col = col - h[self.ignoreIndex, :classes]
        return diag.clone(), col, fn

    @property
    def tp(self):
        return self._counts()[0]

    @property
    def fp(self):
        return self._counts()[1]

    @property
    def fn(self):
        return self._counts()[2]

    def getIoU(self):
        tp, fp, fn = self._counts()
        num = tp
        den = tp + fp + fn + 1e-15
        iou = num / den
        return torch.mean(iou), iou  # "iou mean", "iou per class"


class colors:
    RED = '\033[31;1m'
    GREEN = '\033[32;1m'
    YELLOW = '\033[33;1m'
    BLUE = '\033[34;1m'
    MAGENTA = '\033[35;1m'
    CYAN = '\033[36;1m'
    BOLD = '\033[1m'
    UNDERLINE = '\033[4m'
    ENDC = '\033[0m'


def getColorEntry(val):
    """ANSI colour for an IoU value, same thresholds as train/iouEval.py:90-102."""
    if not isinstance(val, float):
        return colors.ENDC
    for limit, col in ((.20, colors.RED), (.40, colors.YELLOW), (.60, colors.BLUE), (.80, colors.CYAN)):
        if val < limit:
            return col
    return colors.GREEN
