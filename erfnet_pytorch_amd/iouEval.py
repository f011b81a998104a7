"""IoU metric with the interface of /root/reference/train/iouEval.py:8-75, counted on the GPU.

The reference builds two N x nClasses x H x W one-hot tensors per batch and sums their
products; here one pass over the two label maps fills an nClasses x nClasses confusion
histogram with LDS atomics (erf_iou_hist_accumulate), kept as int64 on the device.  tp/fp/fn
(float64 CPU tensors, as in the reference) are derived from it on access:

    tp_c = H[c][c]
    fn_c = sum_p H[c][p] - tp_c                       (prediction != c, label == c)
    fp_c = sum_{g != ignore} H[g][c] - tp_c           (prediction == c, label != c, label not ignored)

All counts are exact integers (the reference's float32 sums are exact only up to 2^24 per call).

Inputs that already have nClasses planes (one-hot or soft scores; train/iouEval.py:37-38,45-46 take them as they are)
go through erf_iou_planes_accumulate instead: tp/fp/fn sums of the plane products in float64, kept beside the histogram
and added to the counts above on access.
"""
from __future__ import annotations

import torch

from . import ops


class iouEval:

    def __init__(self, nClasses, ignoreIndex=19, relabel=None):
        """relabel = (olabel, nlabel): label-map targets are counted as after Relabel(olabel, nlabel) (eval/eval_iou.py:37:
        (255, 19)), so the raw 8-bit label images can be handed to addBatch; predictions are never relabelled."""
        self.nClasses = nClasses
        self.relabel = None if relabel is None else (int(relabel[0]), int(relabel[1]))
        if self.relabel is not None and not (0 <= self.relabel[0] and 0 <= self.relabel[1] < nClasses):
            raise ValueError(f"iouEval: relabel {relabel} must map a non-negative value to a class below {nClasses}")
        # an ignoreIndex >= nClasses means "no ignore class" (train/iouEval.py:12)
        self.ignoreIndex = ignoreIndex if nClasses > ignoreIndex else -1
        if nClasses > 32:
            raise ValueError("iouEval: at most 32 classes are supported by the histogram kernel")
        self._hist = None
        self.reset()

    def reset(self):
        self._hist = None
        self._host = torch.zeros(self.nClasses * self.nClasses + 1, dtype=torch.int64)
        self._planes = None                        # float64 [3, classes] on the device: tp / fp / fn of plane-form batches
        self._planes_host = None

    # -- counting ---------------------------------------------------------------------------
    def addBatch(self, x, y):  # x = predictions, y = targets; both [N,1,H,W] (or [N,H,W]) label maps
        ops.require_gpu(x if x.is_cuda else y, "iouEval.addBatch")
        dev = x.device if x.is_cuda else y.device
        x, y = x.to(dev), y.to(dev)           # the reference moves both to the GPU when either is there (:27-29)
        if x.dim() == 4 and x.size(1) != 1 or y.dim() == 4 and y.size(1) != 1:
            return self._add_planes(x, y, dev)
        self.reserve(dev)
        ops.iou_hist_accumulate(x, y, self.nClasses, self._hist, self.relabel)

    def reserve(self, dev):
        """Allocates (and clears) the device counters for label-map inputs on the current stream.  addBatch does this by
        itself; a caller that is about to count from several streams at once calls it first, on the stream they fork from."""
        dev = torch.device(dev)
        if self._hist is None or self._hist.device != dev:
            self._flush()
            self._hist = torch.zeros(self.nClasses * self.nClasses + 1, dtype=torch.int64, device=dev)

    def _add_planes(self, x, y, dev):
        """At least one input comes as N x nClasses x H x W planes (train/iouEval.py:32-46 `else` branches)."""
        from ._lib import call, size_query

        def prep(t, who):
            if t.dim() == 4 and t.size(1) != 1:
                if t.size(1) != self.nClasses:
                    raise ValueError(f"iouEval.addBatch: {who} has {t.size(1)} planes, expected 1 or nClasses = {self.nClasses}")
                return t.float().contiguous(), 0
            if t.dtype not in (torch.uint8, torch.int64):
                t = t.long()
            return t.contiguous(), (1 if t.dtype == torch.int64 else 2)
        xs, xk = prep(x, "x")
        ys, yk = prep(y, "y")
        n, hw = x.size(0), x.shape[-1] * x.shape[-2]
        if y.size(0) != n or y.shape[-1] * y.shape[-2] != hw:
            raise ValueError("prediction and label maps differ in size")
        classes = self.nClasses if self.ignoreIndex == -1 else self.ignoreIndex
        if self.ignoreIndex not in (-1, self.nClasses - 1):
            # the reference keeps planes [0, ignoreIndex) but nClasses - 1 accumulators (:16,50-51): it fails on this too
            raise ValueError("iouEval: plane-form inputs need ignoreIndex == nClasses - 1 or no ignore class")
        if self._planes is None or self._planes.device != dev:
            self._flush_planes()
            self._planes = torch.zeros(3, classes, dtype=torch.float64, device=dev)
        ws = torch.empty(size_query("erf_iou_planes_workspace_bytes", self.nClasses), dtype=torch.uint8, device=dev)
        call("erf_iou_planes_accumulate", xs.data_ptr(), xk, ys.data_ptr(), yk, n, self.nClasses, hw, self.ignoreIndex,
             self._planes.data_ptr(), ws.data_ptr(), ops._stream())

    def _flush_planes(self):
        if self._planes is not None:
            h = self._planes.cpu()
            self._planes_host = h if self._planes_host is None else self._planes_host + h
            self._planes = None

    def _plane_sums(self):
        tot = self._planes_host
        if self._planes is not None:
            h = self._planes.cpu()
            tot = h if tot is None else tot + h
        return tot

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
        tp = diag.clone()
        extra = self._plane_sums()
        if extra is not None:
            tp, col, fn = tp + extra[0], col + extra[1], fn + extra[2]
        return tp, col, fn

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
