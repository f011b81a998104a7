"""Times the fused conv3x1 -> conv1x3 launch at 128 channels, dilation 16 (whole-row kernel) on a 16 x 64 x 128 map,
with and without the residual, against the two single launches it replaces.  Usage: python scripts/pair_rows_time.py [batch ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from erfnet_pytorch_amd import ops  # noqa: E402


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def run(nb):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    c, d = 128, 16
    x = torch.randn(nb, 64, 128, c, device=dev).to(torch.bfloat16)
    r = torch.randn(nb, 64, 128, c, device=dev).to(torch.bfloat16)
    code = ops.dt(x)
    pa = ops.pack_conv1d(torch.randn(c, c, 3, 1, device=dev) * 0.05, code)
    pb = ops.pack_conv1d(torch.randn(c, c, 1, 3, device=dev) * 0.05, code)
    ba, sb, bb = (torch.randn(c, device=dev) * 0.1 for _ in range(3))
    for res in (None, r):
        tag = "residual" if res is not None else "plain"
        t = timed(lambda: ops.conv1d_pair(x, pa, ba, pb, sb, bb, res, d, True))

        def singles():
            m = ops.conv1d(x, pa, None, ba, None, 0, d, True)
            return ops.conv1d(m, pb, sb, bb, res, 1, d, True)
        print(f"batch {nb} {tag}: fused {t:.1f} us, two single launches {timed(singles):.1f} us", flush=True)


if __name__ == "__main__":
    for nb in ([int(a) for a in sys.argv[1:]] or [16]):
        run(nb)
