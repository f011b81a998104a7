import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erfnet_pytorch_amd import ops
dev = torch.device("cuda:0")
def bench(c, n, h, w, d, reps=20, res=False):
    x = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    wt = torch.randn(c, c, 3, 1, device=dev) * 0.05
    pk = ops.pack_conv1d(wt, ops.dt(x))
    sh = torch.randn(c, device=dev)
    r = x if res else None
    for _ in range(3): ops.conv1d_pair(x, pk, sh, pk, sh, sh, r, d, True, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.conv1d_pair(x, pk, sh, pk, sh, sh, r, d, True, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flop = 2 * 2 * 3 * c * c * n * h * w
    byts = 2 * x.numel() * x.element_size()
    return us, flop / us / 1e6, byts / us / 1e3
cfgs = [(128, 16, 64, 128, 1), (128, 16, 64, 128, 2), (128, 16, 64, 128, 4), (128, 16, 64, 128, 8), (64, 16, 128, 256, 1), (16, 16, 256, 512, 1)]
tag = os.environ.get("ERF_PAIR_DBG", "0")
for (c, n, h, w, d) in cfgs:
    us, tf, gbs = bench(c, n, h, w, d)
    print(f"dbg={tag} C={c:3d} d={d:2d}: {us:8.1f} us  {tf:7.1f} TFLOP/s  {gbs:7.1f} GB/s(alg)", flush=True)
