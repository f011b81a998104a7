"""Fixed and per-row cost of the streaming 64-channel block kernel: the launch at two image heights (the rows per workgroup double,
the prologue does not), replayed from a graph of ten launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erfnet_pytorch_amd import erfnet as E, synthetic as S
dev = torch.device("cuda:0")
blk = E.non_bottleneck_1d(64, 0.03, 1)
S.load_synthetic(blk)
blk = blk.to(dev).eval()
E.FUSE_BLOCKS = (16, 64)
res = {}
for h in (128, 256):
    x = torch.relu(torch.randn(16, h, 256, 64, device=dev)).to(torch.bfloat16)
    with torch.no_grad():
        blk.forward_nhwc(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                y = blk.forward_nhwc(x)
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e2)
    res[h] = sorted(ts)[4]
rows = 16 * 5 * 128 / 256
per_row = (res[256] - res[128]) / rows
print(f"H=128: {res[128]:.1f} us  H=256: {res[256]:.1f} us  per row of a workgroup: {per_row:.3f} us  fixed: {res[128] - per_row * rows:.1f} us", flush=True)
