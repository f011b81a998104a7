#!/usr/bin/env python3
"""Where the host-fed inference leg spends its time: the pinned-memory copy alone, the pipeline on resident uint8 images, the
pipeline on resident fp32 images."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from erfnet_pytorch_amd.iouEval import iouEval

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = bench.build_model(dev, "bf16", 20)
host = torch.randint(0, 256, (16, 3, 512, 1024), dtype=torch.uint8).pin_memory()
u8 = torch.empty_like(host, device=dev)
t = torch.randint(0, 20, (16, 1, 512, 1024), device=dev)

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("h2d copy 25 MB pinned: %.3f ms" % timeit(lambda: u8.copy_(host, non_blocking=True)))
big = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
dbig = torch.empty_like(big, device=dev)
ms = timeit(lambda: dbig.copy_(big, non_blocking=True), 5)
print("h2d copy 256 MB pinned: %.3f ms = %.1f GB/s" % (ms, big.numel() / ms / 1e6))
for name, x in (("uint8 planar", u8), ("uint8 interleaved", u8.permute(0, 2, 3, 1).contiguous()), ("fp32", u8.float() / 255)):
    ev = iouEval(20)
    with torch.no_grad():
        model.evaluate_batch(x, t, ev, torch.uint8, ways=2)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"), torch.no_grad():
        model.evaluate_batch(x, t, ev, torch.uint8, ways=2)
    print("pipeline on resident %s images: %.3f ms" % (name, timeit(g.replay)))
