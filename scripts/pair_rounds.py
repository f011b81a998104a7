import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erfnet_pytorch_amd import ops
dev = torch.device("cuda:0")
def bench(c, n, h, w, d, reps=30):
    x = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16); y = torch.empty_like(x)
    wt = torch.randn(c, c, 3, 1, device=dev) * 0.05
    pk = ops.pack_conv1d(wt, ops.dt(x)); sh = torch.randn(c, device=dev)
    for _ in range(3): ops.conv1d_pair(x, pk, sh, pk, sh, sh, None, d, True, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.conv1d_pair(x, pk, sh, pk, sh, sh, None, d, True, out=y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for c, h, w, per_img, grid in ((128, 64, 128, 72, 256), (64, 128, 256, 16 * 19, 512)):
    for n in (1, 2, 3, 4, 6, 7, 8, 10, 11, 12, 14, 16, 21, 28, 32):
        tiles = n * per_img
        print(f"C={c} N={n:2d} tiles={tiles:5d} rounds={-(-tiles // grid)} ({tiles / grid:.2f}) {bench(c, n, h, w, 1):7.1f} us", flush=True)
