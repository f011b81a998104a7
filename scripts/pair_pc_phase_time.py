"""Developer tool: where a conv1d_pair_pc launch (128 channels) spends its cycles.  Needs the library built with -DERF_DEBUG
and ERF_PAIR_DBG=256 in the environment.  usage: pair_pc_phase_time.py [dilation=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erfnet_pytorch_amd import erfnet as E, ops
from erfnet_pytorch_amd import synthetic as S
dev = torch.device("cuda:0")
c, d = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 1
blk = E.non_bottleneck_1d(c, 0.03, d)
S.load_synthetic(blk)
blk = blk.to(dev).eval()
k = blk._cached(1)
x = torch.relu(torch.randn(16, 64, 128, c, device=dev)).to(torch.bfloat16)
names = ["geo + dma issue / residual issue", "gemm + epilogue", "dma wait", "barrier", "-", "-", "-", "prologue"]
for res in (None, x):
    for rep in range(3):
        y = ops.conv1d_pair(x, k["w3"], k["t3"], k["w4"], k["s4"], k["t4"], res, d)
    torch.cuda.synchronize()
    st = y.view(torch.int64).reshape(-1)[: 256 * 8 * 10].reshape(256, 8, 10).cpu().double()
    print(f"residual={'yes' if res is not None else 'no'}: cycles per wave, entry to exit: median {st[:, :, 8].median():.0f} max {st[:, :, 8].max():.0f}")
    for i, n in enumerate(names):
        if n == "-":
            continue
        print(f"   {n:34s} producers median {st[:, :4, i].median():8.0f} max {st[:, :4, i].max():8.0f}   consumers median {st[:, 4:, i].median():8.0f} max {st[:, 4:, i].max():8.0f}")
