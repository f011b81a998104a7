"""Developer tool: where a conv1d_pair launch spends its cycles (needs conv1d_pair.o built with -DERF_DEBUG and
ERF_PAIR_DBG=256 in the environment).  usage: pair_phase_time.py [channels=128] [dilation=2]"""
import os, sys
sys.path.insert(0, ".")
import torch
from erfnet_pytorch_amd import erfnet as E, ops
from erfnet_pytorch_amd import synthetic as S
dev = torch.device("cuda:0")
c = int(sys.argv[1]) if len(sys.argv) > 1 else 128
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hw = {128: (64, 128), 64: (128, 256), 16: (256, 512)}[c]
blk = E.non_bottleneck_1d(c, 0.03, d)
S.load_synthetic(blk)
blk = blk.to(dev).eval()
k = blk._cached(1)
x = torch.relu(torch.randn(16, *hw, c, device=dev)).to(torch.bfloat16)
names = ["gemm1", "barrier1", "gemm2", "barrier2", "readback", "barrier3", "dma+wait+store", "barrier4"]
for res in (None, x):
    for rep in range(3):
        y = ops.conv1d_pair(x, k["w3"], k["t3"], k["w4"], k["s4"], k["t4"], res, d)
    torch.cuda.synchronize()
    nwg = 256 * (2 if c <= 64 else 1)
    nw = {128: 8, 64: 4, 16: 4}[c]
    st = y.view(torch.int64).reshape(-1)[: nwg * nw * 10].reshape(nwg, nw, 10).cpu().double()
    tot = st[:, :, 8]
    print(f"residual={'yes' if res is not None else 'no'}: loop cycles per wave: median {tot.median():.0f}  max {tot.max():.0f};"
          f" prologue (entry to first barrier) median {st[:, :, 9].median():.0f}  max {st[:, :, 9].max():.0f}")
    for i, n in enumerate(names):
        print(f"   {n:16s} median {st[:, :, i].median():8.0f}   wave0 {st[:, 0, i].median():8.0f}  wave7 {st[:, nw - 1, i].median():8.0f}")
