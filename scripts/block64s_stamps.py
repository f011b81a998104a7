"""Cycle stamps of the streaming 64-channel block kernel (debug build of the library: -DERF_DEBUG, loaded in place of the product
library).  usage: ERF_BLOCK_DBG=256 python scripts/block64s_stamps.py   (other bits of ERF_BLOCK_DBG switch parts of a step off)"""
import ctypes, os, sys
sys.path.insert(0, ".")
import numpy as np
import torch
from erfnet_pytorch_amd import erfnet as E, _lib
from erfnet_pytorch_amd import synthetic as S
dev = torch.device("cuda:0")
blk = E.non_bottleneck_1d(64, 0.03, 1)
S.load_synthetic(blk)
blk = blk.to(dev).eval()
E.FUSE_BLOCKS = (16, 64)
x = torch.relu(torch.randn(16, 128, 256, 64, device=dev)).to(torch.bfloat16)
# replayed from a graph of ten launches: launched one by one the host needs longer per launch than the kernel runs
with torch.no_grad():
    y = blk.forward_nhwc(x)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(10):
            y = blk.forward_nhwc(x)
    graph.replay()
    torch.cuda.synchronize()
    ts = []
    for i in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e2)
print(f"dbg={os.environ.get('ERF_BLOCK_DBG', '0')}: launch median {sorted(ts)[4]:.1f} us", flush=True)
if int(os.environ.get("ERF_BLOCK_DBG", "0")) & 256:
    lib = ctypes.CDLL(_lib.LIB_PATH)
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    assert lib.erf_debug_block64s_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf.reshape(256, 8, 8).astype(np.float64)
    names = ["slots+requests", "frag wait 0", "mfma 0 + frag wait 1", "mfma 1 + epilogue", "finish", "lgkm/vm wait", "barrier", "loop total"]
    roles = ["conv3x1_1", "conv3x1_1", "conv1x3_1", "conv1x3_1", "conv1x3_2", "conv1x3_2", "conv3x1_2", "conv3x1_2"]
    for w in range(8):
        med = np.median(st[:, w, :], axis=0)
        print(f"wave {w} {roles[w]:10s} " + "  ".join(f"{n} {v / 1e3:.1f}k" for n, v in zip(names, med)), flush=True)
