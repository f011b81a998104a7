"""Median per-block time of the one-launch block kernel and of the two-pair-launch path.
usage: block_time.py [channels=64] -- shapes: 64 -> 16x128x256, 16 -> 16x256x512 (the benchmark's maps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erfnet_pytorch_amd import erfnet as E
from erfnet_pytorch_amd import synthetic as S
dev = torch.device("cuda:0")
c = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hw = (128, 256) if c == 64 else (256, 512)
blk = E.non_bottleneck_1d(c, 0.03, 1)
S.load_synthetic(blk)
blk = blk.to(dev).eval()
E.FUSE_BLOCKS = (16, 64)
x = torch.relu(torch.randn(16, *hw, c, device=dev)).to(torch.bfloat16)
for fused in ((16, 64), ()):
  E.FUSE_BLOCKS = fused
  ts = []
  with torch.no_grad():
    for i in range(14):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = blk.forward_nhwc(x); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    if fused:
        y_fused = y
  ts = sorted(ts[4:])
  print(f"{'one launch' if fused else 'two pair launches'}: median {ts[len(ts)//2]:.1f} us  min {ts[0]:.1f}", flush=True)
print("bit-identical:", torch.equal(y, y_fused))
