import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from erfnet_pytorch_amd import ops, autograd as ag
from erfnet_pytorch_amd._lib import AXIS_H, AXIS_W
from tests.golden_util import load, upstream
from tests.test_gpu_training import _make_block
dev = "cuda:0"
def rel(a, b): return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30)).item()
def nhwc(x): return x.permute(0, 2, 3, 1).contiguous()
def back(y): return y.float().permute(0, 3, 1, 2)
g = load("g2_blocks.npz")
tag = "nb_c64_d1_p3"
mod, x, seed = _make_block(tag, g)
mod.train()
x = x.to(dev)
code = 0
w4 = mod.conv1x3_2.weight.detach()
for trial in range(3):
    a3 = torch.relu(torch.randn(2, 64, 24, 40, device=dev))
    dz4 = torch.randn(2, 64, 24, 40, device=dev)
    if trial == 2:
        dz4[0, :5] = 0
    ref = F.conv_transpose2d(dz4, w4, None, padding=(0, 1)) * (a3 > 0)
    outs = []
    for rep in range(3):
        o = ag.conv1d_dgrad(nhwc(dz4), ops.pack_conv1d(w4, code, 1), nhwc(a3), None, AXIS_W, 1)
        outs.append(back(o))
    print("trial", trial, [f"{rel(o, ref):.2e}" for o in outs], "determ", torch.equal(outs[0], outs[1]), torch.equal(outs[1], outs[2]))
# now the real block, step by step
mod.__dict__["_erf_forced_mask"] = torch.from_numpy(g[tag + "_mask"])
xg = x.clone().requires_grad_(True)
y = mod(xg)
gy = upstream(tuple(y.shape), seed).to(dev)
y.backward(gy)
dx1 = xg.grad.clone()
xg2 = x.clone().requires_grad_(True)
y2 = mod(xg2)
y2.backward(gy)
print("block backward deterministic:", torch.equal(dx1, xg2.grad), rel(dx1, xg2.grad))
