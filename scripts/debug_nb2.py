import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from erfnet_pytorch_amd import ops, autograd as ag
from tests.golden_util import load, upstream
from tests.test_gpu_training import _make_block
dev = "cuda:0"
def rel(a, b): return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30)).item()
def back(y): return y.float().permute(0, 3, 1, 2)
g = load("g2_blocks.npz")
tag = "nb_c64_d1_p3"
mod, x, seed = _make_block(tag, g)
mod.train()
mod.__dict__["_erf_forced_mask"] = torch.from_numpy(g[tag + "_mask"])
rec = []
orig = ag.conv1d_dgrad
def spy(dy, packed_t, gate, residual, axis, dilation):
    out = orig(dy, packed_t, gate, residual, axis, dilation)
    torch.cuda.synchronize()
    rec.append((dy.clone(), None if gate is None else gate.clone(), None if residual is None else residual.clone(), axis, dilation, out.clone()))
    return out
ag.conv1d_dgrad = spy
xg = x.to(dev).requires_grad_(True)
y = mod(xg)
y.backward(upstream(tuple(y.shape), seed).to(dev))
ws = [mod.conv1x3_2.weight, mod.conv3x1_2.weight, mod.conv1x3_1.weight, mod.conv3x1_1.weight]
for (dy, gate, res, axis, d, out), w in zip(rec, ws):
    w = w.detach()
    pad = (d, 0) if axis == 0 else (0, d)
    dl = (d, 1) if axis == 0 else (1, d)
    ref = F.conv_transpose2d(back(dy), w, None, padding=pad, dilation=dl)
    if gate is not None: ref = ref * (back(gate) > 0)
    if res is not None: ref = ref + back(res)
    e = (back(out) - ref).abs()
    print("axis", axis, "d", d, "gate", gate is not None, "res", res is not None, "rel", rel(back(out), ref), "nan in dy", torch.isnan(dy).any().item(), "nbad", (e > 1e-3).sum().item(), (e > 1e-3).nonzero()[:5].tolist())
    if gate is not None:
        gg = back(gate)
        print("   gate stats: min", gg.min().item(), "nan", torch.isnan(gg).any().item(), "neg zero count", ((gg == 0) & (torch.signbit(gg))).sum().item())
print("---- torch autograd reference on GPU")
m = mod
mask = torch.from_numpy(g[tag + "_mask"]).to(dev)
xr = x.to(dev).clone().requires_grad_(True)
a1 = F.relu(F.conv2d(xr, m.conv3x1_1.weight, m.conv3x1_1.bias, padding=(1, 0))); a1.retain_grad()
z2 = F.conv2d(a1, m.conv1x3_1.weight, m.conv1x3_1.bias, padding=(0, 1)); z2.retain_grad()
a2 = F.relu(F.batch_norm(z2, None, None, m.bn1.weight, m.bn1.bias, True, 0.1, 1e-3)); a2.retain_grad()
a3 = F.relu(F.conv2d(a2, m.conv3x1_2.weight, m.conv3x1_2.bias, padding=(1, 0))); a3.retain_grad()
z4 = F.conv2d(a3, m.conv1x3_2.weight, m.conv1x3_2.bias, padding=(0, 1)); z4.retain_grad()
y4 = F.batch_norm(z4, None, None, m.bn2.weight, m.bn2.bias, True, 0.1, 1e-3)
out = F.relu(y4 * mask + xr)
out.backward(upstream(tuple(out.shape), seed).to(dev))
print("dz4", rel(back(rec[0][0]), z4.grad), "da3*gate (dz3)", rel(back(rec[0][5]), a3.grad * (a3 > 0)), "da2", rel(back(rec[1][5]), a2.grad),
      "dz2", rel(back(rec[2][0]), z2.grad), "dz1", rel(back(rec[2][5]), a1.grad * (a1 > 0)), "dx", rel(back(rec[3][5]), xr.grad))
print("dz3 input to call1 vs ours", rel(back(rec[1][0]), a3.grad * (a3 > 0)))
