"""Debug helper: compare each training primitive with torch ops ON THE GPU (debug only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from erfnet_pytorch_amd import ops, autograd as ag
from erfnet_pytorch_amd._lib import AXIS_H, AXIS_W
dev = torch.device("cuda:0")
torch.manual_seed(0)
def rel(a, b): return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30)).item()
def nhwc(x, dt): return x.permute(0, 2, 3, 1).contiguous().to(dt)
def back(y): return y.float().permute(0, 3, 1, 2)
for dt in (torch.float32, torch.bfloat16):
    for (n, c, h, w, ax, d) in [(2, 64, 24, 40, 0, 1), (2, 64, 24, 40, 1, 1), (1, 128, 40, 40, 0, 16), (1, 128, 40, 40, 1, 16), (2, 16, 24, 48, 1, 1)]:
        x = torch.randn(n, c, h, w, device=dev).to(dt).float()
        dy = torch.randn(n, c, h, w, device=dev).to(dt).float()
        wt = (torch.randn(c, c, 3, 1, device=dev) * 0.05) if ax == 0 else (torch.randn(c, c, 1, 3, device=dev) * 0.05)
        wt_r = wt.to(dt).float()
        pad, dl = ((d, 0), (d, 1)) if ax == 0 else ((0, d), (1, d))
        xr = x.clone().requires_grad_(True); wr = wt_r.clone().requires_grad_(True)
        y = F.conv2d(xr, wr, None, padding=pad, dilation=dl)
        y.backward(dy)
        code = ops.dt(torch.empty(0, dtype=dt))
        yo = ops.conv1d(nhwc(x, dt), ops.pack_conv1d(wt, code), None, None, None, ax, d, False)
        gate = torch.randn(n, c, h, w, device=dev)
        res = torch.randn(n, c, h, w, device=dev).to(dt).float()
        dxo = ag.conv1d_dgrad(nhwc(dy, dt), ops.pack_conv1d(wt, code, 1), nhwc(gate, dt), nhwc(res, dt), ax, d)
        dx_ref = xr.grad * (gate.to(dt).float() > 0) + res
        dwo = ag.conv1d_wgrad(nhwc(x, dt), nhwc(dy, dt), wt.shape, ax, d)
        print(f"{str(dt)[6:]:8s} C={c} {h}x{w} ax={ax} d={d}: fwd {rel(back(yo), y):.2e} dgrad {rel(back(dxo), dx_ref):.2e} wgrad {rel(dwo, wr.grad):.2e}")
    # batchnorm fwd/bwd
    n, c, h, w = 2, 64, 24, 40
    z = (torch.randn(n, c, h, w, device=dev) * 1.5 + 0.3).to(dt).float()
    bn = torch.nn.BatchNorm2d(c, eps=1e-3).to(dev); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.uniform_(-0.2, 0.2)
    bn2 = torch.nn.BatchNorm2d(c, eps=1e-3).to(dev); bn2.load_state_dict(bn.state_dict())
    zr = z.clone().requires_grad_(True)
    mask = (torch.rand(n, c, device=dev) > 0.3).float() / 0.7
    resid = torch.randn(n, c, h, w, device=dev).to(dt).float()
    out = F.relu(bn(zr) * mask[:, :, None, None] + resid)
    g = torch.randn(n, c, h, w, device=dev).to(dt).float()
    out.backward(g)
    zh = nhwc(z, dt)
    mean, invstd, s, t = ag.bn_forward_stats(zh, bn2, True)
    o = ag.bn_apply(zh, mean, s, t, residual=nhwc(resid, dt), mask=mask)
    dz, g_out, dg, db = ag.bn_backward(nhwc(g, dt), o, mask, zh, mean, invstd, bn2.weight.detach(), True, want_g_out=True)
    print(f"{str(dt)[6:]:8s} bn: out {rel(back(o), out):.2e} dz {rel(back(dz), zr.grad):.2e} dgamma {rel(dg, bn.weight.grad):.2e} dbeta {rel(db, bn.bias.grad):.2e} rm {rel(bn2.running_mean, bn.running_mean):.2e} rv {rel(bn2.running_var, bn.running_var):.2e} gout {rel(back(g_out), g * (out > 0)):.2e}")
