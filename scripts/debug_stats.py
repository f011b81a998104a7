import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from erfnet_pytorch_amd import ops, autograd as ag
from erfnet_pytorch_amd._lib import AXIS_H, AXIS_W
dev = torch.device("cuda:0")
torch.manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    for c, shape in ((16, (2, 40, 48)), (64, (2, 24, 40)), (128, (1, 16, 24)), (16, (2, 256, 512))):
        for axis in (AXIS_W, AXIS_H):
            n, h, w = shape
            x = torch.randn(n, h, w, c, device=dev).to(dtype)
            wt = torch.randn(c, c, 1, 3, device=dev) * 0.1
            pk = ops.pack_conv1d(wt, ops.dt(x))
            b = torch.randn(c, device=dev)
            bn = torch.nn.BatchNorm2d(c, eps=1e-3).to(dev)
            z, mean, invstd, scale, beta = ag.conv1d_bn_stats(x, pk, b, axis, 1, bn, True)
            z_ref = ops.conv1d(x, pk, None, b, None, axis, 1, False)
            zf = z_ref.float().reshape(-1, c)
            m_ref = zf.double().mean(0).float()
            v_ref = zf.double().var(0, unbiased=False)
            is_ref = (1.0 / torch.sqrt(v_ref + 1e-3)).float()
            print(str(dtype)[6:], c, shape, axis, "z equal", torch.equal(z, z_ref), "mean err", (mean - m_ref).abs().max().item(),
                  "invstd relerr", ((invstd - is_ref).abs() / is_ref).max().item(), flush=True)
print("---- detail")
c, (n, h, w), dtype, axis = 16, (2, 40, 48), torch.bfloat16, AXIS_W
x = torch.randn(n, h, w, c, device=dev).to(dtype)
wt = torch.randn(c, c, 1, 3, device=dev) * 0.1
pk = ops.pack_conv1d(wt, ops.dt(x))
b = torch.randn(c, device=dev)
bn = torch.nn.BatchNorm2d(c, eps=1e-3).to(dev)
z, *_ = ag.conv1d_bn_stats(x, pk, b, axis, 1, bn, True)
z_ref = ops.conv1d(x, pk, None, b, None, axis, 1, False)
ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wt.to(dtype).float(), b, padding=(0, 1)).permute(0, 2, 3, 1)
d = (z.float() - z_ref.float()).abs()
print("ndiff", int((d > 0).sum()), "of", d.numel(), "max", d.max().item())
print("z vs torch", (z.float() - ref).abs().max().item(), "z_ref vs torch", (z_ref.float() - ref).abs().max().item())
idx = (d > 0).nonzero()[:10]
print(idx.tolist())
bad = (d > 0)
pix = bad.any(-1).nonzero()
print("bad pixels", pix.shape[0], "n:", pix[:,0].unique().tolist(), "rows:", pix[:,1].unique().tolist(), "cols:", pix[:,2].unique().tolist())
print("bad channels:", bad.any(0).any(0).any(0).nonzero().flatten().tolist())
i = pix[0].tolist()
print("at", i, "z", z[i[0], i[1], i[2]].float().tolist(), "\nref", z_ref[i[0], i[1], i[2]].float().tolist())
