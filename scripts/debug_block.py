import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.golden_util import load, rel_err, upstream, pre_bn_bias
from tests.test_gpu_training import _make_block, _oracle_block
g = load("g2_blocks.npz")
for tag, dtype in [("nb_c64_d1_p3", torch.float32), ("nb_c64_d1_p0", torch.float32), ("nb_c128_d4_p30", torch.bfloat16), ("up_128_64", torch.bfloat16), ("nb_c64_d1_p0", torch.bfloat16)]:
    mod, x, seed = _make_block(tag, g)
    mod.train()
    if (tag + "_mask") in g.files:
        mod.__dict__["_erf_forced_mask"] = torch.from_numpy(g[tag + "_mask"])
        print("mask zeros:", int((g[tag + "_mask"] == 0).sum()), "of", g[tag + "_mask"].size)
    if dtype == torch.bfloat16: x = x.bfloat16().float()
    xg = x.to("cuda:0", dtype).requires_grad_(True)
    y = mod(xg)
    y.backward(upstream(tuple(y.shape), seed).to("cuda:0", y.dtype))
    ry, rdx, rgrads, rbufs = _oracle_block(tag, g, x, seed)
    print(tag, dtype, "y", f"{rel_err(y.float().cpu(), ry):.2e}", "dx", f"{rel_err(xg.grad.float().cpu(), rdx):.2e}")
    d = (xg.grad.float().cpu() - rdx).abs()
    print("   dx err: max", d.max().item(), "mean", d.mean().item(), "ref max", rdx.abs().max().item(), "frac>1e-3*max", (d > 1e-3 * rdx.abs().max()).float().mean().item())
    # where are the big errors
    idx = (d > 0.5 * d.max()).nonzero()
    print("   big err locations (n,c,h,w) first 6:", idx[:6].tolist())
    for k, p in mod.named_parameters():
        print("   grad", k, f"{rel_err(p.grad.cpu(), rgrads[k], 30.0 if pre_bn_bias(k) else 0.0):.2e}")
