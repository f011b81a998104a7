"""Error budget for fp32 storage with split-bf16 matrix instructions, on the gate-safe fixtures (g9_gate_safe_blocks.npz).

Not a test (pytest does not collect it): `python tests/split_bf16_budget.py` prints, per fixture, the max-norm error of the block
output, the input gradient and the worst parameter gradient against the fp32 oracle when every convolution -- forward, data
gradient and weight gradient -- is evaluated as a sum of bf16 x bf16 products with fp32 accumulation:

    1 product   x_h w_h                                  (bf16 operands, what the bf16 path does)
    3 products  x_h w_h + x_h w_l + x_l w_h              (x = x_h + x_l, the two leading bf16 pieces of an fp32 value)
    6 products  + x_h w_ll + x_ll w_h + x_l w_l          (three pieces)

The products of two bf16 values are exact in fp32 and torch's CPU convolution accumulates in fp32, which is what a bf16 matrix
instruction with an fp32 accumulator does; the summation order differs from any kernel's, as the fp32 oracle's does."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import erfnet_oracle as O                                           # noqa: E402
from tests.golden_util import load, activations, upstream                        # noqa: E402
from erfnet_pytorch_amd import synthetic as S                                    # noqa: E402


def pieces(t, n):
    out, rest = [], t
    for _ in range(n):
        h = rest.to(torch.bfloat16).to(torch.float32)
        out.append(h)
        rest = rest - h
    return out


def pairs(n):
    """index pairs (i, j) of the products kept for an n-piece split: all with i + j < n"""
    return [(i, j) for i in range(n) for j in range(n) if i + j < n]


def split_sum(op, a, b, n):
    pa, pb = pieces(a, n), pieces(b, n)
    acc = None
    for i, j in pairs(n):
        v = op(pa[i], pb[j])
        acc = v if acc is None else acc + v
    return acc


class SplitConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, n, transposed, kw):
        ctx.save_for_backward(x, w)
        ctx.n, ctx.transposed, ctx.kw = n, transposed, kw
        op = (lambda a, b: F.conv_transpose2d(a, b, None, **kw)) if transposed else (lambda a, b: F.conv2d(a, b, None, **kw))
        y = split_sum(op, x, w, n)
        return y if bias is None else y + bias.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        n, kw = ctx.n, ctx.kw
        if ctx.transposed:
            def dgrad(gg, ww):
                with torch.enable_grad():
                    xx = torch.zeros_like(x).requires_grad_(True)
                    return torch.autograd.grad(F.conv_transpose2d(xx, ww, None, **kw), xx, gg)[0]

            def wgrad(xx, gg):
                with torch.enable_grad():
                    ww = torch.zeros_like(w).requires_grad_(True)
                    return torch.autograd.grad(F.conv_transpose2d(xx, ww, None, **kw), ww, gg)[0]
        else:
            def dgrad(gg, ww):
                return torch.nn.grad.conv2d_input(x.shape, ww, gg, **kw)

            def wgrad(xx, gg):
                return torch.nn.grad.conv2d_weight(xx, w.shape, gg, **kw)
        dx = split_sum(dgrad, g, w, n)
        dw = split_sum(wgrad, x, g, n)
        return dx, dw, g.sum((0, 2, 3)), None, None, None


class SplitF:
    """Stands in for torch.nn.functional inside the oracle: the two convolutions go through SplitConv."""

    def __init__(self, n):
        self.n = n

    def __getattr__(self, name):
        return getattr(F, name)

    def conv2d(self, x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
        return SplitConv.apply(x, w, bias, self.n, False, dict(stride=stride, padding=padding, dilation=dilation, groups=groups))

    def conv_transpose2d(self, x, w, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
        return SplitConv.apply(x, w, bias, self.n, True,
                               dict(stride=stride, padding=padding, output_padding=output_padding, groups=groups, dilation=dilation))


def case(tag, g):
    prefix = str(g[tag + "__prefix"])
    shape = tuple(int(v) for v in g[tag + "__xshape"])
    seed = int(g[tag + "__seed"])
    parts = tag.split("_")
    if parts[0] == "nb":
        c, d, p = int(parts[1]), int(parts[2]), int(parts[3]) / 100.0
        tmpl, fwd = O.nb_template(prefix, c), (lambda sd, xx, st: O.nb1d(sd, prefix, xx, d, p, st))
    elif parts[0] == "down":
        tmpl, fwd = O.down_template(prefix, int(parts[1]), int(parts[2])), (lambda sd, xx, st: O.downsampler(sd, prefix, xx, st))
    else:
        tmpl, fwd = O.up_template(prefix, int(parts[1]), int(parts[2])), (lambda sd, xx, st: O.upsampler(sd, prefix, xx, st))
    x = S.synthetic_images(shape[0], shape[2], shape[3], seed=seed) if shape[1] == 3 else activations(shape, seed)
    mask = torch.from_numpy(g[tag + "_mask"]) if (tag + "_mask") in g.files else None
    return O.synthetic_state(tmpl), prefix, fwd, x, seed, mask


def run(sd, prefix, fwd, x, seed, mask):
    leaf = {k: (v.clone().requires_grad_(True) if (v.dtype.is_floating_point and "running" not in k) else v) for k, v in sd.items()}
    st = O.RunState(training=True, dropout_masks={prefix: mask} if mask is not None else {})
    xr = x.clone().requires_grad_(True)
    y = fwd(leaf, xr, st)
    y.backward(upstream(tuple(y.shape), seed))
    return y.detach(), xr.grad, {k: v.grad for k, v in leaf.items() if v.requires_grad and v.grad is not None}


def maxnorm(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    g = load("g9_gate_safe_blocks.npz")
    tags = sorted({k.split("__prefix")[0] for k in g.files if k.endswith("__prefix")})
    worst = {1: 0.0, 2: 0.0, 3: 0.0}
    print(f"{'fixture':28s} " + "  ".join(f"{n}-piece: y / dx / worst dW" for n in (1, 2, 3)))
    for tag in tags:
        c = case(tag, g)
        y0, dx0, gr0 = run(*c)
        row = []
        for n in (1, 2, 3):
            O.F = SplitF(n)
            try:
                y, dx, gr = run(*c)
            finally:
                O.F = F
            # weights and biases only: the gradient of a bias in front of a batch-statistics BatchNorm is rounding noise around zero
            ew = max(maxnorm(gr[k], gr0[k]) for k in gr0 if k.endswith("weight"))
            e = (maxnorm(y, y0), maxnorm(dx, dx0), ew)
            worst[n] = max(worst[n], *e)
            row.append(" / ".join(f"{v:.1e}" for v in e))
        print(f"{tag:28s} " + "   ".join(row), flush=True)
    print("worst over all fixtures: " + "  ".join(f"{n} piece(s) ({len(pairs(n))} products): {worst[n]:.1e}" for n in (1, 2, 3)))


if __name__ == "__main__":
    main()
