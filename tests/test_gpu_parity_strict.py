"""Element-wise parity of the training kernels on inputs without fragile decisions.

A gradient that passed through ReLUs / max-pools is discontinuous in the forward values: when a pre-activation lies within
rounding distance of zero, two correct implementations that sum in different orders open that gate differently and the
gradients differ by O(1) in its neighbourhood.  The fixtures of g9_gate_safe_blocks.npz were searched (by
tests/golden/make_golden.py, running the reference's modules) so that NO pre-activation of the reference lies within 2e-5 of a
decision boundary (relative to its tensor's rms) -- neither in the reference's fp32 arithmetic nor in the same graph with
bf16 storage.  Every fp32 decision is therefore the same on both sides.

  fp32  max-norm over ALL elements, no outlier allowance, against the REFERENCE's stored outputs: 1e-5 of each tensor's max
        magnitude for the block output, the input gradient, every parameter gradient and the running statistics (measured
        1e-7 .. 1.7e-6; the north-star bar is 1e-4).  Gradients of a conv bias in front of a batch-statistics BatchNorm are
        mathematically zero (the engine writes exact zeros, the reference rounding noise): judged on an absolute scale.
  bf16  against the oracle evaluated with bf16 storage (RunState(storage_dtype=bfloat16): same graph, every stored feature
        map / weight / feature-map gradient rounded where the engine rounds it, fp32 accumulation).  Most fixtures agree to
        <= 4.2e-3 everywhere (block outputs often bit for bit).  A second-order effect remains that no fixture can exclude:
        summation order moves a value across a bf16 rounding boundary in a few elements per thousand (1 ulp = 0.4 %), and
        when such an element feeds a pre-activation that is itself within 0.4 % of zero, that gate opens differently; one
        such gate changes the gradients in its receptive field.  Stated bf16 bound: error <= 1.5e-2 of the tensor's max
        magnitude on at least 99 % of the elements AND relative L2 error <= 2e-2, for every tensor (measured worst case with
        one flipped gate: 0.7 % of the elements, L2 1.3e-2; the earlier bar was 0.15 on 90 % and L2 0.3).  Per-channel vectors
        (fewer than 4096 elements; a channel's sum over the 256-384 pixels of these fixtures moves by 4e-2 when one of its
        gates flips): every element within 5e-2, L2 <= 2e-2.
"""
import numpy as np
import pytest
import torch

from erfnet_pytorch_amd import synthetic as S
from oracle import erfnet_oracle as O
from tests.golden_util import load, unpack_full, activations, upstream, rel_err, pre_bn_bias

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FP32_TOL = 1e-5
BF16_ACT_TOL, BF16_PARAM_TOL, BF16_OVER, BF16_L2, FLOOR_BF16 = 1.5e-2, 1.5e-2, 0.01, 2e-2, 100.0


def _tags():
    g = load("g9_gate_safe_blocks.npz")
    return sorted({k.split("__prefix")[0] for k in g.files if k.endswith("__prefix")})


def _case(tag, g):
    from erfnet_pytorch_amd import erfnet as E
    prefix = str(g[tag + "__prefix"])
    shape = tuple(int(v) for v in g[tag + "__xshape"])
    seed = int(g[tag + "__seed"])
    parts = tag.split("_")
    if parts[0] == "nb":
        c, d, p = int(parts[1]), int(parts[2]), int(parts[3]) / 100.0
        mod, tmpl = E.non_bottleneck_1d(c, p, d), O.nb_template(prefix, c)
        fwd = lambda sd, xx, st: O.nb1d(sd, prefix, xx, d, p, st)
    elif parts[0] == "down":
        cin, cout = int(parts[1]), int(parts[2])
        mod, tmpl = E.DownsamplerBlock(cin, cout), O.down_template(prefix, cin, cout)
        fwd = lambda sd, xx, st: O.downsampler(sd, prefix, xx, st)
    else:
        cin, cout = int(parts[1]), int(parts[2])
        mod, tmpl = E.UpsamplerBlock(cin, cout), O.up_template(prefix, cin, cout)
        fwd = lambda sd, xx, st: O.upsampler(sd, prefix, xx, st)
    x = S.synthetic_images(shape[0], shape[2], shape[3], seed=seed) if shape[1] == 3 else activations(shape, seed)
    sd = O.synthetic_state(tmpl)
    mod.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()})
    mask = torch.from_numpy(g[tag + "_mask"]) if (tag + "_mask") in g.files else None
    return mod.to(DEV).train(), sd, prefix, fwd, x, seed, mask


def _oracle(sd, prefix, fwd, x, seed, mask, storage_dtype):
    leaf = {k: (v.clone().requires_grad_(True) if (v.dtype.is_floating_point and "running" not in k) else v) for k, v in sd.items()}
    st = O.RunState(training=True, dropout_masks={prefix: mask} if mask is not None else {}, storage_dtype=storage_dtype,
                    record_gates=True)
    xr = x.clone().requires_grad_(True)
    y = fwd(leaf, xr, st)
    y.backward(upstream(tuple(y.shape), seed))
    grads = {k[len(prefix) + 1:]: v.grad for k, v in leaf.items() if v.requires_grad}
    bufs = {k[len(prefix) + 1:]: v for k, v in st.new_running.items()}
    return y.detach(), xr.grad, grads, bufs, min(m for _, m in st.gate_margins)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("tag", _tags())
def test_gate_safe_block_elementwise(tag, dtype):
    g = load("g9_gate_safe_blocks.npz")
    mod, sd, prefix, fwd, x, seed, mask = _case(tag, g)
    bf16 = dtype == torch.bfloat16
    is_stem = tag.startswith("down_3_")
    if mask is not None:
        mod.__dict__["_erf_forced_mask"] = mask
    if bf16 and not is_stem:
        x = x.bfloat16().float()
    if is_stem and bf16:
        mod.__dict__["_erf_code"] = 1
    xg = x.to(DEV, torch.float32 if is_stem else dtype).requires_grad_(not is_stem)
    y = mod(xg)
    y.backward(upstream(tuple(y.shape), seed).to(DEV, y.dtype))
    torch.cuda.synchronize()
    margin_fp32, margin_bf16 = (float(v) for v in g[tag + "__margins"])
    need = float(g["margin_required"])
    assert margin_fp32 >= need and margin_bf16 >= need
    if bf16:
        ry, rdx, rgrads, rbufs, margin = _oracle(sd, prefix, fwd, x, seed, mask, torch.bfloat16)
        assert margin >= 0.5 * need, margin           # the decisions of this evaluation are as clear as the generator found them
        atol, ptol = BF16_ACT_TOL, BF16_PARAM_TOL
    else:
        ry, rdx = unpack_full(g, tag + "_y_train"), (None if is_stem else unpack_full(g, tag + "_dx"))
        rgrads = {k: unpack_full(g, tag + "_grad_" + k) for k, _ in mod.named_parameters()}
        rbufs = {k: unpack_full(g, tag + "_buf_" + k) for k, _ in mod.named_buffers() if "num_batches" not in k}
        atol = ptol = FP32_TOL
    def check(name, a, b, tol, floor=0.0):
        a, b = a.double().reshape(-1).cpu(), b.double().reshape(-1).cpu()
        den = max(float(b.abs().max()), floor)
        e = (a - b).abs() / den
        worst = float(e.max())
        if not bf16:
            assert worst <= tol, (tag, name, worst)
            return worst
        over = float((e > tol).sum()) / e.numel()
        l2 = float((a - b).norm()) / max(float(b.norm()), floor)
        if e.numel() < 4096:               # per-channel vectors: a share of the elements means nothing, bound the worst one
            assert worst <= 5e-2 and l2 <= BF16_L2, (tag, name, worst, over, l2)
        else:
            assert over <= BF16_OVER and l2 <= BF16_L2, (tag, name, worst, over, l2)
        return worst

    errs = {"y": check("y", y.float(), ry, atol)}
    if not is_stem:
        errs["dx"] = check("dx", xg.grad.float(), rdx, atol)
    for k, p in mod.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32 and p.grad.shape == p.shape, k
        errs[k] = check(k, p.grad, rgrads[k], ptol, floor=FLOOR_BF16 if (bf16 and pre_bn_bias(k)) else (30.0 if pre_bn_bias(k) else 0.0))
    for k, b in mod.named_buffers():
        if "num_batches" not in k:
            assert rel_err(b.cpu(), rbufs[k]) <= (2e-3 if bf16 else 1e-5), k
    print(tag, dtype, {k: f"{v:.2e}" for k, v in errs.items()})
