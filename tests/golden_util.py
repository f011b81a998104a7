"""Helpers shared by the parity tests: load tests/golden/*.npz and compare tensors
against entries stored by tests/golden/make_golden.py's `pack` (strided sample + sums)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def rel_err(a: torch.Tensor, b: torch.Tensor, floor: float = 0.0) -> float:
    """max |a-b| / max(max |b|, floor)  -- the 'relative error' used by every fp parity check.
    `floor` gives quantities that are mathematically zero (a conv bias gradient in front of a
    training-mode BatchNorm is pure cancellation noise) an absolute scale to be judged on."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = max(b.abs().max().item(), floor)
    return (a - b).abs().max().item() / (den if den > 0 else 1.0)


def gate_tolerant_err(a: torch.Tensor, b: torch.Tensor, outlier_frac: float, floor: float = 0.0):
    """(error quantile, relative L2 error) for GRADIENT comparisons.

    A gradient that passed through ReLUs is discontinuous in the forward values: when a
    pre-activation lies within rounding distance of zero, two correct implementations that sum in
    different orders (CPU oneDNN vs MFMA) open/close that gate differently and a handful of
    gradient elements differ by O(1) (measured: one flipped gate in 4.9e5 fp32 pre-activations moved
    2.7e-2 of a bias gradient).  So gradients are judged on the (1 - outlier_frac) quantile of the
    element errors, relative to the reference's max magnitude, plus the relative L2 error, which a
    few flipped gates barely move but any systematic fault does."""
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    den = max(b.abs().max().item(), floor)
    den = den if den > 0 else 1.0
    e = (a - b).abs() / den
    k = max(int(round((1.0 - outlier_frac) * e.numel())), 1)
    q = torch.kthvalue(e, min(k, e.numel())).values.item()
    l2 = (a - b).norm().item() / max(b.norm().item(), floor, 1e-30)
    return q, l2


def check_packed(g, name, t: torch.Tensor, tol: float, what: str = "", floor: float = 0.0, outlier_frac: float = 0.0,
                 l2tol: float = None):
    """Compare tensor t with the packed golden entry `name`: sample within tol relative to the
    sample's max magnitude (for gradients: all but `outlier_frac` of the elements, plus a relative L2
    bound); global sum within tol * abssum."""
    shape = tuple(int(v) for v in g[name + "__shape"])
    assert tuple(t.shape) == shape, f"{what}{name}: shape {tuple(t.shape)} != golden {shape}"
    stride = int(g[name + "__stride"])
    ref = torch.from_numpy(g[name + "__sample"])
    flat = t.detach().cpu().contiguous().reshape(-1)
    got = flat[::stride]
    if ref.dtype.is_floating_point:
        if outlier_frac > 0:
            err, l2 = gate_tolerant_err(got.float(), ref.float(), outlier_frac, floor)
            assert err <= tol, f"{what}{name}: {1 - outlier_frac:.3f}-quantile rel err {err:.3e} > {tol:.1e}"
            lim = l2tol if l2tol is not None else 100 * tol
            assert l2 <= lim, f"{what}{name}: relative L2 err {l2:.3e} > {lim:.1e}"
            return err
        err = rel_err(got.float(), ref.float(), floor)
        assert err <= tol, f"{what}{name}: sample rel err {err:.3e} > {tol:.1e}"
        s = flat.double().sum().item()
        gs, ga = float(g[name + "__sum"]), float(g[name + "__abssum"])
        assert abs(s - gs) <= tol * max(ga, floor, 1e-30) * 4, f"{what}{name}: sum {s} vs golden {gs} (abssum {ga})"
        return err
    assert torch.equal(got, ref), f"{what}{name}: integer sample mismatch"
    return 0.0


def unpack_full(g, name) -> torch.Tensor:
    assert int(g[name + "__stride"]) == 1, name + " was stored as a sample"
    shape = tuple(int(v) for v in g[name + "__shape"])
    return torch.from_numpy(g[name + "__sample"]).reshape(shape)


def upstream(shape, seed):
    from erfnet_pytorch_amd import synthetic as S
    n = int(np.prod(shape))
    return torch.from_numpy(S.unit_uniform(n, S.key_seed("upstream", seed)).astype(np.float32).reshape(shape))


def activations(shape, seed):
    from erfnet_pytorch_amd import synthetic as S
    n = int(np.prod(shape))
    u = S.unit_uniform(n, S.key_seed("act", seed))
    return torch.from_numpy(np.maximum(u * 1.5 - 0.2, 0).astype(np.float32).reshape(shape))


def pre_bn_bias(key: str) -> bool:
    """True for a conv bias that feeds a training-mode BatchNorm directly: its gradient is
    mathematically zero and what both sides hold is fp32 cancellation noise."""
    return key.endswith(("conv1x3_1.bias", "conv1x3_2.bias", ".conv.bias")) or key in ("conv.bias",)
