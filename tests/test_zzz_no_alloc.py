"""The entry point that allocates nothing (erf_bn_stats_finalize_static) against the workspace-taking entry point it is
a variant of (erf_bn_stats_finalize), through the C ABI: bit-identical outputs on two calls in a row, the second on a
smaller input (the library's fixed array keeps the first call's partial rows), and the size limit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(z, c, seed):
    from erfnet_pytorch_amd import _lib, ops
    dev = z.device
    g = torch.Generator(device="cpu").manual_seed(seed)
    gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
    beta = torch.randn(c, generator=g).to(dev)
    rm0, rv0 = torch.randn(c, generator=g).to(dev), (torch.rand(c, generator=g) + 0.5).to(dev)
    pixels = z.numel() // c
    st = ops._stream()
    outs = []
    for static in (False, True):
        rm, rv = rm0.clone(), rv0.clone()
        mean, invstd, scale, shift = (torch.full((c,), float("nan"), device=dev) for _ in range(4))
        args = [z.data_ptr(), ops.dt(z), pixels, c, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-3,
                mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(), shift.data_ptr()]
        if static:
            _lib.call("erf_bn_stats_finalize_static", *args, st)
        else:
            ws = torch.empty(_lib.query("erf_bn_workspace_bytes", c), dtype=torch.uint8, device=dev)
            _lib.call("erf_bn_stats_finalize", *args, ws.data_ptr(), st)
        torch.cuda.synchronize()
        outs.append([t.cpu() for t in (mean, invstd, scale, shift, rm, rv)])
    return outs


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_static_scratch_entry_point_matches_workspace_entry_point(dtype):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    # first call: large input, every partial row of the fixed array in use (6 x 64 x 128 pixels x 128 channels);
    # second call right after it: smaller input and fewer channels, so stale rows of the first call lie beyond its own
    for (n, h, w, c), seed in (((6, 64, 128, 128), 1), ((1, 24, 40, 64), 2), ((2, 16, 16, 16), 3)):
        z = (torch.randn(n, h, w, c, generator=g) * 1.7 + 0.3).to(dev, dtype)
        ref, got = _both(z, c, seed)
        for a, b in zip(ref, got):
            assert torch.isfinite(a).all()
            assert torch.equal(a, b)
        # and both agree with torch's own batch statistics
        zf = z.float().reshape(-1, c).cpu()
        assert torch.allclose(ref[0], zf.mean(0), atol=2e-5, rtol=1e-5)
        assert torch.allclose(ref[1], torch.rsqrt(zf.var(0, unbiased=False) + 1e-3), atol=1e-5, rtol=2e-5)


def test_static_scratch_entry_point_rejects_sizes_beyond_its_array():
    from erfnet_pytorch_amd import _lib, ops
    dev = torch.device("cuda:0")
    assert _lib.query("erf_static_scratch_max_channels") == 128
    c = 256
    z = torch.randn(2, 8, 8, c, device=dev)
    v = [torch.zeros(c, device=dev) for _ in range(8)]
    with pytest.raises(_lib.ErfError, match="fixed limit"):
        _lib.call("erf_bn_stats_finalize_static", z.data_ptr(), ops.dt(z), 128, c, v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(),
                  v[3].data_ptr(), 0.1, 1e-3, v[4].data_ptr(), v[5].data_ptr(), v[6].data_ptr(), v[7].data_ptr(), ops._stream())
    torch.cuda.synchronize()
    assert all(float(t.abs().sum()) == 0 for t in v[2:])          # nothing was launched


_CAPTURE_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
from erfnet_pytorch_amd import _lib, ops
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
c = 64
g = torch.Generator().manual_seed(3)
z = torch.randn(2, 16, 24, c, generator=g).to(dev)
gamma, beta = torch.rand(c, generator=g).to(dev) + 0.5, torch.randn(c, generator=g).to(dev)
rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
outs = [torch.zeros(c, device=dev) for _ in range(4)]
ws = torch.empty(_lib.query("erf_bn_workspace_bytes", c), dtype=torch.uint8, device=dev)
def args(rm_, rv_, o):
    return [z.data_ptr(), ops.dt(z), z.numel() // c, c, gamma.data_ptr(), beta.data_ptr(), rm_.data_ptr(), rv_.data_ptr(), 0.1, 1e-3] + [t.data_ptr() for t in o]
if {warm}:
    _lib.call("erf_bn_stats_finalize_static", *args(rm.clone(), rv.clone(), [t.clone() for t in outs]), ops._stream())
    torch.cuda.synchronize()
side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
    _lib.call("erf_bn_stats_finalize_static", *args(rm, rv, outs), ops._stream())
for rep in range(2):
    z.copy_(torch.randn(2, 16, 24, c, generator=g) * (rep + 1.5))
    rm.zero_(); rv.fill_(1.0)
    graph.replay()
    torch.cuda.synchronize()
    rm2, rv2 = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    ref = [torch.zeros(c, device=dev) for _ in range(4)]
    _lib.call("erf_bn_stats_finalize", *args(rm2, rv2, ref), ws.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(outs + [rm, rv], ref + [rm2, rv2])), rep
print("CAPTURE-OK")
"""


@pytest.mark.parametrize("warm", [True, False])
def test_static_scratch_entry_point_under_graph_capture(warm):
    """The advertised caller: the call captured into a HIP graph and replayed twice on different inputs, bit-identical to the
    workspace entry point -- once after a warm call outside the capture, once as the library's very first call on the device
    (a fresh process: the per-device symbol lookup then happens while the stream is capturing)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CAPTURE_SCRIPT.format(root=root, warm=warm)], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0 and "CAPTURE-OK" in r.stdout, r.stderr[-2000:]
