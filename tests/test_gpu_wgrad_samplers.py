"""Weight gradients of the samplers / output layer / 1x1 head through the C ABI (erf_downsampler_wgrad, erf_upsampler_wgrad,
erf_output_conv_wgrad, erf_pointwise_wgrad -- the one-tap-per-workgroup kernel) against torch CPU autograd of the reference's
layers (train/erfnet.py:15 Conv2d(3x3, stride 2, padding 1), :149 ConvTranspose2d(3x3, stride 2, padding 1, output_padding 1),
:171 ConvTranspose2d(2x2, stride 2), :96 Conv2d(1x1)).  The operation is linear in both operands, so the bars are rounding
only: fp32 1e-5 of the largest gradient; bf16 operands are rounded first and the same 1e-5 applies (fp32 accumulation).
Shapes include pixel counts that are not a multiple of the kernel's 128-pixel step and ones smaller than a single step."""
import pytest
import torch
import torch.nn.functional as F

from tests.golden_util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from erfnet_pytorch_amd import ops
    from erfnet_pytorch_amd._lib import call, size_query
    return ops, call, size_query


def _rand(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _operands(xshape, dzshape, dtype, seed):
    x = _rand(xshape, seed)
    dz = _rand(dzshape, seed + 1)
    if dtype == torch.bfloat16:
        x, dz = x.bfloat16().float(), dz.bfloat16().float()
    return x, dz


def _nhwc(t, dtype):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)


def _run(fn, x, dz, wshape, rows, cin, taps, pixels, dtype, *dims, phases=1):
    ops, call, size_query = _ops()
    xd, dzd = _nhwc(x, dtype), _nhwc(dz, dtype)
    dw = torch.full(wshape, float("nan"), device=DEV)
    nb = phases * size_query("erf_wgrad_workspace_bytes", rows, cin, taps, pixels)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    call(fn, xd.data_ptr(), dzd.data_ptr(), dw.data_ptr(), ops.dt(xd), *dims, ws.data_ptr(), nb, ops._stream())
    torch.cuda.synchronize()
    return dw.cpu()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,n,h,w", [(16, 64, 2, 20, 28), (64, 128, 1, 10, 14), (64, 128, 3, 32, 48), (16, 64, 1, 6, 10)])
def test_downsampler_weight_gradient(cin, cout, n, h, w, dtype):
    cc = cout - cin
    x, dz = _operands((n, cin, h, w), (n, cout, h // 2, w // 2), dtype, 7 + cin + h)
    wt = torch.zeros(cc, cin, 3, 3, requires_grad=True)
    (F.conv2d(x, wt, None, stride=2, padding=1) * dz[:, :cc]).sum().backward()
    dw = _run("erf_downsampler_wgrad", x, dz, (cc, cin, 3, 3), cc, cin, 9, n * (h // 2) * (w // 2), dtype, n, h, w, cin, cout)
    assert rel_err(dw, wt.grad) <= 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,n,h,w", [(128, 64, 1, 9, 13), (64, 16, 2, 12, 20), (128, 64, 2, 16, 32), (64, 16, 1, 5, 7)])
def test_upsampler_weight_gradient(cin, cout, n, h, w, dtype):
    x, dz = _operands((n, cin, h, w), (n, cout, 2 * h, 2 * w), dtype, 11 + cin + h)
    wt = torch.zeros(cin, cout, 3, 3, requires_grad=True)
    (F.conv_transpose2d(x, wt, None, stride=2, padding=1, output_padding=1) * dz).sum().backward()
    dw = _run("erf_upsampler_wgrad", x, dz, (cin, cout, 3, 3), cout, cin, 4, n * h * w, dtype, n, h, w, cin, cout, phases=4)
    assert rel_err(dw, wt.grad) <= 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nc,n,h,w", [(20, 1, 14, 22), (6, 2, 16, 24), (20, 1, 4, 6)])
def test_output_conv_weight_gradient(nc, n, h, w, dtype):
    """Decoder.output_conv: the gradient of the logits arrives as the patch tensor [N, H, W, ld], channel 4 * class + phase."""
    ops, call, size_query = _ops()
    x, _ = _operands((n, 16, h, w), (1, 1, 1, 1), dtype, 13 + nc)
    dlog = _rand((n, nc, 2 * h, 2 * w), 17 + nc)
    if dtype == torch.bfloat16:
        dlog = dlog.bfloat16().float()
    wt = torch.zeros(16, nc, 2, 2, requires_grad=True)
    (F.conv_transpose2d(x, wt, None, stride=2) * dlog).sum().backward()
    ld = size_query("erf_output_conv_patch_channels", nc)
    patches = torch.zeros(n, h, w, ld)
    for a in range(2):
        for b in range(2):
            patches[..., [4 * k + 2 * a + b for k in range(nc)]] = dlog[:, :, a::2, b::2].permute(0, 2, 3, 1)
    xd, pd = _nhwc(x, dtype), patches.to(DEV, dtype)
    dw = torch.full((16, nc, 2, 2), float("nan"), device=DEV)
    nb = size_query("erf_wgrad_workspace_bytes", 4 * nc, 16, 1, n * h * w)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    call("erf_output_conv_wgrad", xd.data_ptr(), pd.data_ptr(), dw.data_ptr(), ops.dt(xd), n, h, w, nc, ws.data_ptr(), nb, ops._stream())
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), wt.grad) <= 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nc,n,h,w", [(20, 2, 9, 15), (6, 1, 16, 32)])
def test_pointwise_weight_gradient(nc, n, h, w, dtype):
    """Encoder.output_conv (1x1, 128 -> nc): the logits gradient arrives NHWC with the classes padded to a multiple of 16."""
    ops, call, size_query = _ops()
    x, _ = _operands((n, 128, h, w), (1, 1, 1, 1), dtype, 19 + nc)
    dlog = _rand((n, nc, h, w), 23 + nc)
    if dtype == torch.bfloat16:
        dlog = dlog.bfloat16().float()
    wt = torch.zeros(nc, 128, 1, 1, requires_grad=True)
    (F.conv2d(x, wt) * dlog).sum().backward()
    ld = size_query("erf_pointwise_grad_channels", nc)
    g = torch.zeros(n, h, w, ld)
    g[..., :nc] = dlog.permute(0, 2, 3, 1)
    xd, gd = _nhwc(x, dtype), g.to(DEV, dtype)
    dw = torch.full((nc, 128, 1, 1), float("nan"), device=DEV)
    nb = size_query("erf_wgrad_workspace_bytes", nc, 128, 1, n * h * w)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    call("erf_pointwise_wgrad", xd.data_ptr(), gd.data_ptr(), dw.data_ptr(), ops.dt(xd), n, h, w, 128, nc, ws.data_ptr(), nb, ops._stream())
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), wt.grad) <= 1e-5
