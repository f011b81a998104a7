"""The operators as torch.library custom ops (namespace erfnet_mi355): every op is registered with a device kernel and a
fake kernel; the fake kernels give the shapes / dtypes the device kernels produce (checked without a GPU on meta tensors and
under FakeTensorMode, and against the real kernels with torch.library.opcheck on the GPU); CPU tensors are refused."""
import pytest
import torch

from erfnet_pytorch_amd import torch_ops as TO
from erfnet_pytorch_amd._lib import BF16, F32

T = torch.ops.erfnet_mi355


def _samples(dev, dtype=torch.bfloat16):
    """op name -> (args, expected output (shape, dtype) list)"""
    def t(*shape, dt=dtype):
        return torch.empty(shape, dtype=dt, device=dev)
    x64 = t(2, 8, 16, 64)
    pk = t(4096, dt=torch.uint8)
    f = lambda n: t(n, dt=torch.float32)
    logits = t(2, 20, 16, 32, dt=torch.float32)
    target = t(2, 16, 32, dt=torch.int64)
    return {
        "conv1d": ((x64, pk, f(64), f(64), x64, 0, 2, True), [((2, 8, 16, 64), dtype)]),
        "conv1d_pair": ((x64, pk, f(64), pk, f(64), f(64), x64, 2, True), [((2, 8, 16, 64), dtype)]),
        "nb1d_block": ((x64, pk, pk, pk, pk, t(8, 64, dt=torch.float32), 1), [((2, 8, 16, 64), dtype)]),
        "conv1d_dgrad": ((x64, pk, x64, None, 1, 4), [((2, 8, 16, 64), dtype)]),
        "conv1d_wgrad": ((x64, x64, 1, 1), [((64, 64, 1, 3), torch.float32)]),
        "downsampler": ((t(2, 8, 16, 16), pk, f(64), f(64), 64, True), [((2, 4, 8, 64), dtype)]),
        "stem": ((t(2, 3, 8, 16, dt=torch.float32), f(351), f(16), f(16), BF16, True), [((2, 4, 8, 16), torch.bfloat16)]),
        "upsampler": ((x64, pk, f(16), f(16), 16, True), [((2, 16, 32, 16), dtype)]),
        "output_conv": ((t(2, 8, 16, 16), pk, f(20), 20, True, True, 1), [((2, 20, 16, 32), torch.float32), ((2, 16, 32), torch.uint8)]),
        "pointwise": ((t(2, 8, 16, 128), pk, f(20), 20, True), [((2, 20, 8, 16), torch.float32)]),
        "conv3x3": ((x64, pk, f(64), f(64), 64, 2, True), [((2, 4, 8, 64), dtype)]),
        "maxpool2x2_affine": ((x64, f(64), f(64), True), [((2, 4, 8, 64), dtype)]),
        "concat_channels": ((x64, t(2, 8, 16, 16)), [((2, 8, 16, 80), dtype)]),
        "convT2x2s2": ((t(2, 8, 16, 32), pk, f(1), 1, True), [((2, 1, 16, 32), torch.float32)]),
        "spatial_mean": ((x64,), [((2, 64), torch.float32)]),
        "linear": ((t(2, 64, dt=torch.float32), t(14, 64, dt=torch.float32), f(14)), [((2, 14), torch.float32)]),
        "image_to_nhwc_channels": ((t(2, 3, 8, 16, dt=torch.float32), BF16, 16), [((2, 8, 16, 16), torch.bfloat16)]),
        "bn_apply": ((x64, f(64), f(64), f(64), x64, t(2, 64, dt=torch.float32), True), [((2, 8, 16, 64), dtype)]),
        "channel_sum": ((x64,), [((64,), torch.float32)]),
        "dropout2d_mask": ((x64, 2, 64, 0.3, -5), [((2, 64), torch.float32)]),
        "weighted_ce_fwd": ((logits, target, f(20)), [((3,), torch.float32)]),
        "weighted_ce_bwd": ((logits, target, None, f(1)), [((2, 20, 16, 32), torch.float32)]),
        "argmax_c": ((logits, 8), [((2, 16, 32), torch.int64)]),
        "iou_hist_accumulate": ((t(2, 16, 32, dt=torch.uint8), target, 20, t(401, dt=torch.int64)), []),
        "labels_from_u8": ((t(2, 16, 32, dt=torch.uint8), 255, 19), [((2, 16, 32), torch.int64)]),
        "adam_fused_arena_": ((f(100), f(100), f(100), f(100), 5e-4, 0.9, 0.999, 1e-8, 1e-4, 3, 1.0), []),
    }


def _outs(res):
    if res is None:
        return []
    return list(res) if isinstance(res, (tuple, list)) else [res]


def test_every_op_is_registered_with_a_schema_a_device_kernel_and_a_fake_kernel():
    assert len(TO.OPS) >= 18 and set(TO.OPS) == set(_samples("meta"))
    for name in TO.OPS:
        op = getattr(T, name).default
        assert op._schema.name == f"erfnet_mi355::{name}"
        assert torch._C._dispatch_has_kernel_for_dispatch_key(op.name(), "CUDA"), name
        assert torch._C._dispatch_has_kernel_for_dispatch_key(op.name(), "Meta"), name
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(op.name(), "CPU"), name


@pytest.mark.parametrize("name", sorted(_samples("meta")))
def test_fake_kernels_give_shapes_on_meta_tensors(name):
    args, want = _samples("meta")[name]
    got = _outs(getattr(T, name)(*args))
    assert [(tuple(o.shape), o.dtype) for o in got] == want
    assert all(o.device.type == "meta" for o in got)


def test_ops_trace_under_fake_tensor_mode():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty(4, 64, 128, 64, dtype=torch.bfloat16, device="cuda")
        pk = torch.empty(1 << 14, dtype=torch.uint8, device="cuda")
        v = torch.empty(64, device="cuda")
        y = T.conv1d_pair(x, pk, v, pk, v, v, None, 1, True)
        z = T.conv1d_pair(y, pk, v, pk, v, v, x, 4, True)
        up = T.upsampler(z, pk, torch.empty(16, device="cuda"), torch.empty(16, device="cuda"), 16, True)
        logits, labels = T.output_conv(up, pk, torch.empty(20, device="cuda"), 20, True, False, 8)
    assert tuple(z.shape) == (4, 64, 128, 64) and z.dtype == torch.bfloat16 and z.device.type == "cuda"
    assert tuple(logits.shape) == (4, 20, 256, 512) and logits.dtype == torch.bfloat16
    assert tuple(labels.shape) == (4, 256, 512) and labels.dtype == torch.int64


def test_cpu_tensors_are_refused():
    x = torch.zeros(1, 4, 4, 64)
    with pytest.raises(NotImplementedError):
        T.conv1d(x, torch.zeros(16, dtype=torch.uint8), None, None, None, 0, 1, True)


@pytest.mark.gpu
def test_opcheck_against_the_device_kernels():
    """torch.library.opcheck: the schema's aliasing / mutation annotations and the fake kernels agree with what the device
    kernels do, on real packed weights."""
    from erfnet_pytorch_amd import erfnet as E, ops
    from oracle import erfnet_oracle as O
    dev = "cuda:0"
    blk = E.non_bottleneck_1d(64, 0.03, 2)
    sd = O.synthetic_state(O.nb_template("encoder.layers.1", 64))
    blk.load_state_dict({k[len("encoder.layers.1."):]: v for k, v in sd.items()})
    blk = blk.to(dev).eval()
    x = torch.relu(torch.randn(2, 16, 32, 64, device=dev)).bfloat16()
    c = blk._cached(ops.dt(x))
    checks = ("test_schema", "test_faketensor")
    torch.library.opcheck(T.conv1d.default, (x, c["w1"], None, c["t1"], None, 0, 1, True), test_utils=checks)
    torch.library.opcheck(T.conv1d_pair.default, (x, c["w3"], c["t3"], c["w4"], c["s4"], c["t4"], x, 2, True), test_utils=checks)
    torch.library.opcheck(T.bn_apply.default, (x, None, c["s2"], c["t2"], x, None, True), test_utils=checks)
    torch.library.opcheck(T.channel_sum.default, (x,), test_utils=checks)
    logits = torch.randn(2, 20, 16, 32, device=dev)
    target = torch.randint(0, 20, (2, 16, 32), device=dev)
    torch.library.opcheck(T.weighted_ce_fwd.default, (logits, target, None), test_utils=checks)
    torch.library.opcheck(T.argmax_c.default, (logits, 1), test_utils=checks)
    hist = torch.zeros(401, dtype=torch.int64, device=dev)
    torch.library.opcheck(T.iou_hist_accumulate.default, (T.argmax_c(logits, 8), target, 20, hist), test_utils=checks)
    p, g, m, v = (torch.randn(1000, device=dev) for _ in range(4))
    torch.library.opcheck(T.adam_fused_arena_.default, (p, g, m, v.abs(), 5e-4, 0.9, 0.999, 1e-8, 1e-4, 1, 1.0), test_utils=checks)
