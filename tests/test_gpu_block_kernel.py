"""erf_nb1d_block_infer (a whole non_bottleneck_1d block in one launch, bf16, 64 or 16 channels, dilation 1) against the path the
modules use by default (two conv1d_pair launches): the same fragments meet the same matrix instructions in the same order, so
the outputs must be BIT-IDENTICAL, including on tiles that touch the image border, ragged sizes and images smaller than a
tile.  The golden block output of the reference (train/erfnet.py:58-74, eval mode) is checked through the same entry point."""
import pytest
import torch

from erfnet_pytorch_amd import erfnet as E, ops
from oracle import erfnet_oracle as O
from tests.golden_util import load, check_packed, activations

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _block(c=64, d=1):
    blk = E.non_bottleneck_1d(c, 0.03, d)
    sd = O.synthetic_state(O.nb_template("encoder.layers.1", c))
    blk.load_state_dict({k[len("encoder.layers.1."):]: v for k, v in sd.items()})
    return blk.to(DEV).eval()


def _run(blk, x, fused):
    old = E.FUSE_BLOCKS
    E.FUSE_BLOCKS = (16, 64) if fused else ()
    try:
        with torch.no_grad():
            return blk.forward_nhwc(x)
    finally:
        E.FUSE_BLOCKS = old


@pytest.mark.parametrize("shape", [(1, 8, 32), (1, 16, 64), (2, 24, 40), (1, 13, 37), (3, 40, 100), (2, 7, 5), (1, 2, 3), (1, 1, 1),
                                   (1, 128, 256), (5, 64, 96), (1, 9, 130), (2, 70, 66)])
@pytest.mark.parametrize("chann", [64, 16])
def test_block_kernel_is_bit_identical_to_the_pair_path(shape, chann):
    n, h, w = shape
    blk = _block(chann)
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + w)
    x = torch.relu(torch.randn(n, h, w, chann, generator=g)).to(DEV, torch.bfloat16)
    a = _run(blk, x, True)
    b = _run(blk, x, False)
    a2 = _run(blk, x, True)                        # and it is deterministic
    c = blk._cached(ops.dt(x))
    pair_taken = ops.conv1d_pair(x, c["w1"], c["t1"], c["w2"], c["s2"], c["t2"], None, 1) is not None
    torch.cuda.synchronize()
    assert torch.equal(a, a2)
    if pair_taken:
        assert torch.equal(a, b), float((a.float() - b.float()).abs().max())
    else:
        # images that fill less than 60 % of a pair tile go through the single-convolution kernels, whose summation order
        # differs: one bf16 ulp at most
        assert float(((a.float() - b.float()).abs() / b.float().abs().clamp_min(1.0)).max()) <= 2.0 ** -7


def test_block_kernel_declines_what_it_does_not_cover():
    x = torch.zeros(1, 8, 32, 64, device=DEV, dtype=torch.bfloat16)
    blk = _block()
    c = blk._cached(ops.dt(x))
    w4 = (c["w1"], c["w2"], c["w3"], c["w4"])
    assert ops.nb1d_block(x, w4, c["table"], 2) is None                       # dilation
    assert ops.nb1d_block(x.float(), w4, c["table"], 1) is None               # fp32 storage
    x128 = torch.zeros(1, 8, 32, 128, device=DEV, dtype=torch.bfloat16)
    assert ops.nb1d_block(x128, w4, c["table"], 1) is None                    # channel count
    with pytest.raises(RuntimeError):
        ops.nb1d_block(x, w4, c["table"], 1, out=x)                           # in place is refused
    x16 = torch.zeros(1, 8, 32, 16, device=DEV, dtype=torch.bfloat16)
    blk16 = _block(16)
    c16 = blk16._cached(ops.dt(x16))
    w16 = (c16["w1"], c16["w2"], c16["w3"], c16["w4"])
    assert ops.nb1d_block(x16, w16, c16["table"], 1) is not None
    assert ops.nb1d_block(x16, w16, c16["table"], 2) is None
    assert ops.nb1d_block(x16.float(), w16, c16["table"], 1) is None


def test_registered_block_op_falls_back_to_the_pair_launches():
    """torch.ops.erfnet_mi355.nb1d_block on a shape the one-launch kernels decline (dilation 2): the op issues the two
    conv1d_pair launches itself instead of failing, so a module that routes a block through it keeps working."""
    blk = _block(64, 2)
    x = torch.relu(torch.randn(2, 24, 40, 64, generator=torch.Generator().manual_seed(3))).to(DEV, torch.bfloat16)
    c = blk._cached(ops.dt(x))
    y = torch.ops.erfnet_mi355.nb1d_block(x, c["w1"], c["w2"], c["w3"], c["w4"], c["table"], 2)
    assert torch.equal(y, _run(blk, x, False))


def test_block_kernel_vs_reference_golden():
    """The reference's own eval-mode output of a 64-channel dilation-1 block (g2_blocks.npz, written by tests/golden/make_golden.py
    from train/erfnet.py's module), through the one-launch kernel; the bf16 bar is the one the default path is held to."""
    g = load("g2_blocks.npz")
    tag = "nb_c64_d1_p3"
    prefix = str(g[tag + "__prefix"])
    shape = tuple(int(v) for v in g[tag + "__xshape"])
    mod = E.non_bottleneck_1d(64, 0.03, 1)
    sd = O.synthetic_state(O.nb_template(prefix, 64))
    mod.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()})
    mod = mod.to(DEV).eval()
    x = ops.to_nhwc(activations(shape, int(g[tag + "__seed"])).to(DEV, torch.bfloat16))
    y = ops.to_nchw_view(_run(mod, x, True))
    assert y.shape == tuple(int(v) for v in g[tag + "_y_eval__shape"])
    check_packed(g, tag + "_y_eval", y.float().cpu(), 2e-2)
