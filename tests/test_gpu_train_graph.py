"""Trainer(graph=True): the optimisation step of train/main.py:196-205 captured into one HIP graph and replayed.

The step's changing scalars (dropout seed counter, Adam's step count and learning rate) live in device memory whether the
step is replayed or launched kernel by kernel, so both ways must produce the SAME BITS: losses, parameters, Adam moments,
BatchNorm running statistics and batch counters after several steps on different batches, with a learning-rate change in
between (train/main.py:156-165 changes it every epoch)."""
import pytest
import torch

from erfnet_pytorch_amd import erfnet as E, synthetic as S
from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS, poly_lr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(bf16, graph, **kw):
    torch.manual_seed(77)
    model = E.ERFNet(20)
    S.load_synthetic(model)
    model = model.to(DEV).train()
    model.compute_dtype = torch.bfloat16 if bf16 else None
    return Trainer(model, CITYSCAPES_WEIGHTS, graph=graph, **kw)


def _batches(n, steps, h=64, w=128):
    g = torch.Generator(device=DEV).manual_seed(5)
    out = []
    for _ in range(steps):
        x = torch.randn(n, 3, h, w, device=DEV, generator=g)
        t = torch.randint(0, 20, (n, 1, h, w), device=DEV, generator=g)
        out.append((x, t))
    return out


@pytest.mark.parametrize("bf16", [True, False])
def test_replayed_step_equals_launched_step_bit_for_bit(bf16):
    steps = 5
    data = _batches(2, steps)
    runs = []
    for graph in (False, True):
        tr = _make(bf16, graph)
        losses = []
        for i, (x, t) in enumerate(data):
            if i == 3:
                tr.set_epoch(7, 150)                     # a new learning rate reaches a replayed step too
            loss, _ = tr.step(x, t)
            losses.append(float(loss))                   # read before the next replay overwrites the graph's loss tensor
        torch.cuda.synchronize()
        bufs = {k: v.clone() for k, v in tr.model.named_buffers()}
        runs.append((losses, tr.flat.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), bufs, tr.step_count, tr))
    (la, pa, ma, va, ba, sa, tra), (lb, pb, mb, vb, bb, sb, trb) = runs
    assert trb._graphs, "the second trainer never captured a graph"
    assert la == lb, (la, lb)
    assert sa == sb == steps and int(trb._step_dev) == steps
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    for k in ba:
        assert torch.equal(ba[k], bb[k]), k
    assert all(l == l for l in la) and la[0] != la[-1]
    assert abs(trb.lr - poly_lr(5e-4, 7, 150)) < 1e-15 and float(trb._lr_dev) == trb.lr


def test_replayed_step_draws_new_dropout_masks_every_step():
    """Same batch twice: were the masks frozen into the graph, steps 3 and 4 of a model with lr = 0 would give equal losses."""
    tr = _make(True, True, lr=0.0, weight_decay=0.0)
    x, t = _batches(2, 1)[0]
    losses = []
    for _ in range(5):
        loss, _ = tr.step(x, t)
        losses.append(float(loss))
    assert tr._graphs
    assert len(set(losses[2:])) == 3, losses              # replayed steps: three different mask draws, three different losses
    assert int(tr._seed_state[1]) == 5 * 13                # the 13 encoder blocks with p > 0 draw a mask per step


def test_graph_and_shape_change():
    tr = _make(True, True)
    (x, t), = _batches(2, 1)
    (x2, t2), = _batches(1, 1, 32, 64)
    for _ in range(3):
        tr.step(x, t)
    for _ in range(3):
        tr.step(x2, t2)
    loss, _ = tr.step(x, t)
    assert len(tr._graphs) == 2 and tr.step_count == 7 and float(loss) == float(loss)


def test_device_seeded_masks_equal_host_seeded_masks():
    """erf_dropout2d_mask_many_counter draws what erf_dropout2d_mask_many draws for seeds base + (count + s + 1) * K, and
    moves the device count on by the number of segments."""
    import numpy as np
    from erfnet_pytorch_amd._lib import call
    from erfnet_pytorch_amd import ops
    K = 0xD1B54A32D192ED03
    base, count0 = 0x0123456789ABCDEF, 41
    counts = np.array([6 * 64, 6 * 128, 6 * 128], dtype=np.int32)
    ps = np.array([0.03, 0.3, 0.3], dtype=np.float32)
    total = int(counts.sum())
    state = torch.tensor([base, count0], dtype=torch.int64, device=DEV)
    a = torch.empty(total, device=DEV)
    b = torch.empty(total, device=DEV)
    for rep in range(2):                                   # the second call continues from the advanced count
        call("erf_dropout2d_mask_many_counter", a.data_ptr(), 3, counts.ctypes.data, ps.ctypes.data, state.data_ptr(), ops._stream())
        seeds = np.array([(base + (count0 + 3 * rep + s + 1) * K) & 0xFFFFFFFFFFFFFFFF for s in range(3)], dtype=np.uint64)
        call("erf_dropout2d_mask_many", b.data_ptr(), 3, counts.ctypes.data, ps.ctypes.data, seeds.ctypes.data, ops._stream())
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        assert int(state[1]) == count0 + 3 * (rep + 1) and int(state[0]) == base
    keep = float((a[: 6 * 64] > 0).float().mean())
    assert 0.9 < keep <= 1.0 and float(a.max()) == pytest.approx(1 / 0.7, rel=1e-6)


def test_adam_with_device_step_equals_adam_with_host_step():
    from erfnet_pytorch_amd._lib import call
    from erfnet_pytorch_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    n = 100_003
    p0, gr = torch.randn(n, device=DEV, generator=g), torch.randn(n, device=DEV, generator=g)
    m0, v0 = torch.randn(n, device=DEV, generator=g) * 0.1, torch.rand(n, device=DEV, generator=g) * 0.01
    lr, b1, b2, eps, wd = 3.7e-4, 0.9, 0.999, 1e-8, 1e-4
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    step = torch.tensor([6], dtype=torch.int32, device=DEV)
    lr_dev = torch.tensor([lr], dtype=torch.float64, device=DEV)
    hyper = torch.zeros(2, device=DEV)
    for k in (7, 8):
        call("erf_adam_fused_arena", pa.data_ptr(), gr.data_ptr(), ma.data_ptr(), va.data_ptr(), n, lr, b1, b2, eps, wd, k, 1.0, ops._stream())
        call("erf_adam_advance", step.data_ptr(), lr_dev.data_ptr(), hyper.data_ptr(), b1, b2, ops._stream())
        call("erf_adam_fused_arena_hyper", pb.data_ptr(), gr.data_ptr(), mb.data_ptr(), vb.data_ptr(), n, hyper.data_ptr(), b1, b2, eps, wd,
             1.0, ops._stream())
        torch.cuda.synchronize()
        assert int(step) == k
        # the two step-dependent factors are formed in double on either side; device and host pow() may differ in the last
        # place of the fp32 factor, one ulp of the update
        assert float((pa - pb).abs().max()) <= 2e-7 * float(pa.abs().max())
        assert torch.equal(ma, mb) and torch.equal(va, vb)

