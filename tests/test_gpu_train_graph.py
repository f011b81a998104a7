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
