"""ImageNet pre-training variant (encoder + MaxPool/AvgPool/Linear head, CrossEntropyLoss, SGD) on the GPU
kernels against the golden vectors generated from the reference's train/erfnet_imagenet.py (G6) and the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from erfnet_pytorch_amd import synthetic as S
from tests.golden_util import check_packed, load, pre_bn_bias, rel_err
from oracle import erfnet_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _model():
    from erfnet_pytorch_amd import erfnet_imagenet as EI
    m = EI.ERFNet(1000)
    sd = O.synthetic_state(O.imagenet_template(1000))
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    return m.to(DEV), sd


def test_head_kernels_vs_torch():
    from erfnet_pytorch_amd.erfnet_imagenet import _LinearFn, _PoolMeanFn
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
        x = (torch.randn(3, 128, 28, 28, generator=torch.Generator().manual_seed(3)) * 2).to(dtype)
        xr = x.float().requires_grad_(True)
        ref = F.avg_pool2d(F.max_pool2d(xr, 2, stride=2), 14, 1, 0)
        gy = torch.randn(3, 128, 1, 1, generator=torch.Generator().manual_seed(4))
        ref.backward(gy)
        xd = x.to(DEV).contiguous(memory_format=torch.channels_last).detach().requires_grad_(True)
        y = _PoolMeanFn.apply(xd)
        y.backward(gy.to(DEV))
        assert rel_err(y.cpu(), ref.detach()) <= tol
        assert rel_err(xd.grad.float().cpu(), xr.grad) <= tol
    f = torch.randn(5, 128, generator=torch.Generator().manual_seed(5)).requires_grad_(True)
    w = (torch.randn(1000, 128, generator=torch.Generator().manual_seed(6)) * 0.1).requires_grad_(True)
    b = torch.randn(1000, generator=torch.Generator().manual_seed(7)).requires_grad_(True)
    gy = torch.randn(5, 1000, generator=torch.Generator().manual_seed(8))
    F.linear(f, w, b).backward(gy)
    fd, wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (f, w, b))
    y = _LinearFn.apply(fd, wd, bd)
    y.backward(gy.to(DEV))
    assert rel_err(y.detach().cpu(), F.linear(f, w, b).detach()) <= 1e-5
    for mine, ref in ((fd.grad, f.grad), (wd.grad, w.grad), (bd.grad, b.grad)):
        assert rel_err(mine.cpu(), ref) <= 1e-5


def test_eval_logits_vs_golden():
    g = load("g6_imagenet.npz")
    m, _ = _model()
    m.eval()
    x = S.synthetic_images(2, 224, 224, seed=60).to(DEV)
    with torch.no_grad():
        y = m(x)
    check_packed(g, "logits_eval", y.cpu(), 5e-4)
    m.compute_dtype = torch.bfloat16
    with torch.no_grad():
        yb = m(x)
    assert rel_err(yb.cpu(), y.cpu()) <= 5e-2            # bf16 activations through 37 convolutions
    with pytest.raises(ValueError, match="224x224"):
        m(torch.zeros(1, 3, 128, 256, device=DEV))


def test_train_step_and_sgd_vs_golden():
    from erfnet_pytorch_amd.trainer import ArenaSGD
    g = load("g6_imagenet.npz")
    m, sd = _model()
    m.train()
    for name, mod in m.named_modules():
        if "mask_" + name in g.files:
            mod.__dict__["_erf_forced_mask"] = torch.from_numpy(g["mask_" + name])
    x = S.synthetic_images(2, 224, 224, seed=60).to(DEV)
    t = torch.from_numpy(g["target"]).to(DEV)
    tr = ArenaSGD(m, lr=0.1, momentum=0.9, weight_decay=1e-4)
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    loss, y = tr.step(x, t)
    gold = float(g["loss"])
    assert abs(loss.item() - gold) <= 2e-4 * abs(gold), (loss.item(), gold)
    check_packed(g, "logits_train", y.cpu(), 1e-3)
    names = [str(s) for s in g["param_names"]]
    gn = g["gradnorm"]
    worst = 0.0
    for i, k in enumerate(names):
        o, n = tr.offsets[k]
        mine = tr.gflat[o:o + n].double().norm().item()
        if pre_bn_bias(k):
            assert mine < 2e-3 and gn[i] < 2e-3           # analytically zero; the reference holds rounding noise
            continue
        worst = max(worst, abs(mine - gn[i]) / max(gn[i], 1e-6))
    assert worst <= 2e-2, worst
    # SGD step: p' = p - lr * (g + wd * p) on the first step (momentum buffer = g)
    for k, p in m.named_parameters():
        o, n = tr.offsets[k]
        want = before[k].reshape(-1) - 0.1 * (tr.gflat[o:o + n] + 1e-4 * before[k].reshape(-1))
        assert rel_err(p.detach().reshape(-1).cpu(), want.cpu()) <= 1e-6, k
    ps = np.array([p.detach().double().sum().item() for _, p in m.named_parameters()])
    gold_ps = g["paramsum"]
    for i, k in enumerate(names):
        if pre_bn_bias(k):
            continue
        assert abs(ps[i] - gold_ps[i]) <= 2e-3 * max(abs(gold_ps[i]), 1.0) + 0.1 * 2e-2 * gn[i] * np.sqrt(tr.offsets[k][1]), k


def test_sgd_step_replayed_from_a_graph_equals_the_launched_step():
    """ArenaSGD(graph=True): steps 0 and 1 run launched, step 2 is captured and replayed, steps 3-4 are replays; dropout on (the
    seed counter lives on the device in both modes).  Loss, gradients, parameters and momentum buffers bit for bit."""
    from erfnet_pytorch_amd.trainer import ArenaSGD
    x = S.synthetic_images(2, 224, 224, seed=61).to(DEV)
    t = torch.randint(0, 1000, (2,), generator=torch.Generator().manual_seed(5)).to(DEV)
    res = []
    for graph in (False, True):
        torch.manual_seed(77)                       # the dropout seed state follows torch's seed
        m, _ = _model()
        m.train()
        m.compute_dtype = torch.bfloat16
        tr = ArenaSGD(m, lr=0.05, momentum=0.9, weight_decay=1e-4, graph=graph)
        losses = []
        for _ in range(5):
            loss, _ = tr.step(x, t)
            losses.append(float(loss))
        torch.cuda.synchronize()
        assert bool(tr._graphs) == graph and tr.step_count == 5
        res.append((losses, tr.gflat.cpu().numpy().copy(), tr.flat.cpu().numpy().copy(), tr.buf.cpu().numpy().copy()))
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert np.array_equal(a, b)


def test_sgd_schedule_changes_reach_the_replayed_step_without_a_new_capture():
    """imagenet/main.py:276-286 moves the learning rate every epoch and drops the weight decay at a milestone.  Both are
    read from device memory by the update kernel, so ONE captured graph serves every value: seven steps with lr and wd
    changed between replays must equal the launched run bit for bit, and the graph table must keep its single entry."""
    from erfnet_pytorch_amd.trainer import ArenaSGD
    x = S.synthetic_images(2, 224, 224, seed=62).to(DEV)
    t = torch.randint(0, 1000, (2,), generator=torch.Generator().manual_seed(6)).to(DEV)
    schedule = {3: (0.05 * 0.95, 1e-4), 4: (0.05 * 0.95 ** 2, 0.0), 6: (0.05 * 0.95 ** 3, 0.0)}
    res = []
    for graph in (False, True):
        torch.manual_seed(78)
        m, _ = _model()
        m.train()
        m.compute_dtype = torch.bfloat16
        tr = ArenaSGD(m, lr=0.05, momentum=0.9, weight_decay=1e-4, graph=graph)
        losses = []
        for i in range(7):
            if i in schedule:
                tr.lr, tr.wd = schedule[i]
            loss, _ = tr.step(x, t)
            losses.append(float(loss))
        torch.cuda.synchronize()
        assert len(tr._graphs) == (1 if graph else 0)
        res.append((losses, tr.flat.cpu().numpy().copy(), tr.buf.cpu().numpy().copy()))
    assert res[0][0] == res[1][0]
    for a, b in zip(res[0][1:], res[1][1:]):
        assert np.array_equal(a, b)
