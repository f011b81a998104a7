"""The training path the benchmark times -- Trainer(fused_head_loss=True), fp32 and bf16 activations -- against the
reference's three optimisation steps (g5_train_steps.npz: train/main.py:196-205 run by the reference's own modules with
torch.optim.Adam and the poly schedule) and against exact arithmetic.

Full-network gradients of this 74-convolution / 39-BatchNorm graph on 2 x 64 x 128 inputs cannot be compared element by
element at 1e-4: the reference's OWN fp32 gradients sit up to 1e-2 away from the same graph evaluated in float64 (ReLU gates
of pre-activations within rounding distance of zero open differently under any change of summation order, and every 1/std
amplifies rounding).  So the yardstick is the float64 evaluation of the oracle, and the bar is the reference's own distance
from it: per parameter tensor, our relative L2 error against float64 may be at most 3x the error of the reference's fp32
arithmetic (the oracle in fp32, itself pinned to the reference's stored gradients) plus 1e-4.  The element-wise 1e-5 / 1e-4
bars live in test_gpu_parity_strict.py, on inputs whose gates are decisive.
bf16: the yardstick is the oracle evaluated with bf16 storage (same rounding points as the engine); stated bounds below.
"""
import numpy as np
import pytest
import torch

from erfnet_pytorch_amd import synthetic as S
from oracle import erfnet_oracle as O
from tests.golden_util import load, rel_err, pre_bn_bias, unpack_full

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W = torch.tensor(O.CITYSCAPES_WEIGHTS)


def _masks(g, step, dtype=torch.float32):
    return {str(k): torch.from_numpy(g[f"s{step}_mask_{k}"]).to(dtype) for k in g[f"s{step}_masks__keys"]}


def _force_masks(model, g, step):
    for name, mod in model.named_modules():
        key = f"s{step}_mask_{name}"
        if key in g.files:
            mod.__dict__["_erf_forced_mask"] = torch.from_numpy(g[key])


def _build(sd, bf16):
    from erfnet_pytorch_amd import erfnet as E
    m = E.ERFNet(20)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    m.compute_dtype = torch.bfloat16 if bf16 else None
    return m


def _l2(a, b, floor=0.0):
    a, b = a.double().reshape(-1).cpu(), b.double().reshape(-1).cpu()
    return float((a - b).norm()) / max(float(b.norm()), floor, 1e-30)


def _oracle_trajectory(g, steps=3, storage_dtype=None):
    """The reference recipe's first optimisation steps replayed by the oracle on the CPU (tests/test_oracle_golden.py pins this
    very trajectory to the reference's).  Per step: the state before it (parameters + buffers, Adam moments), the loss, the
    gradients, and the state after it."""
    sd = O.synthetic_state(O.erfnet_template(20))
    names = [str(s) for s in g["param_names"]]
    m = [torch.zeros_like(sd[k]) for k in names]
    v = [torch.zeros_like(sd[k]) for k in names]
    out = []
    for step in range(steps):
        lr = O.poly_lr(5e-4, step + 1, 150)
        x = S.synthetic_images(2, 64, 128, seed=50 + step)
        t = S.synthetic_labels(2, 64, 128, 20, seed=50 + step)
        before = ({k: v_.clone() for k, v_ in sd.items()}, [a.clone() for a in m], [a.clone() for a in v])
        loss, grads, new_running, _, _ = O.train_step_grads(sd, x, t[:, 0], W, _masks(g, step), storage_dtype=storage_dtype)
        O.adam_step([sd[k] for k in names], [grads[k] for k in names], m, v, step + 1, lr=lr)
        sd.update(new_running)
        out.append({"before": before, "x": x, "t": t, "lr": lr, "loss": float(loss), "grads": grads,
                    "after": ({k: v_.clone() for k, v_ in sd.items()}, [a.clone() for a in m], [a.clone() for a in v])})
    return names, out


@pytest.mark.parametrize("bf16", [False, True])
def test_benchmarked_trainer_path_step_by_step_from_the_reference_trajectory(bf16):
    """Steps 0, 1 and 2 of the reference recipe, each started from the REFERENCE trajectory's state (parameters, BatchNorm
    buffers, Adam moments and step count as the oracle's replay of train/main.py:196-205 leaves them), so that every step is
    pinned by itself: free-running trajectories part ways after the first Adam step because an update is ~lr * sign(g) and
    noise-level gradient elements take either sign (the reference run with 1 and with 8 CPU threads differs by 1e-1 in its own
    step-2 logits).  Checked per step:
      loss                 fp32 2e-5 (measured 6e-7), bf16 2e-3 against the bf16-storage evaluation;
      gradients, fp32      per parameter tensor, relative L2 distance from the float64 evaluation <= 5x the reference
                           arithmetic's own distance + 2e-3 (measured: median 2.7x -- the matrix units accumulate a K = 3C
                           dot product as one sequential fmaf chain, oneDNN in blocked partial sums);
      gradients, bf16      relative L2 distance from the bf16-storage evaluation <= 2x that evaluation's own distance from
                           float64 + 1e-2 (measured <= 1.34x).  On this fixture bf16 storage ALONE moves the encoder's
                           gradients by more than 100 % (random weights, batch statistics over as few as 256 samples, 39
                           stacked BatchNorms), so only the last decoder layers are pinned tightly here (1e-2 .. 9e-2); the
                           tight bf16 bars are in test_gpu_parity_strict.py;
      optimiser            parameters and both Adam moments after the step == torch.optim.Adam's arithmetic applied to our
                           gradient from the same state, element by element (1e-6), at step counts 1, 2, 3; fp32 also against
                           the reference's own moments / updates;
      running statistics   fp32 1e-5, bf16 3e-2 (deep layers: the bf16-storage evaluation itself is only that well defined)."""
    from erfnet_pytorch_amd import ops
    from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
    g = load("g5_train_steps.npz")
    names, traj = _oracle_trajectory(g, storage_dtype=torch.bfloat16 if bf16 else None)
    model = _build(traj[0]["before"][0], bf16)
    tr = Trainer(model, CITYSCAPES_WEIGHTS, fused_head_loss=True)
    params, bufs = dict(model.named_parameters()), dict(model.named_buffers())
    for step, ent in enumerate(traj):
        sd0, m0, v0 = ent["before"]
        with torch.no_grad():
            for k in names:
                params[k].copy_(sd0[k])                               # views of the trainer's flat parameter arena
                o, n = tr.offsets[k]
                tr.exp_avg[o:o + n].copy_(m0[names.index(k)].reshape(-1))
                tr.exp_avg_sq[o:o + n].copy_(v0[names.index(k)].reshape(-1))
            for k, b in bufs.items():
                b.copy_(sd0[k])
        ops.bump_param_epoch()
        tr.step_count = step
        tr.set_epoch(step + 1, 150)
        assert abs(tr.lr - ent["lr"]) < 1e-15
        _force_masks(model, g, step)
        loss, _ = tr.step(ent["x"].to(DEV), ent["t"].to(DEV))
        ltol = 2e-5 if not bf16 else 2e-3
        assert abs(loss.item() - ent["loss"]) <= ltol * abs(ent["loss"]), (step, loss.item(), ent["loss"])
        # gradients
        sd64 = {k: (v_.double() if v_.dtype.is_floating_point else v_) for k, v_ in sd0.items()}
        g64 = O.train_step_grads(sd64, ent["x"].double(), ent["t"][:, 0], W.double(), _masks(g, step, torch.float64))[1]
        rows = []
        for k in names:
            o, n = tr.offsets[k]
            mine = tr.gflat[o:o + n].view_as(params[k]).cpu()
            ref = ent["grads"][k]
            if ref is None:
                assert float(mine.abs().sum()) == 0.0, k              # encoder.output_conv takes no part in the full forward
                continue
            if pre_bn_bias(k):
                assert float(mine.abs().max()) == 0.0, k              # analytically zero: written as exact zeros
                continue
            if not bf16:
                e_ours, e_ref = _l2(mine, g64[k]), _l2(ref, g64[k])
            else:
                e_ours, e_ref = _l2(mine, ref), _l2(ref, g64[k])
            rows.append((e_ours, e_ref, k))
        for e_ours, e_ref, k in rows:
            if not bf16:
                assert e_ours <= 5 * e_ref + 2e-3, (step, k, e_ours, e_ref)
            else:
                assert e_ours <= 2 * e_ref + 1e-2, (step, k, e_ours, e_ref)
        # optimiser arithmetic: the state after the step is torch.optim.Adam's restatement applied to OUR gradient from the
        # same state (element by element), whatever the step count and the moments were
        sd1, m1, v1 = ent["after"]
        upd = []
        for i, k in enumerate(names):
            o, n = tr.offsets[k]
            if ent["grads"][k] is None:
                assert torch.equal(params[k].detach().cpu(), sd0[k]), k                      # unused head: no weight decay either
                continue
            gk = tr.gflat[o:o + n].view_as(params[k]).cpu()
            pk, mk, vk = sd0[k].clone(), m0[i].clone(), v0[i].clone()
            O.adam_step([pk], [gk], [mk], [vk], step + 1, lr=tr.lr)
            assert rel_err(params[k].detach().cpu(), pk) <= 1e-6, (step, k)
            assert rel_err(tr.exp_avg[o:o + n].cpu().view_as(mk), mk) <= 1e-6 and rel_err(tr.exp_avg_sq[o:o + n].cpu().view_as(vk), vk) <= 1e-6, (step, k)
            if pre_bn_bias(k):
                continue        # the reference moves these by +-lr with the sign of rounding noise; here: weight decay only
            du_ref = (sd1[k] - sd0[k]).reshape(-1)
            du = params[k].detach().cpu().reshape(-1) - sd0[k].reshape(-1)
            far = float(((du - du_ref).abs() > 0.1 * tr.lr).float().mean())
            upd.append((_l2(tr.exp_avg[o:o + n], m1[i]), _l2(tr.exp_avg_sq[o:o + n], v1[i]), far, k))
        if not bf16:
            # against the reference's own optimiser trajectory (its gradient, its moments): measured worst 3.1e-2 / 3.5e-2, and
            # at most 4.7 % of a tensor's elements step more than lr/10 away from the reference's update
            for em, ev, far, k in upd:
                assert em <= 6e-2 and ev <= 7e-2 and far <= 0.08, (step, k, em, ev, far)
        for k, b in bufs.items():
            if "num_batches" in k:
                assert int(b) == int(sd1[k]), k
            else:
                assert rel_err(b.cpu(), sd1[k]) <= (1e-5 if not bf16 else 3e-2), (step, k, rel_err(b.cpu(), sd1[k]))


@pytest.mark.parametrize("bf16", [False, True])
def test_benchmarked_trainer_path_free_running_three_steps_vs_golden(bf16):
    from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS, poly_lr
    g = load("g5_train_steps.npz")
    sd = O.synthetic_state(O.erfnet_template(20))
    model = _build(sd, bf16)
    tr = Trainer(model, CITYSCAPES_WEIGHTS, fused_head_loss=True)
    assert tr.fused_head_loss
    names = [str(s) for s in g["param_names"]]
    params = dict(model.named_parameters())
    losses = []
    for step in range(3):
        tr.set_epoch(step + 1, 150)
        assert abs(tr.lr - float(g[f"s{step}_lr"])) < 1e-12
        _force_masks(model, g, step)
        x = S.synthetic_images(2, 64, 128, seed=50 + step).to(DEV)
        t = S.synthetic_labels(2, 64, 128, 20, seed=50 + step).to(DEV)
        loss, logits = tr.step(x, t)
        assert logits is None                                         # the logits never exist on this path
        losses.append(loss.item())
        if step == 0:
            # parameters after the first optimiser step against the reference's torch.optim.Adam: |update| <= lr per element,
            # and a noise-level gradient whose sign differs moves an element by up to 2 lr
            ps = np.array([params[k].detach().double().sum().item() for k in names])
            n_el = np.array([params[k].numel() for k in names])
            keep = np.array([not pre_bn_bias(k) for k in names])
            flips = np.maximum(np.ceil((0.02 if not bf16 else 0.20) * n_el), 2)
            lim = 1e-5 * np.maximum(g["s0_paramabs"], 1e-3) + flips * 2 * tr.lr
            assert np.all((np.abs(ps - g["s0_paramsum"]) <= lim)[keep])
            bufs = dict(model.named_buffers())
            bs = np.array([bufs[str(k)].double().sum().item() for k in g["buffer_names"]])
            ref = g["s0_bufsum"]
            assert np.all(np.abs(bs - ref) <= (1e-5 if not bf16 else 8e-2) * np.maximum(np.abs(ref), 1.0))
    gold = [float(g[f"s{i}_loss"]) for i in range(3)]
    # free-running losses: step 0 is pinned; after that the trajectories part ways (see the step-by-step test) -- measured
    # 1.3e-3 / 1.7e-3 (fp32) and 2.4e-3 / 5.2e-3 (bf16) at steps 1 / 2
    ltol = [1e-5, 4e-3, 6e-3] if not bf16 else [3e-3, 8e-3, 1.5e-2]
    for i in range(3):
        assert abs(losses[i] - gold[i]) <= ltol[i] * abs(gold[i]), (i, losses[i], gold[i])


def test_trainer_step_at_the_benchmark_size():
    """BASELINE configs[2] at its full size -- batch 6, 3 x 512 x 1024, bf16 activations, the fused head+loss path the
    benchmark times -- through size-independent properties: the step is deterministic (two fresh trainers, same forced
    Dropout2d masks: bit-identical loss, gradients and updated parameters), the loss is finite and equals the loss of the
    stored-logits path, and the gradient arena matches that path (bf16 patch rounding aside, tensors agree to 2e-2 in L2)."""
    from erfnet_pytorch_amd import erfnet as E
    from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
    n, h, w = 6, 512, 1024
    sd = O.synthetic_state(O.erfnet_template(20))
    x = S.synthetic_images(n, h, w, seed=21).to(DEV)
    t = S.synthetic_labels(n, h, w, 20, seed=21).to(DEV)

    def run(fused):
        m = _build(sd, True)
        for i, mod in enumerate(mm for mm in m.modules() if isinstance(mm, E.non_bottleneck_1d) and mm.dropout.p != 0):
            c = mod.conv3x1_1.out_channels
            u = torch.from_numpy(S.unit_uniform(n * c, S.key_seed("fullmask", i)).astype(np.float32)).reshape(n, c)
            mod.__dict__["_erf_forced_mask"] = (u > 2 * mod.dropout.p - 1).float() / (1 - mod.dropout.p)
        tr = Trainer(m, CITYSCAPES_WEIGHTS, fused_head_loss=fused)
        loss, _ = tr.step(x, t)
        torch.cuda.synchronize()
        return loss.item(), tr.gflat.clone(), tr.flat.clone(), tr

    la, ga, pa, tr = run(True)
    lb, gb, pb, _ = run(True)
    assert np.isfinite(la) and la == lb and torch.equal(ga, gb) and torch.equal(pa, pb)
    lc, gc, _, _ = run(False)
    assert abs(la - lc) <= 1e-5 * abs(lc), (la, lc)
    for k, (o, cnt) in tr.offsets.items():
        if float(gc[o:o + cnt].abs().max()) == 0.0:
            assert float(ga[o:o + cnt].abs().max()) == 0.0, k
            continue
        assert _l2(ga[o:o + cnt], gc[o:o + cnt]) <= 2e-2, (k, _l2(ga[o:o + cnt], gc[o:o + cnt]))
