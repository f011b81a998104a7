"""Generate tests/golden/*.npz by EXECUTING THE REFERENCE's own modules.

Run in the build container only (the reference tree does not exist on the GPU box):

    python tests/golden/make_golden.py

It imports /root/reference/train/{erfnet,iouEval,erfnet_imagenet}.py, loads the
closed-form weights of erfnet_pytorch_amd.synthetic into them and stores inputs-by-seed
plus expected outputs.  No reference source or bytecode is written anywhere.  Large
tensors are stored as a strided sample plus float64 sums (see `pack`).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/train")

import erfnet as ref_erfnet                  # noqa: E402  (reference)
import multi_models as ref_multi             # noqa: E402  (reference)
import erfnet_imagenet as ref_imagenet       # noqa: E402  (reference)
import iouEval as ref_iou                    # noqa: E402  (reference)

from erfnet_pytorch_amd import synthetic as S  # noqa: E402

torch.set_num_threads(8)
MAX_FULL = 24000
CITYSCAPES_WEIGHTS = [
    2.8149201869965, 6.9850029945374, 3.7890393733978, 9.9428062438965, 9.7702074050903,
    9.5110931396484, 10.311357498169, 10.026463508606, 4.6323022842407, 9.5608062744141,
    7.8698215484619, 9.5168733596802, 10.373730659485, 6.6616044044495, 10.260489463806,
    10.287888526917, 10.289801597595, 10.405355453491, 10.138095855713, 0.0,
]


def pack(out: dict, name: str, t: torch.Tensor):
    """Store t fully if small, else a strided flat sample + sum + abs-sum (float64)."""
    a = t.detach().cpu().contiguous()
    flat = a.reshape(-1)
    n = flat.numel()
    stride = 1 if n <= MAX_FULL else -(-n // MAX_FULL)
    out[name + "__shape"] = np.array(a.shape, dtype=np.int64)
    out[name + "__stride"] = np.array(stride, dtype=np.int64)
    out[name + "__sample"] = flat[::stride].numpy().copy()
    if a.dtype.is_floating_point:
        out[name + "__sum"] = np.array(flat.double().sum().item())
        out[name + "__abssum"] = np.array(flat.double().abs().sum().item())


def upstream(shape, seed):
    n = int(np.prod(shape))
    return torch.from_numpy(S.unit_uniform(n, S.key_seed("upstream", seed)).astype(np.float32).reshape(shape))


def activations(shape, seed):
    """Block inputs: non-negative-ish like post-ReLU maps, with exact zeros."""
    n = int(np.prod(shape))
    u = S.unit_uniform(n, S.key_seed("act", seed))
    return torch.from_numpy(np.maximum(u * 1.5 - 0.2, 0).astype(np.float32).reshape(shape))


def capture_masks(model):
    """Forward hooks on every Dropout2d: recover the [N,C,1,1] multiplier actually applied."""
    masks, handles = {}, []
    for name, mod in model.named_modules():
        if isinstance(mod, torch.nn.Dropout2d) and mod.p != 0:
            def hook(m, inp, out, name=name):
                if m.training:
                    keep = (out.detach().abs().sum(dim=(2, 3), keepdim=True) > 0).float()
                    masks[name.rsplit(".dropout", 1)[0]] = keep / (1.0 - m.p)
            handles.append(mod.register_forward_hook(hook))
    return masks, handles


# ----------------------------------------------------------------------------- G1
def g1_full_eval():
    out = {}
    for nc in (20, 6):
        m = S.load_synthetic(ref_erfnet.ERFNet(nc)).eval()
        for tag, (n, h, w, seed) in {"a": (1, 64, 128, 1), "b": (2, 128, 256, 2)}.items():
            if nc == 6 and tag == "b":
                continue
            x = S.synthetic_images(n, h, w, seed=seed)
            with torch.no_grad():
                y = m(x)
            top2 = y.topk(2, dim=1).values
            out[f"nc{nc}_{tag}__in"] = np.array([n, h, w, seed], dtype=np.int64)
            pack(out, f"nc{nc}_{tag}_logits", y)
            out[f"nc{nc}_{tag}_argmax"] = y.max(1)[1].numpy().astype(np.uint8)
            out[f"nc{nc}_{tag}_margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32)
        if nc == 20:
            x = S.synthetic_images(1, 64, 128, seed=1)
            with torch.no_grad():
                pack(out, "nc20_a_only_encode", m(x, only_encode=True))
                pack(out, "nc20_a_encoder_feat", m.encoder(x))
    np.savez_compressed(os.path.join(HERE, "g1_full_eval.npz"), **out)


# ----------------------------------------------------------------------------- G2
def block_case(out, tag, mod, prefix, x, seed, p_drop=0.0):
    """Eval forward; train forward/backward with fixed upstream grad; running stats."""
    sd = S.fill_state_dict({prefix + "." + k: v for k, v in mod.state_dict().items()})
    mod.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()})
    out[tag + "__prefix"] = np.array(prefix)
    out[tag + "__xshape"] = np.array(x.shape, dtype=np.int64)
    out[tag + "__seed"] = np.array(seed, dtype=np.int64)
    mod.eval()
    with torch.no_grad():
        pack(out, tag + "_y_eval", mod(x))
    mod.train()
    masks, handles = capture_masks(mod) if p_drop else ({}, [])
    torch.manual_seed(1234 + seed)
    xr = x.clone().requires_grad_(True)
    y = mod(xr)
    dy = upstream(y.shape, seed)
    y.backward(dy)
    for h in handles:
        h.remove()
    pack(out, tag + "_y_train", y)
    pack(out, tag + "_dx", xr.grad)
    for k, v in mod.named_parameters():
        pack(out, tag + "_grad_" + k, v.grad)
    for k, v in mod.named_buffers():
        if "num_batches" not in k:
            pack(out, tag + "_buf_" + k, v)
    for k, v in masks.items():
        out[tag + "_mask"] = v.numpy()


def g2_blocks():
    out = {}
    seed = 10
    for (c, d, p, shape) in [(64, 1, 0.03, (2, 64, 24, 40)), (16, 1, 0.0, (2, 16, 24, 48)),
                             (128, 2, 0.3, (2, 128, 16, 24)), (128, 4, 0.3, (2, 128, 16, 24)),
                             (128, 8, 0.3, (2, 128, 24, 16)), (128, 16, 0.3, (1, 128, 40, 40)),
                             (64, 1, 0.0, (1, 64, 16, 24))]:
        seed += 1
        mod = ref_erfnet.non_bottleneck_1d(c, p, d)
        block_case(out, f"nb_c{c}_d{d}_p{int(p * 100)}", mod, f"blk{seed}", activations(shape, seed), seed, p)
    for (cin, cout, shape) in [(3, 16, (2, 3, 32, 48)), (16, 64, (2, 16, 24, 32)), (64, 128, (2, 64, 16, 24))]:
        seed += 1
        x = S.synthetic_images(shape[0], shape[2], shape[3], seed=seed) if cin == 3 else activations(shape, seed)
        block_case(out, f"down_{cin}_{cout}", ref_erfnet.DownsamplerBlock(cin, cout), f"blk{seed}", x, seed)
    for (cin, cout, shape) in [(128, 64, (2, 128, 8, 12)), (64, 16, (2, 64, 12, 16))]:
        seed += 1
        # decoder upsampler weights are ConvTranspose2d: name the prefix so synthetic.fill_tensor scales them so
        block_case(out, f"up_{cin}_{cout}", ref_erfnet.UpsamplerBlock(cin, cout), f"decoder.layers.{seed}",
                   activations(shape, seed), seed)
    for nc in (20, 6):
        seed += 1
        mod = torch.nn.ConvTranspose2d(16, nc, 2, stride=2, padding=0, output_padding=0, bias=True)
        sd = S.fill_state_dict({"decoder.output_conv." + k: v for k, v in mod.state_dict().items()}, salt=seed)
        mod.load_state_dict({k.rsplit(".", 1)[-1]: v for k, v in sd.items()})
        x = activations((2, 16, 12, 16), seed).requires_grad_(True)
        y = mod(x)
        dy = upstream(y.shape, seed)
        y.backward(dy)
        tag = f"outconv_{nc}"
        out[tag + "__seed"] = np.array(seed, dtype=np.int64)
        pack(out, tag + "_y", y)
        pack(out, tag + "_dx", x.grad)
        pack(out, tag + "_grad_weight", mod.weight.grad)
        pack(out, tag + "_grad_bias", mod.bias.grad)
    np.savez_compressed(os.path.join(HERE, "g2_blocks.npz"), **out)


# ----------------------------------------------------------------------------- G3
def g3_loss():
    out = {}
    w = torch.tensor(CITYSCAPES_WEIGHTS, dtype=torch.float32)
    for tag, (n, h, w_, seed, all_ignored) in {"mixed": (2, 16, 24, 31, False), "ignored": (1, 8, 8, 32, True)}.items():
        z = (3.0 * upstream((n, 20, h, w_), seed)).requires_grad_(True)
        t = S.synthetic_labels(n, h, w_, 20, seed=seed)[:, 0]
        if all_ignored:
            t = torch.full_like(t, 19)
        crit = torch.nn.CrossEntropyLoss(w)
        loss = crit(z, t)
        loss.backward()
        out[tag + "__in"] = np.array([n, h, w_, seed, int(all_ignored)], dtype=np.int64)
        out[tag + "_loss"] = np.array(loss.item(), dtype=np.float32)
        pack(out, tag + "_dlogits", z.grad)
        # the nn.Module spelled out in the reference (log_softmax + NLLLoss) must agree
        alt = torch.nn.NLLLoss(w)(torch.nn.functional.log_softmax(z.detach(), dim=1), t)
        out[tag + "_loss_nll"] = np.array(alt.item(), dtype=np.float32)
    # unweighted CE on class logits (imagenet/main.py:81)
    z = (2.0 * upstream((8, 1000), 33)).requires_grad_(True)
    t = S.synthetic_labels(8, 1, 1, 1000, seed=33, ignore_frac=0)[:, 0, 0, 0]
    loss = torch.nn.CrossEntropyLoss()(z, t)
    loss.backward()
    out["cls_loss"] = np.array(loss.item(), dtype=np.float32)
    pack(out, "cls_dlogits", z.grad)
    np.savez_compressed(os.path.join(HERE, "g3_loss.npz"), **out)


# ----------------------------------------------------------------------------- G4
def g4_iou():
    out = {}
    for nc in (20, 6, 4):
        ev = ref_iou.iouEval(nc)
        for b in range(3):
            seed = 40 + b
            pred = S.synthetic_labels(2, 32, 48, nc, seed=seed + 100, ignore_frac=0.0)
            lab = S.synthetic_labels(2, 32, 48, nc, seed=seed, ignore_frac=0.07 if nc == 20 else 0.0)
            if nc == 20:   # predictions of the ignore class occur too (argmax can return 19)
                pred = torch.where(S.synthetic_labels(2, 32, 48, 50, seed=seed + 200, ignore_frac=0) == 0,
                                   torch.full_like(pred, 19), pred)
            ev.addBatch(pred, lab)
            out[f"nc{nc}_b{b}_tp"] = ev.tp.numpy().copy()
            out[f"nc{nc}_b{b}_fp"] = ev.fp.numpy().copy()
            out[f"nc{nc}_b{b}_fn"] = ev.fn.numpy().copy()
        mean, per = ev.getIoU()
        out[f"nc{nc}_iou_mean"] = np.array(mean.item())
        out[f"nc{nc}_iou"] = per.numpy().copy()
    # inputs that already have nClasses planes (train/iouEval.py:37-38,45-46): one-hot predictions against a label map,
    # a label map against one-hot targets, and soft scores (softmax of seeded noise) against one-hot targets
    for nc in (20, 4):
        pred = S.synthetic_labels(2, 24, 40, nc, seed=471, ignore_frac=0.0)
        lab = S.synthetic_labels(2, 24, 40, nc, seed=472, ignore_frac=0.07 if nc == 20 else 0.0)
        onehot = lambda t: torch.zeros(t.size(0), nc, t.size(2), t.size(3)).scatter_(1, t, 1)
        soft = torch.softmax(3.0 * upstream((2, nc, 24, 40), 473), dim=1)
        for tag, (x, y) in {"onehot_x": (onehot(pred), lab), "onehot_y": (pred, onehot(lab)),
                            "soft_x": (soft, onehot(lab))}.items():
            ev = ref_iou.iouEval(nc)
            ev.addBatch(x, y)
            ev.addBatch(x, y)                     # accumulation over two calls
            for k in ("tp", "fp", "fn"):
                out[f"planes_nc{nc}_{tag}_{k}"] = getattr(ev, k).numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g4_iou.npz"), **out)


# ----------------------------------------------------------------------------- G5
def g5_train_steps():
    out = {}
    m = S.load_synthetic(ref_erfnet.ERFNet(20)).train()
    w = torch.tensor(CITYSCAPES_WEIGHTS, dtype=torch.float32)
    crit = torch.nn.CrossEntropyLoss(w)
    opt = torch.optim.Adam(m.parameters(), 5e-4, (0.9, 0.999), eps=1e-08, weight_decay=1e-4)
    num_epochs = 150
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: pow((1 - ((e - 1) / num_epochs)), 0.9))
    masks, handles = capture_masks(m)
    torch.manual_seed(5)
    for step in range(3):
        epoch = step + 1
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sched.step(epoch)
        out[f"s{step}_lr"] = np.array(opt.param_groups[0]["lr"])
        x = S.synthetic_images(2, 64, 128, seed=50 + step)
        t = S.synthetic_labels(2, 64, 128, 20, seed=50 + step)
        y = m(x)
        opt.zero_grad()
        loss = crit(y, t[:, 0])
        loss.backward()
        out[f"s{step}_loss"] = np.array(loss.item(), dtype=np.float32)
        out[f"s{step}_masks__keys"] = np.array(sorted(masks.keys()))
        for k in sorted(masks.keys()):
            out[f"s{step}_mask_{k}"] = masks[k].numpy().copy()
        names = [k for k, _ in m.named_parameters()]
        out["param_names"] = np.array(names)
        out[f"s{step}_gradnorm"] = np.array([(-1.0 if p.grad is None else p.grad.double().norm().item())
                                             for _, p in m.named_parameters()])
        if step == 0:
            for k, p in m.named_parameters():
                if p.grad is not None and (k.endswith("conv3x1_1.weight") and ("layers.1." in k or "layers.7." in k)
                                           or "initial_block" in k or "output_conv" in k):
                    pack(out, "s0_grad_" + k, p.grad)
        opt.step()
        out[f"s{step}_paramsum"] = np.array([p.detach().double().sum().item() for _, p in m.named_parameters()])
        out[f"s{step}_paramabs"] = np.array([p.detach().double().abs().sum().item() for _, p in m.named_parameters()])
        bn = [(k, b) for k, b in m.named_buffers() if "num_batches" not in k]
        out["buffer_names"] = np.array([k for k, _ in bn])
        out[f"s{step}_bufsum"] = np.array([b.double().sum().item() for _, b in bn])
        pack(out, f"s{step}_logits", y)
    st = opt.state[next(m.parameters())]
    pack(out, "final_exp_avg_first", st["exp_avg"])
    pack(out, "final_exp_avg_sq_first", st["exp_avg_sq"])
    for h in handles:
        h.remove()
    np.savez_compressed(os.path.join(HERE, "g5_train_steps.npz"), **out)


# ----------------------------------------------------------------------------- G6
def g6_imagenet():
    out = {}
    m = S.load_synthetic(ref_imagenet.ERFNet(1000)).eval()
    x = S.synthetic_images(2, 224, 224, seed=60)
    with torch.no_grad():
        pack(out, "logits_eval", m(x))
    m.train()
    masks, handles = capture_masks(m)
    torch.manual_seed(6)
    t = S.synthetic_labels(2, 1, 1, 1000, seed=60, ignore_frac=0)[:, 0, 0, 0]
    opt = torch.optim.SGD(m.parameters(), 0.1, momentum=0.9, weight_decay=1e-4)
    y = m(x)
    loss = torch.nn.CrossEntropyLoss()(y, t)
    opt.zero_grad()
    loss.backward()
    out["loss"] = np.array(loss.item(), dtype=np.float32)
    out["target"] = t.numpy()
    pack(out, "logits_train", y)
    out["param_names"] = np.array([k for k, _ in m.named_parameters()])
    out["gradnorm"] = np.array([p.grad.double().norm().item() for _, p in m.named_parameters()])
    out["masks__keys"] = np.array(sorted(masks.keys()))
    for k in sorted(masks.keys()):
        out["mask_" + k] = masks[k].numpy().copy()
    opt.step()
    out["paramsum"] = np.array([p.detach().double().sum().item() for _, p in m.named_parameters()])
    for h in handles:
        h.remove()
    np.savez_compressed(os.path.join(HERE, "g6_imagenet.npz"), **out)


# ----------------------------------------------------------------------------- G7
def g7_encoder_only():
    """Encoder stage of the training recipe (train/main.py:203 with enc=True): the 1x1 head alone, then one
    optimisation step of the whole network run with only_encode=True on targets at 1/8 resolution."""
    out = {}
    seed = 70
    for nc in (20, 6):
        seed += 1
        mod = torch.nn.Conv2d(128, nc, 1, stride=1, padding=0, bias=True)
        sd = S.fill_state_dict({"encoder.output_conv." + k: v for k, v in mod.state_dict().items()}, salt=seed)
        mod.load_state_dict({k.rsplit(".", 1)[-1]: v for k, v in sd.items()})
        x = activations((2, 128, 9, 14), seed).requires_grad_(True)
        y = mod(x)
        y.backward(upstream(y.shape, seed))
        tag = f"head_{nc}"
        out[tag + "__seed"] = np.array(seed, dtype=np.int64)
        pack(out, tag + "_y", y)
        pack(out, tag + "_dx", x.grad)
        pack(out, tag + "_grad_weight", mod.weight.grad)
        pack(out, tag + "_grad_bias", mod.bias.grad)
    m = S.load_synthetic(ref_erfnet.ERFNet(20)).train()
    crit = torch.nn.CrossEntropyLoss(torch.tensor(CITYSCAPES_WEIGHTS, dtype=torch.float32))
    opt = torch.optim.Adam(m.parameters(), 5e-4, (0.9, 0.999), eps=1e-08, weight_decay=1e-4)
    masks, handles = capture_masks(m)
    torch.manual_seed(7)
    x = S.synthetic_images(2, 64, 128, seed=77)
    t = S.synthetic_labels(2, 8, 16, 20, seed=77)
    y = m(x, only_encode=True)
    opt.zero_grad()
    loss = crit(y, t[:, 0])
    loss.backward()
    out["loss"] = np.array(loss.item(), dtype=np.float32)
    out["masks__keys"] = np.array(sorted(masks.keys()))
    for k in sorted(masks.keys()):
        out["mask_" + k] = masks[k].numpy().copy()
    out["param_names"] = np.array([k for k, _ in m.named_parameters()])
    out["gradnorm"] = np.array([(-1.0 if p.grad is None else p.grad.double().norm().item()) for _, p in m.named_parameters()])
    for k, p in m.named_parameters():
        if p.grad is not None and ("encoder.output_conv" in k or "initial_block" in k or k.endswith("layers.7.conv3x1_1.weight")):
            pack(out, "grad_" + k, p.grad)
    before = [p.detach().clone() for _, p in m.named_parameters()]
    opt.step()
    out["paramsum"] = np.array([p.detach().double().sum().item() for _, p in m.named_parameters()])
    out["paramabs"] = np.array([p.detach().double().abs().sum().item() for _, p in m.named_parameters()])
    out["moved"] = np.array([float((p.detach() - b).abs().max().item()) for (_, p), b in zip(m.named_parameters(), before)])
    pack(out, "logits", y)
    for h in handles:
        h.remove()
    np.savez_compressed(os.path.join(HERE, "g7_encoder_only.npz"), **out)


# ----------------------------------------------------------------------------- G8
def g8_multi_models():
    """train/multi_models.py Multi_models(14).eval(): the three outputs, the encoder's skips, the state_dict key list."""
    out = {}
    m = S.load_synthetic(ref_multi.Multi_models(14)).eval()
    out["keys"] = np.array(list(m.state_dict().keys()))
    out["shapes"] = np.array([",".join(str(d) for d in v.shape) for v in m.state_dict().values()])
    for tag, (n, h, w, seed) in {"a": (2, 64, 128, 81), "b": (1, 128, 256, 82)}.items():
        x = S.synthetic_images(n, h, w, seed=seed)
        with torch.no_grad():
            trav, depth, cls = m(x)
            feat, s1, s2 = m.encoder(x)
        out[tag + "__in"] = np.array([n, h, w, seed], dtype=np.int64)
        pack(out, tag + "_trav", trav)
        pack(out, tag + "_depth", depth)
        pack(out, tag + "_class", cls)
        pack(out, tag + "_feat", feat)
        pack(out, tag + "_skip1", s1)
        pack(out, tag + "_skip2", s2)
    np.savez_compressed(os.path.join(HERE, "g8_multi_models.npz"), **out)


# ----------------------------------------------------------------------------- G9
def _oracle_margin(kind, prefix, sd, x, args, masks, storage_dtype):
    """Smallest relative distance of a ReLU / max-pool decision from its boundary, from the oracle's evaluation of the
    same block (same torch ops in the same order as the reference module).  Exact zeros / exact ties are reproducible
    decisions (zero inputs, dropped channels), not fragile ones, and are skipped by the recorder's caller below."""
    from oracle import erfnet_oracle as O
    st = O.RunState(training=True, dropout_masks=masks, storage_dtype=storage_dtype, record_gates=True)
    xx = x.to(storage_dtype).float() if (storage_dtype is not None and x.shape[1] != 3) else x
    with torch.no_grad():
        if kind == "nb":
            y = O.nb1d(sd, prefix, xx, args[1], args[2], st)
        elif kind == "down":
            y = O.downsampler(sd, prefix, xx, st)
        else:
            y = O.upsampler(sd, prefix, xx, st)
    return min(m for _, m in st.gate_margins), y


def g9_gate_safe_blocks():
    """Block fixtures whose every ReLU gate and max-pool route is DECISIVE: the search below walks seeds until no
    pre-activation of the reference lies within MARGIN (relative to its tensor's rms) of a decision boundary, both in the
    reference's own fp32 arithmetic and in the same graph evaluated with bf16 storage.  Two correct implementations then
    take identical decisions, and gradients can be compared element by element in max-norm with no outlier allowance.
    Tensors are stored in full."""
    global MAX_FULL
    from oracle import erfnet_oracle as O
    keep, MAX_FULL = MAX_FULL, 1 << 30
    MARGIN = 2e-5
    out = {"margin_required": np.array(MARGIN)}
    cases = [("nb", (64, 1, 0.03), (2, 64, 8, 24)), ("nb", (16, 1, 0.0), (1, 16, 16, 32)), ("nb", (128, 2, 0.3), (2, 128, 8, 16)),
             ("nb", (128, 4, 0.3), (1, 128, 16, 16)), ("nb", (128, 8, 0.3), (1, 128, 24, 16)), ("nb", (128, 16, 0.3), (1, 128, 40, 8)),
             ("down", (16, 64), (1, 16, 16, 32)), ("down", (64, 128), (1, 64, 16, 16)), ("down", (3, 16), (1, 3, 16, 32)),
             ("up", (128, 64), (1, 128, 8, 8)), ("up", (64, 16), (1, 64, 8, 16))]
    for kind, args, shape in cases:
        tag = f"{kind}_" + "_".join(str(int(a * 100) if isinstance(a, float) else a) for a in args)
        found = None
        for seed in range(1000, 3000):
            prefix = f"safe{seed}" if kind != "up" else f"decoder.layers.{seed}"
            if kind == "nb":
                mod = ref_erfnet.non_bottleneck_1d(args[0], args[2], args[1])
            elif kind == "down":
                mod = ref_erfnet.DownsamplerBlock(*args)
            else:
                mod = ref_erfnet.UpsamplerBlock(*args)
            sd = S.fill_state_dict({prefix + "." + k: v for k, v in mod.state_dict().items()})
            x = S.synthetic_images(shape[0], shape[2], shape[3], seed=seed) if shape[1] == 3 else activations(shape, seed)
            masks = {}
            if kind == "nb" and args[2] > 0:
                torch.manual_seed(1234 + seed)
                keepm = (torch.rand(shape[0], shape[1], 1, 1) >= args[2]).float()
                masks = {prefix: keepm / (1.0 - args[2])}
            m32, _ = _oracle_margin(kind, prefix, sd, x, args, masks, None)
            if m32 < MARGIN:
                continue
            m16, _ = _oracle_margin(kind, prefix, sd, x, args, masks, torch.bfloat16)
            if m16 < MARGIN:
                continue
            found = (seed, prefix, mod, sd, x, masks, m32, m16)
            break
        assert found is not None, tag
        seed, prefix, mod, sd, x, masks, m32, m16 = found
        print(tag, "seed", seed, "margins", m32, m16)
        mod.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()})
        mod.train()
        if masks:        # make the reference's Dropout2d apply exactly this mask
            mk = masks[prefix]
            mod.dropout.forward = lambda t, mk=mk: t * mk
        xr = x.clone().requires_grad_(True)
        y = mod(xr)
        dy = upstream(y.shape, seed)
        y.backward(dy)
        out[tag + "__prefix"] = np.array(prefix)
        out[tag + "__xshape"] = np.array(x.shape, dtype=np.int64)
        out[tag + "__seed"] = np.array(seed, dtype=np.int64)
        out[tag + "__margins"] = np.array([m32, m16])
        pack(out, tag + "_y_train", y)
        pack(out, tag + "_dx", xr.grad)
        for k, v in mod.named_parameters():
            pack(out, tag + "_grad_" + k, v.grad)
        for k, v in mod.named_buffers():
            if "num_batches" not in k:
                pack(out, tag + "_buf_" + k, v)
        if masks:
            out[tag + "_mask"] = masks[prefix].numpy()
    MAX_FULL = keep
    np.savez_compressed(os.path.join(HERE, "g9_gate_safe_blocks.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:                       # e.g. `make_golden.py g7_encoder_only`: regenerate the named sets only
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g1_full_eval()
    g2_blocks()
    g3_loss()
    g4_iou()
    g5_train_steps()
    g6_imagenet()
    g7_encoder_only()
    g8_multi_models()
    g9_gate_safe_blocks()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
