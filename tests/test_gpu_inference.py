"""GPU parity of the inference path (module API -> ctypes -> C ABI -> gfx950 kernels) against
the CPU oracle and the golden vectors produced by the reference.

Tolerances (relative to the reference tensor's max magnitude):
  fp32 kernels: 1e-4 (north-star bar); the f32 MFMA is an exact fp32 fma chain, observed ~1e-6
  bf16 kernels: 3e-2 per layer / 6e-2 end to end (8-bit mantissa activations and weights)
Label maps: exact wherever the reference's own top-2 margin exceeds the logit tolerance.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from erfnet_pytorch_amd import synthetic as S
from oracle import erfnet_oracle as O
from tests.golden_util import load, check_packed, activations, rel_err

pytestmark = pytest.mark.gpu
FP32_TOL = 1e-4
BF16_TOL = 3e-2


def _dev():
    return torch.device("cuda:0")


def _nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(_dev(), dtype)


def _back(y):
    return y.float().cpu().permute(0, 3, 1, 2)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, FP32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("c,shape", [(16, (2, 16, 12, 40)), (64, (2, 64, 20, 24)), (128, (1, 128, 40, 24)),
                                     (128, (4, 128, 64, 128))])     # the last: enough tiles for the 128-pixel tiling
@pytest.mark.parametrize("axis,dil", [(0, 1), (1, 1), (0, 2), (1, 4), (0, 16), (1, 16), (0, 8)])
def test_conv1d_factorized(dtype, tol, c, shape, axis, dil):
    from erfnet_pytorch_amd import ops
    seed = 100 + c + axis * 7 + dil
    x = activations(shape, seed) - 0.3
    kshape = (c, c, 3, 1) if axis == 0 else (c, c, 1, 3)
    w = S.fill_tensor("t.conv.weight", kshape, salt=seed)
    b = S.fill_tensor("t.conv.bias", (c,), salt=seed)
    sc = S.fill_tensor("t.bn.weight", (c,), salt=seed)
    res = activations(shape, seed + 1)
    pad, dl = ((dil, 0), (dil, 1)) if axis == 0 else ((0, dil), (1, dil))
    ref = F.relu(F.conv2d(x, w, None, padding=pad, dilation=dl) * sc[None, :, None, None] + b[None, :, None, None] + res)
    packed = ops.pack_conv1d(w.to(_dev()), ops.dt(torch.empty(0, dtype=dtype)))
    y = ops.conv1d(_nhwc(x, dtype), packed, sc.to(_dev()), b.to(_dev()), _nhwc(res, dtype), axis, dil, True)
    torch.cuda.synchronize()
    assert rel_err(_back(y), ref) <= tol
    # no scale, no residual, no relu
    ref2 = F.conv2d(x, w, b, padding=pad, dilation=dl)
    y2 = ops.conv1d(_nhwc(x, dtype), packed, None, b.to(_dev()), None, axis, dil, False)
    assert rel_err(_back(y2), ref2) <= tol


@pytest.mark.parametrize("c,shape,dil", [
    (128, (2, 128, 64, 128), 1), (128, (1, 128, 64, 128), 2), (128, (1, 128, 64, 128), 4), (128, (2, 128, 64, 128), 8),
    (128, (1, 128, 37, 53), 2), (128, (1, 128, 23, 70), 1), (128, (1, 128, 45, 62), 4),
    # dilation 16 on maps of at most 64 x 128: the whole-row kernel (4 or 3 rows per residue class, ragged widths; at least
    # 49 rows and columns, below that the single launches fall back to the generic kernel with its own summation order)
    (128, (2, 128, 64, 128), 16), (128, (1, 128, 50, 100), 16), (128, (3, 128, 49, 53), 16),
    (128, (20, 128, 64, 128), 16),       # more classes than workgroups
    (64, (1, 64, 128, 256), 1), (64, (2, 64, 19, 41), 1), (64, (1, 64, 33, 30), 2),
    (16, (1, 16, 256, 512), 1), (16, (2, 16, 50, 70), 1), (16, (1, 16, 61, 30), 1),
])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv1d_pair_equals_two_launches_and_reference(c, shape, dil, with_res):
    """Fused conv3x1 -> ReLU -> conv1x3 launch (bf16): against the two single launches it replaces (same
    bf16 rounding points, so within one bf16 ulp) and against the fp32 torch reference of the same ops."""
    from erfnet_pytorch_amd import ops
    seed = 300 + c + dil
    x = activations(shape, seed) - 0.3
    wa = S.fill_tensor("p.a.weight", (c, c, 3, 1), salt=seed)
    wb = S.fill_tensor("p.b.weight", (c, c, 1, 3), salt=seed)
    ba = S.fill_tensor("p.a.bias", (c,), salt=seed)
    bb = S.fill_tensor("p.b.bias", (c,), salt=seed)
    sb = S.fill_tensor("p.bn.weight", (c,), salt=seed)
    res = activations(shape, seed + 1) if with_res else None
    mid = F.relu(F.conv2d(x, wa, ba, padding=(dil, 0), dilation=(dil, 1)))
    ref = F.conv2d(mid, wb, None, padding=(0, dil), dilation=(1, dil)) * sb[None, :, None, None] + bb[None, :, None, None]
    ref = F.relu(ref + res) if with_res else F.relu(ref)
    code = ops.dt(torch.empty(0, dtype=torch.bfloat16))
    pa, pb = ops.pack_conv1d(wa.to(_dev()), code), ops.pack_conv1d(wb.to(_dev()), code)
    xd = _nhwc(x, torch.bfloat16)
    rd = _nhwc(res, torch.bfloat16) if with_res else None
    y = ops.conv1d_pair(xd, pa, ba.to(_dev()), pb, sb.to(_dev()), bb.to(_dev()), rd, dil, True)
    assert y is not None, "shape should be taken by the fused kernel"
    m = ops.conv1d(xd, pa, None, ba.to(_dev()), None, 0, dil, True)
    y2 = ops.conv1d(m, pb, sb.to(_dev()), bb.to(_dev()), rd, 1, dil, True)
    torch.cuda.synchronize()
    assert rel_err(_back(y), ref) <= BF16_TOL
    diff = (y.float() - y2.float()).abs()
    ulp = y2.float().abs().clamp_min(1e-2) * 2.0 ** -7
    assert bool((diff <= ulp).all()), f"max diff {diff.max().item()} at {(diff / ulp).max().item()} ulp"


@pytest.mark.parametrize("shape", [(3, 128, 20, 37), (1, 128, 9, 16), (2, 128, 16, 128), (1, 128, 17, 5), (3, 128, 40, 37),
                                   (1, 128, 33, 16), (2, 128, 64, 90)])
@pytest.mark.parametrize("with_res", [False, True])
def test_conv1d_pair_whole_rows_small_maps(shape, with_res):
    """Dilation 16 on small maps: one to four rows per residue class, widths that leave matrix tiles partly or wholly
    empty.  Below 49 rows or columns the single launches fall back to the generic kernel (another summation order), so
    this compares with the fp32 torch reference only."""
    from erfnet_pytorch_amd import ops
    c, dil, seed = 128, 16, 77 + shape[2]
    x = activations(shape, seed) - 0.3
    wa = S.fill_tensor("p.a.weight", (c, c, 3, 1), salt=seed)
    wb = S.fill_tensor("p.b.weight", (c, c, 1, 3), salt=seed)
    ba = S.fill_tensor("p.a.bias", (c,), salt=seed)
    bb = S.fill_tensor("p.b.bias", (c,), salt=seed)
    sb = S.fill_tensor("p.bn.weight", (c,), salt=seed)
    res = activations(shape, seed + 1) if with_res else None
    mid = F.relu(F.conv2d(x, wa, ba, padding=(dil, 0), dilation=(dil, 1)))
    ref = F.conv2d(mid, wb, None, padding=(0, dil), dilation=(1, dil)) * sb[None, :, None, None] + bb[None, :, None, None]
    ref = F.relu(ref + res) if with_res else F.relu(ref)
    code = ops.dt(torch.empty(0, dtype=torch.bfloat16))
    pa, pb = ops.pack_conv1d(wa.to(_dev()), code), ops.pack_conv1d(wb.to(_dev()), code)
    rd = _nhwc(res, torch.bfloat16) if with_res else None
    y = ops.conv1d_pair(_nhwc(x, torch.bfloat16), pa, ba.to(_dev()), pb, sb.to(_dev()), bb.to(_dev()), rd, dil, True)
    assert y is not None, "shape should be taken by the whole-row kernel"
    torch.cuda.synchronize()
    assert rel_err(_back(y), ref) <= BF16_TOL


def test_conv1d_pair_declines_sparse_lattices_and_fp32():
    from erfnet_pytorch_amd import ops
    code = ops.dt(torch.empty(0, dtype=torch.bfloat16))
    w = S.fill_tensor("p.a.weight", (128, 128, 3, 1), salt=1).to(_dev())
    pk = ops.pack_conv1d(w, code)
    b = torch.zeros(128, device=_dev())
    x = torch.zeros(1, 64, 200, 128, device=_dev(), dtype=torch.bfloat16)
    assert ops.conv1d_pair(x, pk, b, pk, None, b, None, 16, True) is None            # 4 x 13 points per residue class, rows wider than 128
    x = torch.zeros(1, 30, 60, 128, device=_dev(), dtype=torch.bfloat16)
    assert ops.conv1d_pair(x, pk, b, pk, None, b, None, 8, True) is None             # 4 x 8 points, but not dilation 16
    x = torch.zeros(1, 64, 128, 128, device=_dev(), dtype=torch.bfloat16)
    pk32 = ops.pack_conv1d(w, ops.dt(torch.empty(0)))
    assert ops.conv1d_pair(x.float(), pk32, b, pk32, None, b, None, 1, True) is None


def _block_tags():
    g = load("g2_blocks.npz")
    return sorted({k.split("__prefix")[0] for k in g.files if k.endswith("__prefix")})


@pytest.mark.parametrize("dtype,tol", [(torch.float32, FP32_TOL), (torch.bfloat16, BF16_TOL)])
@pytest.mark.parametrize("tag", _block_tags())
def test_block_eval_vs_golden(tag, dtype, tol):
    from erfnet_pytorch_amd import erfnet as E
    g = load("g2_blocks.npz")
    prefix = str(g[tag + "__prefix"])
    shape = tuple(int(v) for v in g[tag + "__xshape"])
    seed = int(g[tag + "__seed"])
    kind = tag.split("_")[0]
    if kind == "nb":
        c, d, p = int(tag.split("_")[1][1:]), int(tag.split("_")[2][1:]), int(tag.split("_")[3][1:]) / 100.0
        mod, tmpl, x = E.non_bottleneck_1d(c, p, d), O.nb_template(prefix, c), activations(shape, seed)
    elif kind == "down":
        cin, cout = int(tag.split("_")[1]), int(tag.split("_")[2])
        mod, tmpl = E.DownsamplerBlock(cin, cout), O.down_template(prefix, cin, cout)
        x = S.synthetic_images(shape[0], shape[2], shape[3], seed=seed) if cin == 3 else activations(shape, seed)
    else:
        cin, cout = int(tag.split("_")[1]), int(tag.split("_")[2])
        mod, tmpl, x = E.UpsamplerBlock(cin, cout), O.up_template(prefix, cin, cout), activations(shape, seed)
    sd = O.synthetic_state(tmpl)
    mod.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()})
    mod = mod.to(_dev()).eval()
    with torch.no_grad():
        y = mod(x.to(_dev(), dtype))
    assert y.shape == tuple(int(v) for v in g[tag + "_y_eval__shape"])
    check_packed(g, tag + "_y_eval", y.float().cpu(), tol)


@pytest.mark.parametrize("nc", [20, 6])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, FP32_TOL), (torch.bfloat16, BF16_TOL)])
def test_output_conv_logits_and_fused_argmax(nc, dtype, tol):
    from erfnet_pytorch_amd import ops
    g = load("g2_blocks.npz")
    tag = f"outconv_{nc}"
    seed = int(g[tag + "__seed"])
    w = S.fill_tensor("decoder.output_conv.weight", (16, nc, 2, 2), salt=seed)
    b = S.fill_tensor("decoder.output_conv.bias", (nc,), salt=seed)
    x = activations((2, 16, 12, 16), seed)
    code = ops.dt(torch.empty(0, dtype=dtype))
    packed = ops.pack_output_conv(w.to(_dev()), code)
    logits, labels = ops.output_conv(_nhwc(x, dtype), packed, b.to(_dev()), nc, True, True, True)
    check_packed(g, tag + "_y", logits.cpu(), tol)
    # fused labels == argmax of the logits the same kernel wrote (bit-exact, first max wins)
    assert torch.equal(labels.cpu(), logits.cpu().max(1)[1])
    only_labels = ops.output_conv(_nhwc(x, dtype), packed, b.to(_dev()), nc, False, True, True, torch.uint8)[1]
    assert torch.equal(only_labels.cpu().long(), labels.cpu())
    assert torch.equal(ops.argmax_c(logits).cpu(), labels.cpu())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nc", [1, 2, 3, 4, 5, 7, 8, 9, 13, 16, 19, 21, 29, 32])
def test_output_conv_every_class_count(nc, dtype):
    """Decoder.output_conv for class counts on both sides of every 4-class channel tile (the labels-only kernel masks the classes
    beyond nc of the last tile by a select, the logits kernels by folded bounds): logits against torch's ConvTranspose2d, and the
    labels of the labels-only launch == arg-max of the logits launch, bit for bit; odd spatial sizes."""
    from erfnet_pytorch_amd import ops
    gen = torch.Generator().manual_seed(900 + nc)
    w = torch.randn(16, nc, 2, 2, generator=gen) * 0.3
    b = torch.randn(nc, generator=gen) * 0.2
    x = torch.relu(torch.randn(2, 16, 7, 10, generator=gen))
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv_transpose2d(x, w, b, stride=2)
    code = ops.dt(torch.empty(0, dtype=dtype))
    packed = ops.pack_output_conv(w.to(_dev()), code)
    logits, labels = ops.output_conv(_nhwc(x, dtype), packed, b.to(_dev()), nc, True, True, True)
    assert rel_err(logits.cpu(), ref) <= (1e-5 if dtype == torch.float32 else 1e-5)      # same rounded operands, fp32 accumulation
    assert torch.equal(labels.cpu(), logits.cpu().max(1)[1])
    only = ops.output_conv(_nhwc(x, dtype), packed, b.to(_dev()), nc, False, True, True, torch.uint8)[1]
    assert torch.equal(only.cpu().long(), labels.cpu())
    only64 = ops.output_conv(_nhwc(x, dtype), packed, b.to(_dev()), nc, False, True, True)[1]
    assert torch.equal(only64.cpu(), labels.cpu())


def test_argmax_ties_first_index_wins():
    from erfnet_pytorch_amd import ops
    z = torch.zeros(1, 5, 4, 8)
    z[0, 3, 1, :] = 2.0
    z[0, 1, 1, :] = 2.0
    z[0, 4, 2, 3] = -1.0
    lab = ops.argmax_c(z.to(_dev())).cpu()
    assert torch.equal(lab, z.max(1)[1])


@pytest.mark.parametrize("nc,tags", [(20, ("a", "b")), (6, ("a",))])
def test_full_net_fp32_vs_golden(nc, tags):
    from erfnet_pytorch_amd import erfnet as E
    g = load("g1_full_eval.npz")
    model = E.ERFNet(nc)
    model.load_state_dict(O.synthetic_state(O.erfnet_template(nc)))
    model = model.to(_dev()).eval()
    for tag in tags:
        n, h, w, seed = (int(v) for v in g[f"nc{nc}_{tag}__in"])
        x = S.synthetic_images(n, h, w, seed=seed).to(_dev())
        with torch.no_grad():
            y = model(x)
            labels = model.predict_labels(x)
        assert y.is_contiguous() and y.dtype == torch.float32
        err = check_packed(g, f"nc{nc}_{tag}_logits", y.cpu(), FP32_TOL)
        gold = torch.from_numpy(g[f"nc{nc}_{tag}_argmax"].astype(np.int64))
        margin = torch.from_numpy(g[f"nc{nc}_{tag}_margin"])
        lab = labels.cpu()
        assert torch.equal(lab, y.cpu().max(1)[1])                 # fused arg-max == arg-max of our logits
        amax = float(np.abs(g[f"nc{nc}_{tag}_logits__sample"]).max())
        decided = margin > 2 * FP32_TOL * amax                        # reference's own top-2 gap above tolerance
        assert torch.equal(lab[decided], gold[decided]), f"labels differ where margin is decisive (logit err {err:.2e})"
        assert (lab != gold).float().mean().item() < 1e-3
    if nc == 20:
        x = S.synthetic_images(1, 64, 128, seed=1).to(_dev())
        with torch.no_grad():
            check_packed(g, "nc20_a_only_encode", model(x, only_encode=True).cpu(), FP32_TOL)
            check_packed(g, "nc20_a_encoder_feat", model.encoder(x).float().cpu(), FP32_TOL)


BF16_LOGIT_TOL = 6e-2        # stated bf16 tolerance on logits, relative to the largest logit magnitude of the reference


@pytest.mark.parametrize("nc,tags", [(20, ("a", "b")), (6, ("a",))])
def test_full_net_bf16_vs_golden(nc, tags):
    """bf16 activations / weights with fp32 accumulation against the REFERENCE's fp32 outputs: logits within the stated bf16
    tolerance, and labels EXACT wherever the reference's own top-2 margin exceeds twice that tolerance (a smaller margin cannot
    be decided by any bf16 evaluation) -- the same rule the fp32 test applies with its 1e-4."""
    from erfnet_pytorch_amd import erfnet as E
    g = load("g1_full_eval.npz")
    model = E.ERFNet(nc)
    model.load_state_dict(O.synthetic_state(O.erfnet_template(nc)))
    model = model.to(_dev()).eval()
    model.compute_dtype = torch.bfloat16
    for tag in tags:
        n, h, w, seed = (int(v) for v in g[f"nc{nc}_{tag}__in"])
        x = S.synthetic_images(n, h, w, seed=seed).to(_dev())
        with torch.no_grad():
            y = model(x)
            labels = model.predict_labels(x)
        assert y.dtype == torch.float32
        check_packed(g, f"nc{nc}_{tag}_logits", y.cpu(), BF16_LOGIT_TOL)
        lab = labels.cpu()
        assert torch.equal(lab, y.cpu().max(1)[1])                 # fused arg-max == arg-max of our logits
        gold = torch.from_numpy(g[f"nc{nc}_{tag}_argmax"].astype(np.int64))
        margin = torch.from_numpy(g[f"nc{nc}_{tag}_margin"])
        amax = float(np.abs(g[f"nc{nc}_{tag}_logits__sample"]).max())
        decided = margin > 2 * BF16_LOGIT_TOL * amax
        assert decided.float().mean().item() > 0.1                 # the rule binds a real share of the pixels (19 % .. 40 %)
        assert torch.equal(lab[decided], gold[decided]), "bf16 labels differ where the reference margin is decisive"
        assert (lab == gold).float().mean().item() > 0.90          # overall agreement on these random-weight logits, for the record


@pytest.mark.parametrize("nc,shape", [(20, (16, 512, 1024)), (6, (8, 1024, 2048))])
def test_full_size_properties(nc, shape):
    """BASELINE configs[1] (batch 16, 3x512x1024, 20 classes) and configs[4] (batch 8, 3x1024x2048, 6 classes) at their
    full sizes, through properties
    that do not depend on the size: fused labels == arg-max of the logits path bit for bit, images of a batch are
    independent (same result alone and in a batch), the confusion matrix counts every pixel once and equals the
    histogram of (label, prediction) pairs computed by torch."""
    from erfnet_pytorch_amd import erfnet as E
    from erfnet_pytorch_amd.iouEval import iouEval
    n, h, w = shape
    model = E.ERFNet(nc)
    model.load_state_dict(O.synthetic_state(O.erfnet_template(nc)))
    model = model.to(_dev()).eval()
    model.compute_dtype = torch.bfloat16
    x = S.synthetic_images(n, h, w, seed=11).to(_dev())
    with torch.no_grad():
        logits = model(x)
        labels = model.predict_labels(x, torch.uint8)
        alone = model.predict_labels(x[n - 1:].contiguous(), torch.uint8)
    assert logits.shape == (n, nc, h, w) and labels.shape == (n, h, w)
    assert torch.equal(labels.long(), logits.max(1)[1])
    assert torch.equal(alone[0], labels[n - 1])
    del logits
    target = S.synthetic_labels(n, h, w, nc, seed=12).to(_dev())
    ev = iouEval(nc)
    ev.addBatch(labels.unsqueeze(1), target)
    conf = ev.confusion()
    assert int(conf.sum()) == n * h * w
    ref = torch.bincount((target[:, 0].long() * nc + labels.long()).reshape(-1), minlength=nc * nc).reshape(nc, nc).cpu()
    assert torch.equal(conf.reshape(nc, nc).cpu().long(), ref)


def _oracle_labels_and_margin(ref):
    top2 = ref.topk(2, dim=1).values
    return ref.max(1)[1], top2[:, 0] - top2[:, 1], float(ref.abs().max())


@pytest.mark.parametrize("nc,batch,h,w,picks", [(20, 16, 512, 1024, (0, 15)), (6, 8, 1024, 2048, (7,))])
def test_full_size_against_the_oracle(nc, batch, h, w, picks):
    """BASELINE configs[1] / configs[4] at their full sizes against the CPU oracle (eval/eval_iou.py:88-93: forward, arg-max,
    iouEval.addBatch), end to end through the kernels and tilings the benchmark runs: the whole-row d = 16 kernel, the
    producer/consumer pair kernel with several tiles per workgroup, the 128-pixel tilings, the one-launch 16-channel blocks.
    fp32 (a batch of the picked images): logits within 1e-4 of the largest logit, labels exact wherever the oracle's top-2
    margin exceeds twice that, fewer than 1e-3 of the pixels differ.  bf16 (the full benchmark batch, picked images compared):
    logits within the stated 6e-2, labels exact where the margin exceeds twice that.  iouEval counts on OUR labels equal the
    oracle's counting of the same labels exactly, through evaluate_batch with one and with two chains of launches."""
    from erfnet_pytorch_amd import erfnet as E
    from erfnet_pytorch_amd.iouEval import iouEval
    sd = O.synthetic_state(O.erfnet_template(nc))
    model = E.ERFNet(nc)
    model.load_state_dict(sd)
    model = model.to(_dev()).eval()
    x = S.synthetic_images(batch, h, w, seed=31)
    target = S.synthetic_labels(batch, h, w, nc, seed=32)
    idx = list(picks)
    with torch.no_grad():
        ref = O.erfnet(sd, x[idx])
    gold, margin, amax = _oracle_labels_and_margin(ref)

    # fp32, the picked images as one batch
    xs = x[idx].contiguous().to(_dev())
    with torch.no_grad():
        y = model(xs).cpu()
        lab = model.predict_labels(xs).cpu()
    assert (y - ref).abs().max().item() <= FP32_TOL * amax
    assert torch.equal(lab, y.max(1)[1])
    decided = margin > 2 * FP32_TOL * amax
    assert torch.equal(lab[decided], gold[decided])
    assert (lab != gold).float().mean().item() < 1e-3
    ev = iouEval(nc)
    ev.addBatch(lab.unsqueeze(1).to(_dev()), target[idx].to(_dev()))
    tp, fp, fn = O.iou_counts(lab.unsqueeze(1), target[idx], nc)
    assert torch.equal(ev.tp, tp) and torch.equal(ev.fp, fp) and torch.equal(ev.fn, fn)
    del y

    # bf16, the whole benchmark batch
    model.compute_dtype = torch.bfloat16
    xd, td = x.to(_dev()), target.to(_dev())
    ev1, ev2 = iouEval(nc), iouEval(nc)
    with torch.no_grad():
        yb = model(xd[idx].contiguous()).cpu()
        l1 = model.evaluate_batch(xd, td, ev1, torch.uint8)
        l2 = model.evaluate_batch(xd, td, ev2, torch.uint8, ways=2)
    assert torch.equal(l1, l2) and torch.equal(ev1.confusion(), ev2.confusion())
    assert (yb - ref).abs().max().item() <= BF16_LOGIT_TOL * amax
    labb = l1.cpu()[idx].long()
    assert torch.equal(labb, yb.max(1)[1])                         # images are independent: batch of 16 == batch of the picks
    decided = margin > 2 * BF16_LOGIT_TOL * amax
    assert decided.float().mean().item() > 0.05
    assert torch.equal(labb[decided], gold[decided])
    assert (labb == gold).float().mean().item() > 0.90
    tp, fp, fn = O.iou_counts(l1.cpu().long().unsqueeze(1), target, nc)
    for e in (ev1, ev2):
        assert torch.equal(e.tp, tp) and torch.equal(e.fp, fp) and torch.equal(e.fn, fn)


def test_iou_eval_exact_vs_golden_and_oracle():
    from erfnet_pytorch_amd.iouEval import iouEval
    g = load("g4_iou.npz")
    for nc in (20, 6, 4):
        ev = iouEval(nc)
        for b in range(3):
            seed = 40 + b
            pred = S.synthetic_labels(2, 32, 48, nc, seed=seed + 100, ignore_frac=0.0)
            lab = S.synthetic_labels(2, 32, 48, nc, seed=seed, ignore_frac=0.07 if nc == 20 else 0.0)
            if nc == 20:
                pred = torch.where(S.synthetic_labels(2, 32, 48, 50, seed=seed + 200, ignore_frac=0) == 0,
                                   torch.full_like(pred, 19), pred)
            ev.addBatch(pred.to(_dev()), lab.to(_dev()))
            assert np.array_equal(ev.tp.numpy(), g[f"nc{nc}_b{b}_tp"])
            assert np.array_equal(ev.fp.numpy(), g[f"nc{nc}_b{b}_fp"])
            assert np.array_equal(ev.fn.numpy(), g[f"nc{nc}_b{b}_fn"])
        mean, per = ev.getIoU()
        assert np.array_equal(per.numpy(), g[f"nc{nc}_iou"])
        assert mean.item() == float(g[f"nc{nc}_iou_mean"])
    # uint8 predictions straight from the fused arg-max, larger map, against the oracle
    pred = S.synthetic_labels(3, 128, 256, 20, seed=9, ignore_frac=0)
    lab = S.synthetic_labels(3, 128, 256, 20, seed=10)
    ev = iouEval(20)
    ev.addBatch(pred.to(torch.uint8).to(_dev())[:, 0], lab.to(_dev()))
    tp, fp, fn = O.iou_counts(pred, lab, 20)
    assert torch.equal(ev.tp, tp) and torch.equal(ev.fp, fp) and torch.equal(ev.fn, fn)
    assert int(ev.confusion().sum()) == pred.numel()


@pytest.mark.parametrize("cin,cout,shape", [
    (64, 128, (2, 64, 26, 44)), (64, 128, (1, 64, 8, 32)), (64, 128, (1, 64, 128, 256)),
    (16, 64, (2, 16, 38, 52)), (16, 64, (1, 16, 16, 20)), (16, 64, (1, 16, 256, 512)),
])
def test_downsampler_one_launch_kernel_edges(cin, cout, shape):
    """bf16 DownsamplerBlock kernel on maps whose halved size is no multiple of the tile (R rows x 16 columns):
    zero padding at the top / left, clipped tiles at the bottom / right, pooled half, affine + ReLU."""
    from erfnet_pytorch_amd import ops
    seed = 500 + cin
    x = (activations(shape, seed) - 0.4).bfloat16().float()
    w = S.fill_tensor("d.conv.weight", (cout - cin, cin, 3, 3), salt=seed).bfloat16().float()
    sc = S.fill_tensor("d.bn.weight", (cout,), salt=seed)
    sh = S.fill_tensor("d.bn.bias", (cout,), salt=seed)
    ref = torch.cat([F.conv2d(x, w, None, stride=2, padding=1), F.max_pool2d(x, 2, stride=2)], 1)
    ref = F.relu(ref * sc[None, :, None, None] + sh[None, :, None, None])
    code = ops.dt(torch.empty(0, dtype=torch.bfloat16))
    y = ops.downsampler(_nhwc(x, torch.bfloat16), ops.pack_downsampler(w.to(_dev()), code), sc.to(_dev()), sh.to(_dev()), cout)
    torch.cuda.synchronize()
    assert rel_err(_back(y), ref) <= 1e-2            # bf16 output rounding only: inputs and weights are bf16-exact


@pytest.mark.parametrize("cin,cout,shape", [
    (128, 64, (2, 128, 13, 22)), (128, 64, (1, 128, 4, 16)), (128, 64, (1, 128, 64, 128)),
    (64, 16, (2, 64, 19, 26)), (64, 16, (1, 64, 8, 16)), (64, 16, (1, 64, 128, 256)),
])
def test_upsampler_one_launch_kernel_edges(cin, cout, shape):
    """bf16 UpsamplerBlock kernel: parity phases, halo row / column beyond the image (zero contribution), clipped tiles."""
    from erfnet_pytorch_amd import ops
    seed = 600 + cin
    x = (activations(shape, seed) - 0.4).bfloat16().float()
    w = S.fill_tensor("u.conv.weight", (cin, cout, 3, 3), salt=seed).bfloat16().float()
    sc = S.fill_tensor("u.bn.weight", (cout,), salt=seed)
    sh = S.fill_tensor("u.bn.bias", (cout,), salt=seed)
    ref = F.conv_transpose2d(x, w, None, stride=2, padding=1, output_padding=1)
    ref = F.relu(ref * sc[None, :, None, None] + sh[None, :, None, None])
    code = ops.dt(torch.empty(0, dtype=torch.bfloat16))
    y = ops.upsampler(_nhwc(x, torch.bfloat16), ops.pack_upsampler(w.to(_dev()), code), sc.to(_dev()), sh.to(_dev()), cout)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (shape[0], 2 * shape[2], 2 * shape[3], cout)
    assert rel_err(_back(y), ref) <= 1e-2


def test_iou_eval_plane_inputs_vs_golden():
    """iouEval.addBatch with one-hot / soft-score planes (train/iouEval.py:37-38,45-46 take them as they are): counts of one-hot
    inputs exact, soft scores to the reference's own fp32 summation error; mixed with a label-map batch in one evaluator."""
    from erfnet_pytorch_amd.iouEval import iouEval
    from tests.test_oracle_golden import _plane_cases
    g = load("g4_iou.npz")
    for nc in (20, 4):
        for tag, (x, y) in _plane_cases(nc).items():
            ev = iouEval(nc)
            ev.addBatch(x.to(_dev()), y.to(_dev()))
            ev.addBatch(x.to(_dev()), y.to(_dev()))
            for k in ("tp", "fp", "fn"):
                ref = g[f"planes_nc{nc}_{tag}_{k}"]
                got = getattr(ev, k).numpy()
                if tag.startswith("onehot"):
                    assert np.array_equal(got, ref), (nc, tag, k)
                else:
                    assert np.allclose(got, ref, rtol=2e-6, atol=1e-6), (nc, tag, k)
        # a label-map batch and its one-hot form add up to twice the label-map counts
        x, y = _plane_cases(nc)["onehot_x"]
        pred = x.argmax(1, keepdim=True)
        ev, ev2 = iouEval(nc), iouEval(nc)
        ev.addBatch(pred.to(_dev()), y.to(_dev()))
        ev.addBatch(x.to(_dev()), y.to(_dev()))
        ev2.addBatch(pred.to(_dev()), y.to(_dev()))
        assert torch.equal(ev.tp, 2 * ev2.tp) and torch.equal(ev.fp, 2 * ev2.fp) and torch.equal(ev.fn, 2 * ev2.fn)
        with pytest.raises(ValueError):
            iouEval(nc).addBatch(x[:, :-1].to(_dev()), y.to(_dev()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, None])
def test_predict_labels_split_over_streams_is_bit_identical(dtype):
    """ERFNet.predict_labels(ways=k): the batch cut into k runs of images on k HIP streams gives the labels of the single
    chain bit for bit (images are independent), launched and replayed from a HIP graph, for even and uneven cuts, and after
    the weights changed (the packed copies are rebuilt before the chains fork)."""
    from erfnet_pytorch_amd import erfnet as E
    model = E.ERFNet(20)
    model.load_state_dict(O.synthetic_state(O.erfnet_template(20)))
    model = model.to(_dev()).eval()
    model.compute_dtype = dtype
    x = S.synthetic_images(5, 64, 128, seed=11).to(_dev())
    one = model.predict_labels(x, torch.uint8)
    for ways in (2, 3, 5, 8):
        assert torch.equal(model.predict_labels(x, torch.uint8, ways=ways), one), ways
    assert torch.equal(model.predict_labels(x, torch.int64, ways=2), one.long())
    static_x = x.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = model.predict_labels(static_x, torch.uint8, ways=2)
    static_x.copy_(S.synthetic_images(5, 64, 128, seed=12).to(_dev()))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, model.predict_labels(static_x, torch.uint8))
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)
    again = model.predict_labels(x, torch.uint8, ways=2)          # first call after the change is a split one
    assert torch.equal(again, model.predict_labels(x, torch.uint8))
    import copy
    twin = copy.deepcopy(model)                                   # nothing unpicklable was hung on the module
    assert torch.equal(twin.predict_labels(x, torch.uint8, ways=2), again)


def test_evaluate_batch_counts_per_chain_equal_the_whole_batch_count():
    """ERFNet.evaluate_batch (eval/eval_iou.py:88-93: labels, then iouEval.addBatch): with two chains every chain counts its own
    images; labels and the confusion matrix equal the one-chain ones exactly, launched and replayed from a HIP graph."""
    from erfnet_pytorch_amd import erfnet as E
    from erfnet_pytorch_amd.iouEval import iouEval
    model = E.ERFNet(20)
    model.load_state_dict(O.synthetic_state(O.erfnet_template(20)))
    model = model.to(_dev()).eval()
    model.compute_dtype = torch.bfloat16
    x = S.synthetic_images(5, 64, 128, seed=21).to(_dev())
    t = S.synthetic_labels(5, 64, 128, 20, seed=21).to(_dev())
    ev1, ev2, ev3 = iouEval(20), iouEval(20), iouEval(20)
    l1 = model.evaluate_batch(x, t, ev1)
    l2 = model.evaluate_batch(x, t, ev2, ways=2)
    assert torch.equal(l1, l2) and torch.equal(ev1.confusion(), ev2.confusion())
    assert int(ev1.confusion().sum()) == 5 * 64 * 128
    ev3.reserve(_dev())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        l3 = model.evaluate_batch(x, t, ev3, ways=2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(l3, l1) and torch.equal(ev3.confusion(), ev1.confusion())

