"""CPU-side checks of the drop-in boundary: the C ABI library exports what include/erfnet_mi355.h
declares, the ctypes table covers it, and the nn.Module mirror has the reference's state_dict
layout and refuses to compute on CPU (there is no fallback path)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "erfnet_mi355.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(erf_[a-zA-Z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from erfnet_pytorch_amd import _lib
    names = _declared()
    assert len(names) >= 50
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in the header but not exported by libermi355.so"
    missing = [n for n in names if n not in _lib.SIGNATURES]
    assert not missing, f"no ctypes signature for {missing}"
    extra = [n for n in _lib.SIGNATURES if n not in names]
    assert not extra, f"ctypes binds undeclared symbols {extra}"
    assert _lib.query("erf_version") >= 100


def test_size_queries_run_without_a_gpu():
    from erfnet_pytorch_amd import _lib
    assert _lib.query("erf_conv1d_packed_bytes", _lib.BF16, 128) == 8 * 12 * 64 * 16
    assert _lib.query("erf_conv1d_packed_bytes", _lib.F32, 128) == 8 * 24 * 64 * 16
    assert _lib.query("erf_output_conv_patch_channels", 20) == 80
    assert _lib.query("erf_output_conv_patch_channels", 6) == 32
    assert _lib.query("erf_bn_workspace_bytes", 128) > 0 and _lib.query("erf_ce_workspace_bytes") > 0
    assert _lib.query("erf_conv1d_packed_bytes", 7, 128) == 0          # bad dtype -> 0, never a crash


def test_state_dict_layout_matches_reference():
    from erfnet_pytorch_amd import erfnet as E
    from oracle import erfnet_oracle as O
    for nc in (20, 6):
        m = E.ERFNet(nc)
        tmpl = O.erfnet_template(nc)                     # verified against the reference by the golden generator
        sd = m.state_dict()
        assert list(sd.keys()) == list(tmpl.keys())
        assert all(tuple(sd[k].shape) == tuple(tmpl[k]) for k in tmpl)
        assert sum(p.numel() for p in m.parameters()) == (2_066_836 if nc == 20 else sum(p.numel() for p in m.parameters()))
    assert E.Net is E.ERFNet
    enc = E.Encoder(20)
    net = E.ERFNet(20, encoder=enc)                       # pretrained-encoder transplant (train/erfnet.py:272-278)
    assert net.encoder is enc
    # DataParallel-style checkpoints ('module.' prefix) load after stripping, like eval/eval_iou.py:54-65
    ckpt = {"module." + k: v for k, v in E.ERFNet(20).state_dict().items()}
    net.load_state_dict({k[len("module."):]: v for k, v in ckpt.items()})
    blk = E.non_bottleneck_1d(64, 0.03, 1)
    assert blk.dropout.p == 0.03 and isinstance(blk.bn1, torch.nn.BatchNorm2d) and blk.bn1.eps == 1e-3


def test_imagenet_variant_state_dict_layout():
    from erfnet_pytorch_amd import erfnet_imagenet as EI
    from oracle import erfnet_oracle as O
    m = EI.ERFNet(1000)
    tmpl = O.imagenet_template(1000)                       # key order verified against the reference by the golden generator
    sd = m.state_dict()
    assert list(sd.keys()) == list(tmpl.keys())
    assert all(tuple(sd[k].shape) == tuple(tmpl[k]) for k in tmpl)
    assert m.features.encoder.layers[1].dropout.p == 0.1 and not hasattr(m.features.encoder, "output_conv")
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(1, 3, 224, 224))


def test_cpu_forward_is_refused():
    from erfnet_pytorch_amd import erfnet as E
    from erfnet_pytorch_amd.iouEval import iouEval
    m = E.ERFNet(20).eval()
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(1, 3, 64, 128))
    with pytest.raises(RuntimeError, match="MI355X"):
        iouEval(20).addBatch(torch.zeros(1, 1, 4, 4, dtype=torch.int64), torch.zeros(1, 1, 4, 4, dtype=torch.int64))


def test_iou_eval_interface():
    from erfnet_pytorch_amd.iouEval import iouEval, getColorEntry, colors
    ev = iouEval(20)
    assert ev.ignoreIndex == 19 and ev.tp.shape == (19,) and ev.tp.dtype == torch.float64
    assert iouEval(4).ignoreIndex == -1 and iouEval(4).fp.shape == (4,)
    mean, per = ev.getIoU()
    assert per.shape == (19,) and float(mean) == 0.0
    assert getColorEntry(0.1) == colors.RED and getColorEntry(0.9) == colors.GREEN and getColorEntry("x") == colors.ENDC


def test_poly_lr_schedule():
    from erfnet_pytorch_amd.trainer import poly_lr
    from oracle import erfnet_oracle as O
    for e in (1, 2, 75, 150):
        assert poly_lr(5e-4, e, 150) == O.poly_lr(5e-4, e, 150)
    assert poly_lr(5e-4, 1, 150) == 5e-4


def test_bucket_ranges_of_the_encoder_only_stage():
    """Encoder stage (train/main.py:203, enc=True): buckets and optimiser range cover encoder + 1x1 head, never the decoder."""
    from erfnet_pytorch_amd import dp
    from oracle import erfnet_oracle as O
    sd = O.synthetic_state(O.erfnet_template(20))
    names = [k for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
    offsets, total = dp.arena_offsets([(k, sd[k].shape) for k in names])
    full_b, full_a = dp.bucket_ranges(offsets)
    enc_b, enc_a = dp.bucket_ranges(offsets, only_encode=True)
    dec_lo = min(offsets[k][0] for k in names if k.startswith("decoder."))
    head = offsets["encoder.output_conv.weight"]
    assert enc_a == [(0, dec_lo)] and all(hi <= dec_lo for _, hi in enc_b)
    assert sorted(enc_b)[0][0] == 0 and sorted(enc_b)[-1][1] == dec_lo and sorted(enc_b)[0][1] == sorted(enc_b)[1][0]
    assert any(lo <= head[0] < hi for lo, hi in enc_b) and not any(lo <= head[0] < hi for lo, hi in full_a)
    assert sum(hi - lo for lo, hi in full_b) + (dec_lo - head[0]) == total


def test_multi_models_state_dict_matches_reference_layout():
    """train/multi_models.py Multi_models(14): same keys, shapes and registration order (checkpoints interchange)."""
    from erfnet_pytorch_amd import multi_models as MM
    from oracle import erfnet_oracle as O
    from tests.golden_util import load
    g = load("g8_multi_models.npz")
    m = MM.Multi_models(14)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [",".join(str(d) for d in v.shape) for v in sd.values()] == [str(s) for s in g["shapes"]]
    m.load_state_dict(O.synthetic_state(O.multi_models_template(14)))
    import pytest
    import torch
    with pytest.raises(RuntimeError):                  # no CPU path: fails loudly
        m.eval()(torch.zeros(1, 3, 64, 128))


def test_tolerant_loader_and_checkpoint_dict_follow_the_reference_conventions(tmp_path):
    """eval/eval_iou.py:54-65 (`module.` prefix accepted, unknown names reported and skipped, partial dictionaries fine) and
    train/main.py:320-326 (checkpoint dictionary with DataParallel-style names, readable by a loader that executes nothing)."""
    import torch
    from erfnet_pytorch_amd import erfnet as E, checkpoint as C, synthetic as S
    src = E.ERFNet(20)
    S.load_synthetic(src, salt=3)
    sd = {"module." + k: v.clone() for k, v in src.state_dict().items()}
    sd["module.not.there"] = torch.zeros(1)
    del sd["module.decoder.output_conv.bias"]
    dst = E.ERFNet(20)
    before = dst.decoder.output_conv.bias.detach().clone()
    seen = []
    C.load_my_state_dict(dst, sd, report=lambda *a: seen.append(a[0]))
    assert seen == ["module.not.there"]
    for k, v in src.state_dict().items():
        if k == "decoder.output_conv.bias":
            assert torch.equal(dst.state_dict()[k], before)
        else:
            assert torch.equal(dst.state_dict()[k], v), k

    ck = C.checkpoint_dict(src, None, epoch=7, best_acc=0.5)
    assert set(ck) == {"epoch", "arch", "state_dict", "best_acc"} and all(k.startswith("module.") for k in ck["state_dict"])
    path, best = tmp_path / "checkpoint.pth.tar", tmp_path / "model_best.pth.tar"
    C.save_checkpoint(ck, True, str(path), str(best))
    other = E.ERFNet(20)
    epoch, acc = C.resume(str(best), other)
    assert (epoch, acc) == (7, 0.5)
    for k, v in src.state_dict().items():
        assert torch.equal(other.state_dict()[k], v), k
    torch.nn.DataParallel(E.ERFNet(20)).load_state_dict(ck["state_dict"])      # what the reference's resume does with it
