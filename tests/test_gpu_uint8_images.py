"""8-bit images straight into the network (the input side of the reference's loops: ToTensor() turns decoded uint8 images
into float / 255, train/main.py:47,66, eval/eval_iou.py:31, before `model(inputs)`).  The initial block applies that
division on load, so a uint8 batch -- planar [N,3,H,W] or interleaved [N,H,W,3] -- must give, bit for bit, what the fp32
batch `images.float() / 255` gives: logits, fused labels, IoU counts, and a whole optimisation step."""
import numpy as np
import pytest
import torch

from erfnet_pytorch_amd import synthetic as S
from oracle import erfnet_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _images(n, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (n, 3, h, w), generator=g, dtype=torch.uint8)
    u8[0, :, :2, :] = 0                          # both ends of the range are present
    u8[0, :, 2:4, :] = 255
    return u8


def _model(nc=20, dtype=None):
    from erfnet_pytorch_amd import erfnet as E
    m = E.ERFNet(nc)
    m.load_state_dict(O.synthetic_state(O.erfnet_template(nc)))
    m = m.to(DEV).eval()
    m.compute_dtype = dtype
    return m


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
def test_uint8_images_equal_float_images_divided_by_255(dtype):
    from erfnet_pytorch_amd.iouEval import iouEval
    u8 = _images(3, 64, 128, seed=5)
    xf = (u8.float() / 255).to(DEV)              # what ToTensor() produces
    planar, inter = u8.to(DEV), u8.permute(0, 2, 3, 1).contiguous().to(DEV)
    m = _model(dtype=dtype)
    t = S.synthetic_labels(3, 64, 128, 20, seed=6).to(DEV)
    with torch.no_grad():
        ref = m(xf)
        lab = m.predict_labels(xf, torch.uint8)
        ev0 = iouEval(20)
        m.evaluate_batch(xf, t, ev0, torch.uint8)
        for img in (planar, inter):
            y = m(img)
            assert y.dtype == torch.float32 and torch.equal(y, ref)
            assert torch.equal(m.predict_labels(img, torch.uint8), lab)
            assert torch.equal(m.predict_labels(img, torch.uint8, ways=2), lab)
            ev = iouEval(20)
            assert torch.equal(m.evaluate_batch(img, t, ev, torch.uint8, ways=2), lab)
            assert torch.equal(ev.confusion(), ev0.confusion())
        assert torch.equal(m(planar, only_encode=True), m(xf, only_encode=True))
    # against the oracle on the float images: the fp32 bar of the inference tests
    if dtype is None:
        with torch.no_grad():
            gold = O.erfnet(O.synthetic_state(O.erfnet_template(20)), u8.float() / 255)
        assert (ref.cpu() - gold).abs().max().item() <= 1e-4 * gold.abs().max().item()
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4, 64, 128, dtype=torch.uint8, device=DEV))


@pytest.mark.parametrize("graph", [False, True])
def test_training_step_on_uint8_images_equals_the_step_on_float_images(graph):
    from erfnet_pytorch_amd import erfnet as E
    from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
    u8 = _images(2, 64, 128, seed=7)
    t = S.synthetic_labels(2, 64, 128, 20, seed=8).to(DEV)
    res = []
    for img in ((u8.float() / 255).to(DEV), u8.to(DEV), u8.permute(0, 2, 3, 1).contiguous().to(DEV)):
        torch.manual_seed(3)
        m = E.ERFNet(20)
        m.load_state_dict(O.synthetic_state(O.erfnet_template(20)))
        m = m.to(DEV).train()
        m.compute_dtype = torch.bfloat16
        tr = Trainer(m, CITYSCAPES_WEIGHTS, graph=graph)
        losses = [float(tr.step(img, t)[0]) for _ in range(3)]
        torch.cuda.synchronize()
        res.append((losses, tr.gflat.cpu().numpy().copy(), tr.flat.cpu().numpy().copy()))
    for other in res[1:]:
        assert other[0] == res[0][0]
        assert np.array_equal(other[1], res[0][1]) and np.array_equal(other[2], res[0][2])


def _raw_labels(n, h, w, seed):
    """8-bit label images as the loaders decode them: classes 0..18 and the void value 255."""
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(0, 19, (n, h, w), generator=g, dtype=torch.uint8)
    t[torch.rand((n, h, w), generator=g) < 0.1] = 255
    k = min(19, w)
    t[0, 0, :k] = torch.arange(k, dtype=torch.uint8)            # every class present, ragged run lengths
    t[0, 1, ::3] = 255
    return t


@pytest.mark.parametrize("shape", [(3, 64, 128), (1, 8, 8), (2, 24, 40)])
def test_uint8_label_images_with_relabel_equal_the_loaders_target_transform(shape):
    """ToLabel() + Relabel(255, 19) (eval/eval_iou.py:33-37, train/transform.py:53-68) on the device: counting with
    iouEval(relabel=...) on the raw 8-bit labels == counting on the int64 targets the reference's transform produces, and
    erf_labels_from_u8 == that transform, for sizes that are and are not multiples of the kernels' 8- and 16-label groups."""
    from erfnet_pytorch_amd import ops
    from erfnet_pytorch_amd.iouEval import iouEval
    n, h, w = shape
    raw = _raw_labels(n, h, w, seed=11)
    gold = raw.long().unsqueeze(1)                               # ToLabel()
    gold[gold == 255] = 19                                       # Relabel(255, 19)
    pred = S.synthetic_labels(n, h, w, 20, seed=12)
    got = ops.labels_from_u8(raw.to(DEV), (255, 19))
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), gold[:, 0])
    assert torch.equal(ops.labels_from_u8(raw.to(DEV)).cpu(), raw.long())
    flat = raw.reshape(-1).to(DEV)
    assert torch.equal(ops.labels_from_u8(flat[3:], (255, 19)).cpu(), gold.reshape(-1)[3:])      # a slice off the 16-byte grid
    ref = iouEval(20)
    ref.addBatch(pred.unsqueeze(1).to(DEV), gold.to(DEV))
    tp, fp, fn = O.iou_counts(pred.unsqueeze(1), gold, 20)
    assert torch.equal(ref.tp, tp) and torch.equal(ref.fp, fp) and torch.equal(ref.fn, fn)
    for p in (pred.to(DEV), pred.to(torch.uint8).to(DEV)):
        for y in (raw.to(DEV), raw.unsqueeze(1).to(DEV), raw.long().to(DEV)):
            ev = iouEval(20, relabel=(255, 19))
            ev.addBatch(p.unsqueeze(1), y)
            assert torch.equal(ev.confusion(), ref.confusion())
    ev = iouEval(20)                                             # without the relabelling the void value is an error, not a class
    ev.addBatch(pred.unsqueeze(1).to(DEV), raw.to(DEV))
    with pytest.raises(ValueError):
        ev.confusion()
    with pytest.raises(ValueError):
        iouEval(20, relabel=(255, 20))


def test_evaluation_and_training_on_raw_label_images():
    """The two loops fed with what a decoder delivers (uint8 images, uint8 label images with 255 = void): evaluate_batch with
    one and two chains counts what the transformed int64 targets give; a Trainer(relabel=(255, 19)) step equals the step on
    the transformed targets bit for bit."""
    from erfnet_pytorch_amd import erfnet as E
    from erfnet_pytorch_amd.iouEval import iouEval
    from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
    u8 = _images(4, 64, 128, seed=13).to(DEV)
    raw = _raw_labels(4, 64, 128, seed=14)
    gold = raw.long()
    gold[gold == 255] = 19
    m = _model(dtype=torch.bfloat16)
    ev0 = iouEval(20)
    with torch.no_grad():
        lab = m.evaluate_batch(u8, gold.unsqueeze(1).to(DEV), ev0, torch.uint8)
        for ways in (1, 2):
            ev = iouEval(20, relabel=(255, 19))
            assert torch.equal(m.evaluate_batch(u8, raw.unsqueeze(1).to(DEV), ev, torch.uint8, ways=ways), lab)
            assert torch.equal(ev.confusion(), ev0.confusion())
    res = []
    for tgt, relabel in ((gold.to(DEV), None), (raw.to(DEV), (255, 19)), (raw.unsqueeze(1).to(DEV), (255, 19))):
        torch.manual_seed(3)
        mt = E.ERFNet(20)
        mt.load_state_dict(O.synthetic_state(O.erfnet_template(20)))
        mt = mt.to(DEV).train()
        mt.compute_dtype = torch.bfloat16
        tr = Trainer(mt, CITYSCAPES_WEIGHTS, graph=True, relabel=relabel)
        losses = [float(tr.step(u8[:2], tgt[:2])[0]) for _ in range(3)]
        torch.cuda.synchronize()
        res.append((losses, tr.flat.cpu().numpy().copy()))
    for other in res[1:]:
        assert other[0] == res[0][0] and np.array_equal(other[1], res[0][1])
