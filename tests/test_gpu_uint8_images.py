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
