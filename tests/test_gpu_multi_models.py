"""Multi-task variant (train/multi_models.py) on the GPU kernels, eval mode, against the golden vectors generated
from the reference (G8) and the CPU oracle."""
import pytest
import torch

from erfnet_pytorch_amd import synthetic as S
from tests.golden_util import check_packed, load, rel_err
from oracle import erfnet_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _model():
    from erfnet_pytorch_amd import multi_models as MM
    m = MM.Multi_models(14)
    sd = O.synthetic_state(O.multi_models_template(14))
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


@pytest.mark.parametrize("tag", ["a", "b"])
def test_fp32_outputs_vs_golden(tag):
    g = load("g8_multi_models.npz")
    m, _ = _model()
    n, h, w, seed = (int(v) for v in g[tag + "__in"])
    x = S.synthetic_images(n, h, w, seed=seed).to(DEV)
    with torch.no_grad():
        trav, depth, cls = m(x)
        feat, s1, s2 = m.encoder(x)
    assert trav.shape == (n, 1, h, w) and depth.shape == (n, 1, h, w) and cls.shape == (n, 14)
    assert trav.dtype == torch.float32 and trav.is_contiguous()
    check_packed(g, tag + "_skip1", s1.float().cpu(), 1e-4)
    check_packed(g, tag + "_skip2", s2.float().cpu(), 1e-4)
    check_packed(g, tag + "_feat", feat.float().cpu(), 1e-4)
    check_packed(g, tag + "_trav", trav.cpu(), 1e-4)
    check_packed(g, tag + "_depth", depth.cpu(), 1e-4)
    check_packed(g, tag + "_class", cls.cpu(), 1e-4)


def test_bf16_outputs_vs_oracle_and_training_is_refused():
    m, sd = _model()
    x = S.synthetic_images(2, 128, 256, seed=83)
    with torch.no_grad():
        ref = O.multi_models(sd, x)
    m.compute_dtype = torch.bfloat16
    with torch.no_grad():
        got = m(x.to(DEV))
    for a, b in zip(got, ref):
        assert a.dtype == torch.float32
        assert rel_err(a.cpu(), b) <= 4e-2           # bf16 activations through 15 blocks; fp32 bar above is 1e-4
    m.train()
    with pytest.raises(NotImplementedError):
        m(x.to(DEV))
