"""erfnet_pytorch_amd: the ERFNet forward/backward hot path of pandongwei/erfnet_pytorch on
hand-written MI355X (gfx950) kernels, behind the reference's nn.Module / iouEval API.

`synthetic` (closed-form weights and inputs) has no native dependency; everything else needs
libermi355.so and raises ImportError without it.
"""
from . import synthetic  # noqa: F401

__all__ = ["synthetic", "ERFNet", "Net", "Encoder", "Decoder", "DownsamplerBlock", "UpsamplerBlock",
           "non_bottleneck_1d", "iouEval"]


def __getattr__(name):
    if name in ("ERFNet", "Net", "Encoder", "Decoder", "DownsamplerBlock", "UpsamplerBlock", "non_bottleneck_1d"):
        from . import erfnet
        return getattr(erfnet, name)
    if name == "iouEval":
        from .iouEval import iouEval
        return iouEval
    raise AttributeError(name)
