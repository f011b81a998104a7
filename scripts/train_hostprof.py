"""cProfile of the host side of Trainer.step (the step is launch-bound once the kernels are short)."""
import sys, time, torch, cProfile, pstats, io
sys.path.insert(0, ".")
from erfnet_pytorch_amd import erfnet as E, synthetic as S
from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(6, 3, 512, 1024, device=dev, generator=gen)
t = torch.randint(0, 20, (6, 1, 512, 1024), device=dev, generator=gen)
m = E.ERFNet(20); S.load_synthetic(m); m = m.to(dev).train(); m.compute_dtype = torch.bfloat16
tr = Trainer(m, CITYSCAPES_WEIGHTS)
for _ in range(5): tr.step(x, t)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20): tr.step(x, t)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45); print(s.getvalue()[:9000])
