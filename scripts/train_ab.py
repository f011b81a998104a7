"""A/B of Trainer options on one box: wall time per step and host enqueue time per step (time until step() returns)."""
import sys, time, torch
sys.path.insert(0, ".")
from erfnet_pytorch_amd import erfnet as E, synthetic as S
from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(6, 3, 512, 1024, device=dev, generator=gen)
t = torch.randint(0, 20, (6, 1, 512, 1024), device=dev, generator=gen)
for rep in range(2):
    for fused in (True, False):
        m = E.ERFNet(20); S.load_synthetic(m); m = m.to(dev).train(); m.compute_dtype = torch.bfloat16
        tr = Trainer(m, CITYSCAPES_WEIGHTS, fused_head_loss=fused)
        for _ in range(5): tr.step(x, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); host = 0.0
        for _ in range(30):
            h0 = time.perf_counter(); tr.step(x, t); host += time.perf_counter() - h0
        torch.cuda.synchronize()
        print(f"fused={fused} wall {1e3*(time.perf_counter()-t0)/30:.2f} ms/step, host enqueue {1e3*host/30:.2f} ms/step", flush=True)
