"""A few bf16 training steps (batch 6, 3x512x1024, launched -- not graph-replayed -- so every kernel shows in a trace).
Usage: rocprofv3 --kernel-trace --stats -d out -- python3 scripts/train_prof.py [steps] [fp32]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import erfnet as E, synthetic as S
from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
fp32 = len(sys.argv) > 2 and sys.argv[2] == "fp32"
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(6, 3, 512, 1024, device=dev, generator=gen)
t = torch.randint(0, 20, (6, 1, 512, 1024), device=dev, generator=gen)
m = E.ERFNet(20); S.load_synthetic(m); m = m.to(dev).train()
if not fp32:
    m.compute_dtype = torch.bfloat16
tr = Trainer(m, CITYSCAPES_WEIGHTS, fused_head_loss=True)
for _ in range(3): tr.step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): loss = tr.step(x, t)
torch.cuda.synchronize()
print(f"{'fp32' if fp32 else 'bf16'} train {1e3*(time.perf_counter()-t0)/steps:.3f} ms/step loss {float(loss[0] if isinstance(loss, tuple) else loss):.5f}", flush=True)
