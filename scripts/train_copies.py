"""Which host-side calls of a training step end in a device copy: aten::copy_ events of one launched step with their shapes and
Python call sites."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import erfnet as E, synthetic as S
from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(6, 3, 512, 1024, device=dev, generator=gen)
t = torch.randint(0, 20, (6, 1, 512, 1024), device=dev, generator=gen)
m = E.ERFNet(20); S.load_synthetic(m); m = m.to(dev).train(); m.compute_dtype = torch.bfloat16
tr = Trainer(m, CITYSCAPES_WEIGHTS)
for _ in range(3): tr.step(x, t)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.step(x, t)
    torch.cuda.synchronize()
seen = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::cat", "aten::fill_", "aten::zero_"):
        site = next((s for s in e.stack if "erfnet_pytorch_amd" in s), e.stack[0] if e.stack else "?")
        seen[(e.name, str(e.input_shapes)[:80], site[-90:])] += 1
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(n, *k, flush=True)
