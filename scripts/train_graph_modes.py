"""Wall time per training step (batch 6, 3x512x1024, bf16) over a ONE-rank RCCL group with the collectives forced on:
launched kernel by kernel, one captured graph (all-reduces inside), five graphs with the all-reduces between them."""
import os, sys, time, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from erfnet_pytorch_amd import erfnet as E, synthetic as S
from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method=f"file://{tempfile.mkdtemp()}/store", rank=0, world_size=1, device_id=dev)
gen = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(6, 3, 512, 1024, device=dev, generator=gen)
t = torch.randint(0, 20, (6, 1, 512, 1024), device=dev, generator=gen)
import traceback
modes = sys.argv[1:] or ["False", "True", "segments"]
for mode in [{"False": False, "True": True}.get(a, a) for a in modes]:
    m = E.ERFNet(20); S.load_synthetic(m); m = m.to(dev).train(); m.compute_dtype = torch.bfloat16
    tr = Trainer(m, CITYSCAPES_WEIGHTS, group=dist.group.WORLD, world_size=1, force_collectives=True, graph=mode)
    for _ in range(5): tr.step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): loss, _ = tr.step(x, t)
    torch.cuda.synchronize()
    print(f"graph={mode!s:9s} {1e3 * (time.perf_counter() - t0) / 30:.2f} ms/step  loss {float(loss):.4f}", flush=True)
    del tr, m
dist.destroy_process_group()
