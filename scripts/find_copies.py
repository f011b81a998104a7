import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
model = bench.build_model(dev, "bf16")
from erfnet_pytorch_amd.iouEval import iouEval
x = torch.randn(2, 3, 512, 1024, device=dev)
t = torch.randint(0, 20, (2, 1, 512, 1024), device=dev)
ev = iouEval(20)
def step():
    with torch.no_grad():
        labels = model.predict_labels(x, torch.uint8)
        ev.addBatch(labels.unsqueeze(1), t)
step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
evs = [e for e in prof.events() if "copy" in e.name.lower() or "Memcpy" in e.name or "memcpy" in e.name]
from collections import Counter
print(Counter(e.name for e in evs))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=12, max_name_column_width=60))
