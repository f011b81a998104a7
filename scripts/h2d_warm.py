import os, sys, time, argparse
import torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
args = argparse.Namespace(batch=0, dtype="bf16", classes=20, height=512, width=1024, infer_ways=2)
for steps, warm in ((60, 250), (60, 60)):
    r = bench.run_infer_h2d(args, dev, steps, warm)
    print(steps, warm, r["ms_per_step"], flush=True)
