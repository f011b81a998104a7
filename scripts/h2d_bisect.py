"""The host-fed inference leg of bench.py at several (steps, warm-up) pairs: the first copies out of a fresh pinned buffer are slow."""
import os, sys, time, argparse
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
args = argparse.Namespace(batch=0, dtype="bf16", classes=20, height=512, width=1024, infer_ways=2)
for steps, warm in ((20, 4), (40, 30), (40, 60), (100, 100)):
    r = bench.run_infer_h2d(args, dev, steps, warm)
    print(steps, warm, r["ms_per_step"], flush=True)
