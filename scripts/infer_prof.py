"""bf16 inference steps (batch 16, 3x512x1024, labels + iouEval histogram), launched -- not graph-replayed -- so that a
kernel trace shows raw kernel durations.  Usage: rocprofv3 --kernel-trace --stats -d out -- python3 scripts/infer_prof.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import erfnet as E, synthetic as S
from erfnet_pytorch_amd.iouEval import iouEval
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
m = E.ERFNet(20); S.load_synthetic(m); m = m.to(dev).eval(); m.compute_dtype = torch.bfloat16
x = S.synthetic_images(16, 512, 1024, seed=1).to(dev)
t = S.synthetic_labels(16, 512, 1024, 20, seed=1).to(dev)
ev = iouEval(20)
with torch.no_grad():
    for _ in range(3):
        ev.addBatch(m.predict_labels(x, torch.uint8).unsqueeze(1), t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        ev.addBatch(m.predict_labels(x, torch.uint8).unsqueeze(1), t)
    torch.cuda.synchronize()
print(f"bf16 inference {1e3*(time.perf_counter()-t0)/steps:.3f} ms/batch (launched, no graph)", flush=True)
