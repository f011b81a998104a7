"""Experiment: two BATCHES in flight.  Two HIP graphs, each the whole pipeline for its own batch of 16 (one chain, or two
chains inside), replayed alternately on two streams: a stream's replays are ordered among themselves, the two streams overlap.
Reports the time per batch.  Usage: python scripts/infer_two_batches.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import erfnet as E, synthetic as S  # noqa: E402
from erfnet_pytorch_amd.iouEval import iouEval  # noqa: E402

dev = torch.device("cuda", 0)
m = E.ERFNet(20)
S.load_synthetic(m)
m = m.to(dev).eval()
m.compute_dtype = torch.bfloat16
ev = iouEval(20)
ev.reserve(dev)


def graph_for(seed, ways):
    x = S.synthetic_images(16, 512, 1024, seed=seed).to(dev)
    t = S.synthetic_labels(16, 512, 1024, 20, seed=seed).to(dev)
    with torch.no_grad():
        m.evaluate_batch(x, t, ev, ways=ways)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            m.evaluate_batch(x, t, ev, ways=ways)
    return g, x, t


for ways in (1, 2):
    ga, gb = graph_for(1, ways), graph_for(2, ways)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for mode in ("one at a time", "two in flight"):
        def run(n):
            for i in range(n):
                if mode == "one at a time":
                    (ga if i % 2 == 0 else gb)[0].replay()
                else:
                    with torch.cuda.stream(sa if i % 2 == 0 else sb):
                        (ga if i % 2 == 0 else gb)[0].replay()
        run(6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        run(n)
        torch.cuda.synchronize()
        print(f"chains per batch {ways}, {mode}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms/batch", flush=True)
