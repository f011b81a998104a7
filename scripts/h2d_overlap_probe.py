#!/usr/bin/env python3
"""Variants of the host-fed inference loop: chains per batch, number of device buffers, creation order of the copy stream.
A copy stream of priority -1 is left out of the list: that variant printed nothing for 150 s (the replays stalled)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from erfnet_pytorch_amd.iouEval import iouEval

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
model = bench.build_model(dev, "bf16", 20)
t = torch.randint(0, 20, (16, 1, 512, 1024), device=dev)


def run(ways, nbuf, prio, steps=30, first_stream=False):
    copy_stream = torch.cuda.Stream(dev, priority=prio) if first_stream else None
    host = [torch.randint(0, 256, (16, 3, 512, 1024), dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
    bufs = [torch.empty_like(h, device=dev) for h in host]
    ev = iouEval(20)
    with torch.no_grad():
        model.evaluate_batch(bufs[0], t, ev, torch.uint8, ways=ways)
    torch.cuda.synchronize()
    graphs = []
    for b in bufs:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"), torch.no_grad():
            model.evaluate_batch(b, t, ev, torch.uint8, ways=ways)
        graphs.append(g)
    if copy_stream is None:
        copy_stream = torch.cuda.Stream(dev, priority=prio)
    main = torch.cuda.current_stream(dev)
    copied = [torch.cuda.Event() for _ in bufs]
    consumed = [torch.cuda.Event() for _ in bufs]
    for e in consumed:
        e.record(main)

    def step(i):
        k = i % nbuf
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[k])
            bufs[k].copy_(host[k], non_blocking=True)
            copied[k].record(copy_stream)
        if i > 0:
            j = (i - 1) % nbuf
            main.wait_event(copied[j])
            graphs[j].replay()
            consumed[j].record(main)

    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(6, 6 + steps):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


T0 = time.time()
for ways, nbuf, prio, first in ((1, 2, 0, True), (2, 2, 0, True), (2, 2, 0, False), (2, 3, 0, False), (1, 2, 0, False), (2, 2, 0, True)):
    ms = run(ways, nbuf, prio, first_stream=first)
    print(f"[{time.time() - T0:6.1f} s] ways {ways} buffers {nbuf} copy-stream priority {prio} created-before-graphs {first}: {ms:.3f} ms/batch", flush=True)
