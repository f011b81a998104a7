"""Experiment: which parts of the forward pass gain from running as two chains of launches (images 0-7 / 8-15 on two HIP
streams) and which are better left as one full-batch chain.  Regions: A = stem + 16->64 down-sampler + five 64-channel blocks,
B = 64->128 down-sampler + eight 128-channel blocks, C = decoder + output layer + histogram.  A mask like "011" splits B and C.
Usage: python scripts/infer_split_regions.py [mask ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from erfnet_pytorch_amd import erfnet as E, ops, synthetic as S  # noqa: E402
from erfnet_pytorch_amd.iouEval import iouEval  # noqa: E402

dev = torch.device("cuda", 0)
m = E.ERFNet(20)
S.load_synthetic(m)
m = m.to(dev).eval()
m.compute_dtype = torch.bfloat16
x = S.synthetic_images(16, 512, 1024, seed=1).to(dev)
t = S.synthetic_labels(16, 512, 1024, 20, seed=1).to(dev)
side = torch.cuda.Stream()
enc, dec = m.encoder, m.decoder


def region_a(img):
    y = enc.initial_block.forward_image(img, E.BF16)
    for layer in enc.layers[:6]:
        y = layer.forward_nhwc(y)
    return y


def region_b(y):
    for layer in enc.layers[6:]:
        y = layer.forward_nhwc(y)
    return y


def region_c(y, tgt, ev):
    _, labels = dec.forward_nhwc(y, want_logits=False, want_labels=True, label_dtype=torch.uint8)
    ev.addBatch(labels.unsqueeze(1), tgt)
    return labels


def run_region(fn, parts, split, extra=None):
    """parts: list of per-chain inputs (1 or 2 tensors).  Returns the outputs in the layout `split` asks for."""
    cur = torch.cuda.current_stream()
    if not split:
        full = parts[0] if len(parts) == 1 else torch.cat(parts)
        return [fn(full) if extra is None else fn(full, *extra(None))]
    halves = parts if len(parts) == 2 else list(parts[0].chunk(2))
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        b = fn(halves[1]) if extra is None else fn(halves[1], *extra(1))
    a = fn(halves[0]) if extra is None else fn(halves[0], *extra(0))
    cur.wait_stream(side)
    return [a, b]


def build(mask):
    ev = iouEval(20)
    ev.reserve(dev)
    ts = t.chunk(2)

    def step():
        parts = run_region(region_a, [x], mask[0] == "1")
        parts = run_region(region_b, parts, mask[1] == "1")
        run_region(region_c, parts, mask[2] == "1", extra=lambda k: (t if k is None else ts[k], ev))

    with torch.no_grad():
        s0 = torch.cuda.Stream()
        s0.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s0):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            step()
    return g


for mask in (sys.argv[1:] or ["000", "111", "100", "010", "001", "110", "011", "101"]):
    g = build(mask)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    print(f"split {mask}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms/batch", flush=True)
