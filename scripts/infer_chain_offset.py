"""Experiment: two chains of launches (images 0-7 / 8-15), the second one started only after the first has finished its first
k layers, so that the chains run DIFFERENT kinds of kernels at the same time (vector-ALU-bound stem / output layer of one under
the memory- and matrix-bound blocks of the other).  Usage: python scripts/infer_chain_offset.py [k ...]"""
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
x = S.synthetic_images(16, 512, 1024, seed=1).to(dev)
t = S.synthetic_labels(16, 512, 1024, 20, seed=1).to(dev)
side = torch.cuda.Stream()
enc, dec = m.encoder, m.decoder
xs, ts = x.chunk(2), t.chunk(2)


def layers(ev, k):
    fs = [lambda y: enc.initial_block.forward_image(y, E.BF16)]
    fs += [layer.forward_nhwc for layer in enc.layers]
    fs += [layer.forward_nhwc for layer in dec.layers]

    def head(y):
        w, b = dec._head(E.BF16)
        from erfnet_pytorch_amd import ops
        _, labels = ops.output_conv(y, w, b, 20, False, True, True, torch.uint8)
        ev.addBatch(labels.unsqueeze(1), ts[k])
        return labels
    return fs + [head]


def build(offset):
    ev = iouEval(20)
    ev.reserve(dev)
    fa, fb = layers(ev, 0), layers(ev, 1)

    def step():
        cur = torch.cuda.current_stream()
        y = xs[0]
        for f in fa[:offset]:
            y = f(y)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            z = xs[1]
            for f in fb:
                z = f(z)
        for f in fa[offset:]:
            y = f(y)
        cur.wait_stream(side)

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


for off in ([int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 7, 11, 16, 0]):
    g = build(off)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    print(f"offset {off}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms/batch", flush=True)
