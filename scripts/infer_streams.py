"""Experiment: bf16 inference of one batch (16 x 3 x 512 x 1024, labels + iouEval histogram) replayed from a HIP graph,
(a) as one chain of launches, (b) split along the batch into `ways` chains on separate streams (fork / join inside the capture),
so that one chain's launch prologues and tails run under the other's tiles.  Usage: python scripts/infer_streams.py [ways ...]"""
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


def build(ways):
    ev = iouEval(20)
    xs, ts = x.chunk(ways), t.chunk(ways)
    if os.environ.get("SPLIT"):                       # e.g. SPLIT=10,6: unequal runs of images
        sizes = [int(v) for v in os.environ["SPLIT"].split(",")]
        xs, ts, ways = x.split(sizes), t.split(sizes), len(sizes)
    side = [torch.cuda.Stream() for _ in range(ways - 1)]

    def step():
        cur = torch.cuda.current_stream()
        outs = [None] * ways
        for k, s in enumerate(side):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs[k + 1] = m.predict_labels(xs[k + 1], torch.uint8)
        outs[0] = m.predict_labels(xs[0], torch.uint8)
        for s in side:
            cur.wait_stream(s)
        for k in range(ways):
            ev.addBatch(outs[k].unsqueeze(1), ts[k])

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
    return g, ev


for ways in ([int(a) for a in sys.argv[1:]] or [1, 2]):
    g, ev = build(ways)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    print(f"ways {ways}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms/batch", flush=True)
