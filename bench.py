#!/usr/bin/env python3
"""Benchmark of the ERFNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 (default): BASELINE.json configs[1] -- ERFNet(20) inference, batch 16, bf16 activations
  with fp32 accumulation, 3x512x1024 synthetic images resident in HBM, fused arg-max and the
  iouEval confusion histogram; one step = one batch through the whole pipeline, replayed from a
  HIP graph.  The line also carries the roofline of the dominant kernel (measured here with
  events on the launch stream) and the CPU baseline (the oracle's fp32 forward, b=1, timed on
  this box's host cores with the eval_forwardTime protocol).
N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): BASELINE.json configs[2] --
  data-parallel training, batch 6 per GPU, weighted cross-entropy, Adam; see `run_train`.
Rank 0 prints exactly one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_FLOP_PER_IMG = 2 * 26_612_727_808            # 512x1024, 20 classes (53.225 GFLOP)
TRAIN_FLOP_PER_IMG = 2 * (3 * 26_612_727_808 - 46_006_272)
PEAK_BF16_TFLOPS = 2500.0                        # dense bf16 MFMA peak of MI355X
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["auto", "infer", "train"], default="auto")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 16 inference, 6 training)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def dist_setup(n_gpus):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
        local = 0
    return world, rank, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(x: float, world, dev) -> float:
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def build_model(dev, dtype_name):
    from erfnet_pytorch_amd import erfnet as E, synthetic as S
    model = E.ERFNet(20)
    S.load_synthetic(model)
    model = model.to(dev).eval()
    model.compute_dtype = torch.bfloat16 if dtype_name == "bf16" else None
    return model


def cpu_baseline(height, width, budget_s=20.0):
    """Oracle forward (fp32, b=1) on the host cores: 1 warm-up, then >= 3 timed iterations
    within ~budget_s (eval_forwardTime.py:40-59 protocol without the sleep)."""
    from erfnet_pytorch_amd import synthetic as S
    from oracle import erfnet_oracle as O
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    sd = O.synthetic_state(O.erfnet_template(20))
    x = torch.randn(1, 3, height, width)
    times = []
    with torch.no_grad():
        O.erfnet(sd, x)
        t_all = time.time()
        while len(times) < 20 and (len(times) < 3 or time.time() - t_all < budget_s):
            t0 = time.time()
            O.erfnet(sd, x)
            times.append(time.time() - t0)
    mean = sum(times) / len(times)
    return {"value": 1.0 / mean, "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle ERFNet(20) fp32 forward b=1 3x{height}x{width}, mean of {len(times)} iterations after 1 warm-up"}


def roofline_infer(model, x, dtype_name):
    """Time the dominant kernel (factorized conv, C=128) alone: every 128-channel 3x1/1x3 launch
    of one forward (32 launches, dilations 1..16) bracketed by events on the launch stream."""
    from erfnet_pytorch_amd import ops
    from erfnet_pytorch_amd._lib import AXIS_H, AXIS_W
    n, _, h, w = x.shape
    hh, ww, c = h // 8, w // 8, 128
    tdt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    feat = torch.randn(n, hh, ww, c, device=x.device).to(tdt)
    out = torch.empty_like(feat)
    blocks = [m for m in model.encoder.layers if getattr(m, "chann", 0) == 128]
    plan = []
    for blk in blocks:
        cc = blk._cached(ops.dt(feat))
        d = blk.dilated
        plan += [(cc["w1"], None, cc["t1"], None, AXIS_H, 1), (cc["w2"], cc["s2"], cc["t2"], None, AXIS_W, 1),
                 (cc["w3"], None, cc["t3"], None, AXIS_H, d), (cc["w4"], cc["s4"], cc["t4"], feat, AXIS_W, d)]
    def run():
        for (wp, s, t, r, ax, d) in plan:
            ops.conv1d(feat, wp, s, t, r, ax, d, True, out=out)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    avg_s = e0.elapsed_time(e1) * 1e-3 / (reps * len(plan))
    flop = 2.0 * 3 * c * c * n * hh * ww
    achieved = flop / avg_s / 1e12
    peak = PEAK_BF16_TFLOPS if dtype_name == "bf16" else PEAK_F32_TFLOPS
    return {"bound": "mfma", "kernel": "conv1d_factorized C=128", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak, "traffic": None, "avg_launch_us": avg_s * 1e6, "launches_per_step": len(plan),
            "flop_per_launch": flop}


def run_infer(args, world, rank, dev):
    from erfnet_pytorch_amd.iouEval import iouEval
    batch = args.batch or 16
    model = build_model(dev, args.dtype)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(batch, 3, args.height, args.width, device=dev, generator=gen)
    target = torch.randint(0, 20, (batch, 1, args.height, args.width), device=dev, generator=gen)
    ev = iouEval(20)

    def step():
        with torch.no_grad():
            labels = model.predict_labels(x, torch.uint8)
            ev.addBatch(labels.unsqueeze(1), target)

    step()                                    # packs weights, allocates the histogram
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        run = graph.replay
    else:
        run = step
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    miou = float(ev.getIoU()[0])
    res = {
        "metric": "img/s ERFNet 512x1024: bf16 inference (1 GPU) + training (1/2/4/8 MI355X)",
        "value": world * batch * args.steps / elapsed, "unit": "img/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"ERFNet(20) inference + fused argmax + iouEval histogram, batch {batch}/GPU, "
                               f"3x{args.height}x{args.width}, {'replicas only' if world > 1 else 'single GPU'}",
                   "hip_graph": graph is not None, "synthetic_miou": miou},
    }
    if rank == 0:
        flops = FWD_FLOP_PER_IMG * (args.height * args.width) / (512 * 1024)
        res["model_tflops"] = res["value"] * flops / 1e12
        if not args.no_roofline:
            res["roofline"] = roofline_infer(model, x, args.dtype)
    return res


def run_train(args, world, rank, dev):
    from erfnet_pytorch_amd import trainer
    return trainer.bench_train(args, world, rank, dev, barrier, max_over_ranks)


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU is visible and the package has no CPU path")
    world, rank, local = dist_setup(args.gpus)
    dev = torch.device("cuda", local)
    mode = args.mode
    if mode == "auto":
        mode = "infer" if world == 1 else "train"
    res = run_infer(args, world, rank, dev) if mode == "infer" else run_train(args, world, rank, dev)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.height, args.width)
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
