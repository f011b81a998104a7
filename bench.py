#!/usr/bin/env python3
"""Benchmark of the ERFNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1, headline (`value`): BASELINE.json configs[1] -- ERFNet(20) inference, batch 16, bf16 activations with fp32
  accumulation, 3x512x1024 synthetic images resident in HBM, fused arg-max and the iouEval confusion histogram; one
  step = one batch through the whole pipeline, replayed from a HIP graph; the batch runs as two chains of launches on two
  HIP streams (ERFNet.evaluate_batch(ways=2): images 0-7 and 8-15, bit-identical labels and counts; `--infer-ways 1` = one chain).
  The same line carries
    "inference_one_chain" the same leg with one chain of launches (short leg),
    "training"       BASELINE.json configs[2]: the bf16 training step, batch 6 (its own steps / warm-up),
    "inference_with_h2d" the headline pipeline fed from pinned HOST memory: uint8 images + uint8 label images on a copy stream,
                     double-buffered, ToTensor()'s / 255 applied by the initial block on load and Relabel(255, 19) by the
                     histogram while counting (PCIe-inclusive; never `value`); the copy stream is the fastest of four
                     fresh ones (`copy_stream_trials_ms`: a copy stream that shares a hardware queue with a compute stream
                     takes turns with its kernels),
    "fp32"           {"inference", "training"}: the same two legs with fp32 storage (short legs),
    "roofline"       the dominant kernel (full-batch launches, one chain), timed here with events on the launch stream: the time
                     a replayed forward pass gains when one more pass over the kernel's launches is appended (`avg_launch_us`,
                     what a launch costs inside a step) and the same launches back to back (`avg_launch_us_back_to_back`),
    "roofline_block" the same measurement priced per non_bottleneck_1d block against the block's algorithmic bytes
                     (x in, y out), which shows what the two-launch block spends above the one-pass model,
    "cpu_baseline"   the oracle's fp32 forward, b=1, on this box's host cores (eval_forwardTime protocol).
    "measured_peak"  what this GPU sustains on an MFMA-only loop (bf16 on constant and on varied operands, fp32) and on a
                     streaming copy (erfnet_pytorch_amd/probe.py);
                     every roofline object carries `frac_of_measured` beside `frac` (which is against the nominal peak).
N > 1, headline: the data-parallel TRAINING step of configs[2]
  -- batch 6 per GPU, weighted cross-entropy, Adam, gradient all-reduce over RCCL inside the timed region ("weak"
  scaling: per-GPU work fixed); the step is replayed from five HIP graphs (forward | three backward sections | Adam) with each
  gradient bucket's asynchronous all-reduce issued right after the section that completes it, i.e. under the next section's
  kernels (`--train-graph`: one graph with the all-reduces captured, `--no-graph`: kernel-by-kernel launches).  "inference"
  then holds the inference leg (independent shards per rank: the path has no exchange step), "world_size" what the process
  group reports, and "allreduce" the time of each gradient bucket's all-reduce alone plus `exposed_ms`, the step time with
  the collectives minus the step time with them muted.
  The N ranks come either from a launcher (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) or, when
  `--gpus N` is given without that environment, from this script itself: the parent starts N fresh child processes of
  bench.py before it touches the GPU, relays rank 0's JSON line and exits non-zero if any rank does.
`--mode train` / `--mode infer` force one leg as the headline at any N.  Rank 0 prints exactly one JSON line.
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
HEAD_MAC_PER_CLASS = 16 * 256 * 512 * 4          # Decoder.output_conv at 512x1024, per class (167,772,160 MAC for 20)
TRAIN_FLOP_PER_IMG = 2 * (3 * 26_612_727_808 - 46_006_272)


def forward_flop(classes, height, width):
    """Multiply-accumulate count of one ERFNet(classes) forward x 2: everything but the head scales with the area only."""
    mac = 26_612_727_808 - 20 * HEAD_MAC_PER_CLASS + classes * HEAD_MAC_PER_CLASS
    return 2.0 * mac * (height * width) / (512 * 1024)
PEAK_BF16_TFLOPS = 2500.0                        # dense bf16 MFMA peak of MI355X
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


FUSE_BLOCKS_FLAG = ""


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["auto", "infer", "train", "imagenet", "multi"], default="auto")
    ap.add_argument("--classes", type=int, default=20, help="ERFNet(num_classes) of the inference leg (6: the 1024x2048 config)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 16 inference, 6 training)")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--infer-ways", type=int, default=2,
                    help="inference: cut the batch into this many runs of images, each a chain of launches on its own HIP stream "
                         "(ERFNet.evaluate_batch(ways=...); bit-identical labels and counts; 1 = one chain)")
    ap.add_argument("--train-graph", action="store_true", help="N > 1: capture the training step (RCCL all-reduces included) into a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="auto mode: skip the training leg")
    ap.add_argument("--train-batch", type=int, default=6)
    ap.add_argument("--train-steps", type=int, default=30)
    ap.add_argument("--train-warmup", type=int, default=5)
    ap.add_argument("--infer-steps", type=int, default=30, help="N > 1: steps of the inference leg reported beside the training headline")
    ap.add_argument("--infer-warmup", type=int, default=5)
    ap.add_argument("--no-fp32", action="store_true", help="auto mode at N = 1: skip the fp32 legs")
    ap.add_argument("--fuse-blocks", default="", help="inference: channel counts whose dilation-1 blocks run as one launch, e.g. 16,64 "
                                                     "(default: the package's erfnet.FUSE_BLOCKS)")
    return ap.parse_args()


def launch_ranks(args):
    """`--gpus N` (N > 1) without a launcher's environment: start N fresh ranks of this script, one per GPU.  Runs before
    anything in this process touches the GPU (children are new processes, nothing is re-executed in place).  Rank 0's
    stdout is relayed line by line; the exit status is the first non-zero one of the ranks; when one rank dies the others
    are given a short grace period and then ended by PID."""
    import socket
    import subprocess
    n = args.gpus
    if os.environ.get("ERF_BENCH_REHEARSAL") != "1":
        have = torch.cuda.device_count()               # counting devices does not initialise the GPU
        if have < n:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=r == 0))
    for line in procs[0].stdout:                       # ends when rank 0 closes its stdout
        sys.stdout.write(line)
        sys.stdout.flush()
    rc, deadline = 0, None
    while any(p.poll() is None for p in procs):
        codes = [p.poll() for p in procs]
        if deadline is None and any(c not in (None, 0) for c in codes):
            deadline = time.time() + 20
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    for p in procs:
        rc = rc or p.returncode
    return rc


def dist_setup(n_gpus):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if n_gpus != world:
        raise SystemExit(f"bench.py --gpus {n_gpus} but the process group has WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("ERF_BENCH_REHEARSAL") == "1":
            # rehearsal of the multi-rank code path on a box with ONE GPU: every rank on device 0, gloo instead of RCCL
            # (RCCL refuses two ranks on one device), collectives on device tensors staged through host copies made
            # here (gloo's own device-tensor path stalled at full size with two busy processes on one GPU).  The
            # numbers of such a run mean nothing; the JSON line says so.
            import faulthandler
            faulthandler.dump_traceback_later(int(os.environ.get("ERF_BENCH_REHEARSAL_LIMIT", "90")), exit=True)
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
            from erfnet_pytorch_amd.dp import stage_collectives_through_host
            stage_collectives_through_host()
        else:
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


def build_model(dev, dtype_name, classes=20):
    from erfnet_pytorch_amd import erfnet as E, synthetic as S
    if FUSE_BLOCKS_FLAG:
        E.FUSE_BLOCKS = tuple(int(c) for c in FUSE_BLOCKS_FLAG.split(","))
    model = E.ERFNet(classes)
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


def measured_traffic(kernel_prefix, pixels):
    """(HBM bytes per launch of the dominant kernel, file they come from): the newest committed rocprofv3 PMC passes
    (profiles/rNN_pmc_traffic.json: (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch, mean over the kernel's
    variants) -- a figure carried over from a profiled run of the same command, not measured in this run (the counters
    need rocprofv3); only quoted when this run has the shape the counters were collected on."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not paths or pixels != 16 * 128 * 256:
        return None, None
    vals = [v for k, v in json.load(open(paths[-1])).items() if kernel_prefix in k]
    return (sum(vals) / len(vals) if vals else None), os.path.join("profiles", os.path.basename(paths[-1]))


def roofline_infer(model, x, dtype_name):
    """Time the dominant kernel of one forward with events on the launch stream: replayed alone from a HIP graph and (the one-launch
    block kernel) as the time a replayed forward pass gains when one more pass over those launches is appended -- the figure
    reported, because that is how long the launches take inside a step.

    bf16: the 64-channel non_bottleneck_1d blocks (7 per forward, the largest share of the step): by default one launch
    per block (nb1d_block64s_kernel: x in, y out = 2 tensors of N*H/4*W/4*64*2 bytes for 4*2*3*C*C flop per pixel, 384
    flop/B -- just above the MI355X ridge of 2500 TFLOP/s / 8 TB/s = 312 flop/B, priced against HBM like the two-launch
    form it replaces, with the matrix rate beside it); with 64 not in erfnet.FUSE_BLOCKS two conv1d_pair launches per block
    (154..192 flop/B, HBM-bound: input + output (+ the residual tensor for the second pair)).
    fp32: the single factorized convolution on the 128-channel blocks (MFMA-bound, 32 launches)."""
    from erfnet_pytorch_amd import ops
    from erfnet_pytorch_amd._lib import AXIS_H, AXIS_W
    n, _, h, w = x.shape
    from erfnet_pytorch_amd import erfnet as E
    fused = dtype_name == "bf16" and 64 in E.FUSE_BLOCKS
    if dtype_name == "bf16":
        hh, ww, c = h // 4, w // 4, 64
        # block inputs are ReLU outputs in the network (about half zeros): the probe tensor has the same statistics
        feat = torch.relu(torch.randn(n, hh, ww, c, device=x.device)).to(torch.bfloat16)
        mid, out = torch.empty_like(feat), torch.empty_like(feat)
        blocks = [m for m in list(model.encoder.layers) + list(model.decoder.layers) if getattr(m, "chann", 0) == 64]
        tensor_bytes = feat.numel() * 2
        plan = []
        if fused:
            # the whole block in one launch (csrc/nb1d_block64s.hip): x in, y out
            for blk in blocks:
                cc = blk._cached(ops.dt(feat))
                plan.append(((cc["w1"], cc["w2"], cc["w3"], cc["w4"]), cc["table"]))
            def run():
                # chained as in the network: a block reads what the block before it has just written (the first one reads `feat`,
                # which nothing writes, so every pass starts from the same values)
                src = feat
                for i, (w4, table) in enumerate(plan):
                    dst = out if i & 1 else mid
                    if ops.nb1d_block(src, w4, table, 1, out=dst) is None:
                        raise RuntimeError("block kernel declined the benchmark shape")
                    src = dst
            bytes_per_launch = 2.0 * tensor_bytes
            flop = 4 * 2.0 * 3 * c * c * n * hh * ww
            name = "nb1d_block64s_kernel (whole non_bottleneck_1d block, 64 channels, one launch)"
        else:
            for blk in blocks:
                cc = blk._cached(ops.dt(feat))
                plan += [(feat, mid, cc["w1"], cc["t1"], cc["w2"], cc["s2"], cc["t2"], None, 1),
                         (mid, out, cc["w3"], cc["t3"], cc["w4"], cc["s4"], cc["t4"], feat, blk.dilated)]
            def run():
                for (src, dst, wa, ta, wb, sb, tb, r, d) in plan:
                    if ops.conv1d_pair(src, wa, ta, wb, sb, tb, r, d, True, out=dst) is None:
                        raise RuntimeError("pair kernel declined the benchmark shape")
            bytes_per_launch = sum(tensor_bytes * (3 if r is not None else 2) for (*_, r, _d) in plan) / len(plan)
            flop = 2 * 2.0 * 3 * c * c * n * hh * ww
            name = "conv1d_pair_kernel<64> (fused conv3x1+conv1x3, 64-channel blocks)"
    else:
        hh, ww, c = h // 8, w // 8, 128
        feat = torch.randn(n, hh, ww, c, device=x.device)
        out = torch.empty_like(feat)
        blocks = [m for m in model.encoder.layers if getattr(m, "chann", 0) == 128]
        plan = []
        for blk in blocks:
            cc = blk._cached(ops.dt(feat))
            d = blk.dilated
            plan += [(cc["w1"], None, cc["t1"], None, AXIS_H, 1), (cc["w2"], cc["s2"], cc["t2"], None, AXIS_W, 1),
                     (cc["w3"], None, cc["t3"], None, AXIS_H, d), (cc["w4"], cc["s4"], cc["t4"], feat, AXIS_W, d)]
        def run():
            for (wp, s_, t_, r, ax, d) in plan:
                ops.conv1d(feat, wp, s_, t_, r, ax, d, True, out=out)
        bytes_per_launch = feat.numel() * 4 * 2.25
        flop = 2.0 * 3 * c * c * n * hh * ww
        name = "conv1d_dma_kernel<float,128> (factorized conv, 128-channel blocks)"
    run()
    torch.cuda.synchronize()
    # replayed from a HIP graph so that the events bracket back-to-back kernels, not host launch gaps; the graph holds the
    # plan several times over, so that the gap between two graph launches is spread over as many kernels as in a real step
    inner = 4
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):   # other threads (a process-group watchdog) may query events meanwhile
        for _ in range(inner):
            run()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    avg_s = e0.elapsed_time(e1) * 1e-3 / (reps * inner * len(plan))
    sustained_us = None
    if fused:
        # Dozens of these launches back to back (the loop above: 13 ms of nothing else) run ~12 % slower per launch than the same
        # launches inside a forward pass, where they come five and two at a time between lighter kernels (rocprofv3 shows both in
        # one trace: 77 us in the run of 168, 67.7 us in the steps).  The figure reported is the one the step sees: the time
        # a replayed forward pass (one chain) gains when one more pass over the seven blocks is appended to it.
        sustained_us = avg_s * 1e6

        def forward_pass():
            with torch.no_grad():
                model.predict_labels(x, torch.uint8)
        forward_pass()
        torch.cuda.synchronize()
        g_plain, g_plus = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_plain, capture_error_mode="thread_local"):
            forward_pass()
        with torch.cuda.graph(g_plus, capture_error_mode="thread_local"):
            forward_pass()
            run()
        for g in (g_plain, g_plus):
            g.replay()
        torch.cuda.synchronize()
        tot = {id(g_plain): 0.0, id(g_plus): 0.0}
        rounds, per = 4, 10
        for _ in range(rounds):                          # alternating, so that both see the same clocks
            for g in (g_plain, g_plus):
                e0.record()
                for _ in range(per):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                tot[id(g)] += e0.elapsed_time(e1)
        in_step_s = (tot[id(g_plus)] - tot[id(g_plain)]) * 1e-3 / (rounds * per * len(plan))
        if in_step_s > 0.5 * avg_s:                       # (a difference of two noisy totals: an implausible one is not reported)
            avg_s = in_step_s
    if dtype_name == "bf16":
        achieved = bytes_per_launch / avg_s / 1e9
        kernel_key = "nb1d_block64s_kernel" if fused else "conv1d_pair_kernel<64,"
        traffic, traffic_file = measured_traffic(kernel_key, n * hh * ww)
        per_launch = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                      "frac": achieved / PEAK_HBM_GBS, "traffic": traffic, "traffic_from_profile": traffic_file,
                      "avg_launch_us": avg_s * 1e6, "avg_launch_us_back_to_back": sustained_us, "launches_per_step": len(plan),
                      "bytes_per_launch": bytes_per_launch, "flop_per_launch": flop, "mfma_tflops": flop / avg_s / 1e12}
        # The same time priced per BLOCK against the block's algorithmic bytes (x in once, y out once = 2 tensors).  One launch
        # per block: the same figure.  Two conv1d_pair launches per block move 5 tensors (x, mid | mid, x again, y).
        launches = 1 if fused else 2
        block_s = launches * avg_s
        block_alg = 2.0 * tensor_bytes
        per_block = {"bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS, "achieved": block_alg / block_s / 1e9,
                     "frac": block_alg / block_s / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes_per_block": block_alg,
                     "launch_bytes_per_block": (2.0 if fused else 5.0) * tensor_bytes, "launches_per_block": launches,
                     "traffic": None if traffic is None else launches * traffic,
                     "avg_block_us": block_s * 1e6, "blocks_per_step": len(plan) // launches,
                     "mfma_tflops": launches * flop / block_s / 1e12,
                     "what": "64-channel non_bottleneck_1d block; model: x read once, y written once"}
        return per_launch, per_block
    achieved = flop / avg_s / 1e12
    return {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_F32_TFLOPS, "traffic": None, "avg_launch_us": avg_s * 1e6, "launches_per_step": len(plan),
            "flop_per_launch": flop}, None


def run_infer(args, world, rank, dev, dtype=None, steps=None, warmup=None, roofline=True):
    from erfnet_pytorch_amd.iouEval import iouEval
    args = argparse.Namespace(**vars(args))
    args.dtype = dtype or args.dtype
    args.steps = steps or args.steps
    args.warmup = args.warmup if warmup is None else warmup
    batch = args.batch or 16
    model = build_model(dev, args.dtype, args.classes)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(batch, 3, args.height, args.width, device=dev, generator=gen)
    target = torch.randint(0, args.classes, (batch, 1, args.height, args.width), device=dev, generator=gen)
    ev = iouEval(args.classes)

    def step():
        with torch.no_grad():
            model.evaluate_batch(x, target, ev, torch.uint8, ways=args.infer_ways)      # labels, then iouEval.addBatch

    step()                                    # packs weights, allocates the histogram
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):   # other threads (a process-group watchdog) may query events meanwhile
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
        "config": {"workload": f"ERFNet({args.classes}) inference + fused argmax + iouEval histogram, batch {batch}/GPU, "
                               f"3x{args.height}x{args.width}, {'independent shards per rank' if world > 1 else 'single GPU'}",
                   "hip_graph": graph is not None, "streams": max(1, min(args.infer_ways, batch)), "synthetic_miou": miou},
    }
    if rank == 0:
        res["model_tflops"] = res["value"] * forward_flop(args.classes, args.height, args.width) / 1e12
        if roofline and not args.no_roofline:
            res["roofline"], per_block = roofline_infer(model, x, args.dtype)
            if per_block is not None:
                res["roofline_block"] = per_block
    return res


COPY_STREAM = {}     # device index -> the stream host-to-device copies travel on; created before any other stream (see main)


def run_infer_h2d(args, dev, steps, warmup):
    """The inference pipeline fed from HOST memory, as an evaluation loop is (eval/eval_iou.py:84-93: the loader hands over
    decoded 8-bit images and label images): every batch travels as uint8 images [N,3,H,W] + uint8 label images [N,1,H,W]
    (255 = void) from pinned host memory on a copy stream into one of two device buffers while the previous batch computes;
    the initial block applies ToTensor()'s / 255 on load and the histogram applies Relabel(255, 19) (eval/eval_iou.py:33-37)
    while counting.  Reported beside the headline, never as `value`."""
    from erfnet_pytorch_amd.iouEval import iouEval
    batch = args.batch or 16
    model = build_model(dev, args.dtype, args.classes)
    g = torch.Generator().manual_seed(99)
    host = [torch.randint(0, 256, (batch, 3, args.height, args.width), generator=g, dtype=torch.uint8).pin_memory() for _ in range(2)]
    void = args.classes - 1                      # the class Relabel maps the void value to (19 of 20 on Cityscapes)
    host_t = []
    for _ in range(2):
        t = torch.randint(0, max(void, 1), (batch, 1, args.height, args.width), generator=g, dtype=torch.uint8)
        t[torch.rand(t.shape, generator=g) < 0.05] = 255
        host_t.append(t.pin_memory())
    bufs = [torch.empty_like(h, device=dev) for h in host]
    tbufs = [torch.empty_like(h, device=dev) for h in host_t]
    tbufs[0].copy_(host_t[0])
    ev = iouEval(args.classes, ignoreIndex=void, relabel=(255, void))
    with torch.no_grad():
        model.evaluate_batch(bufs[0], tbufs[0], ev, torch.uint8, ways=args.infer_ways)
    torch.cuda.synchronize()
    graphs = []
    for b, tb in zip(bufs, tbufs):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, capture_error_mode="thread_local"), torch.no_grad():
            model.evaluate_batch(b, tb, ev, torch.uint8, ways=args.infer_ways)
        graphs.append(gr)
    main = torch.cuda.current_stream(dev)
    copied = [torch.cuda.Event() for _ in bufs]
    consumed = [torch.cuda.Event() for _ in bufs]
    for e in consumed:
        e.record(main)
    # HIP deals its streams round-robin onto a few hardware queues (4 per process by default), and a copy stream that lands on the
    # queue of one of the compute streams (the caller's, the second chain's, the ones captures use) takes turns with its kernels
    # instead of running beside them: the same loop read 1.9 ms per batch on one stream and 1.25-1.45 ms on another.  So a
    # few fresh streams are tried for a dozen batches each and the fastest one is kept.
    candidates = [COPY_STREAM[dev.index]] if dev.index in COPY_STREAM else []
    candidates += [torch.cuda.Stream(dev) for _ in range(4 - len(candidates))]
    copy_stream = candidates[0]

    def step(i):
        k = i & 1
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[k])          # the batch that last used this buffer has been computed
            bufs[k].copy_(host[k], non_blocking=True)
            tbufs[k].copy_(host_t[k], non_blocking=True)
            copied[k].record(copy_stream)
        if i > 0:                                        # compute the batch copied one step earlier
            j = (i - 1) & 1
            main.wait_event(copied[j])
            graphs[j].replay()
            consumed[j].record(main)

    trial = {}
    for k, cand in enumerate(candidates):
        copy_stream = cand
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5, 17):
            step(i)
        torch.cuda.synchronize()
        trial[k] = (time.perf_counter() - t0) / 12
        for e in consumed:                                # the next candidate starts from idle buffers
            e.record(main)
    copy_stream = candidates[min(trial, key=trial.get)]
    for i in range(warmup + 1):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup + 1, warmup + 1 + steps):
        step(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    nbytes = host[0].numel() + host_t[0].numel()
    return {"value": batch * steps / elapsed, "unit": "img/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps, "warmup": warmup,
            "dtype": args.dtype, "h2d_bytes_per_step": nbytes, "h2d_gbs": nbytes * steps / elapsed / 1e9,
            "copy_stream_trials_ms": [round(1e3 * trial[k], 4) for k in sorted(trial)],
            "config": {"workload": f"as the headline, each batch copied from pinned host memory as uint8 images [N,3,{args.height},{args.width}] "
                                   "+ uint8 label images (255 = void) on a copy stream, double-buffered; / 255 applied by the initial "
                                   "block on load, Relabel(255, void class) by the histogram while counting",
                       "hip_graph": True, "streams": max(1, min(args.infer_ways, batch))}}


def run_multi(args, world, rank, dev):
    """Multi-task variant (train/multi_models.py): shared encoder, two regression decoders with skips, classifier."""
    from erfnet_pytorch_amd import multi_models as MM, synthetic as S
    batch = args.batch or 8
    model = MM.Multi_models(14)
    S.load_synthetic(model)
    model = model.to(dev).eval()
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else None
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(batch, 3, args.height, args.width, device=dev, generator=gen)

    def step():
        with torch.no_grad():
            return model(x)

    step()
    torch.cuda.synchronize()
    graph = None
    if not args.no_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):   # other threads (a process-group watchdog) may query events meanwhile
            step()
    run = graph.replay if graph is not None else step
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
    return {"metric": "img/s Multi_models inference (traversability + depth + class)", "value": world * batch * args.steps / elapsed,
            "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Multi_models(14) eval, batch {batch}/GPU, 3x{args.height}x{args.width}", "hip_graph": graph is not None}}


def run_train(args, world, rank, dev, dtype=None, steps=None, warmup=None, batch=None):
    from erfnet_pytorch_amd import trainer
    args = argparse.Namespace(**vars(args))
    args.dtype = dtype or args.dtype
    args.steps = steps or args.steps
    args.warmup = args.warmup if warmup is None else warmup
    args.batch = batch if batch is not None else args.batch
    return trainer.bench_train(args, world, rank, dev, barrier, max_over_ranks)


LEG_KEYS = ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "model_tflops")


def leg(res):
    return {k: res[k] for k in LEG_KEYS if k in res}


def free_gpu():
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def add_measured_peaks(res, dev):
    """`measured_peak` and a `frac_of_measured` beside every `frac`: the same achieved rate against what this GPU sustained on
    the probe kernels in this very run (the nominal peaks are datasheet numbers the part does not reach under load)."""
    from erfnet_pytorch_amd import probe
    mp = probe.measured_peaks(dev)
    res["measured_peak"] = mp
    for key in ("roofline", "roofline_block"):
        rf = res.get(key)
        if not rf:
            continue
        if rf["bound"] == "hbm":
            rf["frac_of_measured"] = rf["achieved"] / mp["hbm_copy_gbs"]
            if "mfma_tflops" in rf:
                rf["mfma_frac"] = rf["mfma_tflops"] / PEAK_BF16_TFLOPS
                rf["mfma_frac_of_measured"] = rf["mfma_tflops"] / mp["mfma_bf16_tflops"]
        else:
            rf["frac_of_measured"] = rf["achieved"] / mp["mfma_f32_tflops"]
    def price(leg, dtype):
        if leg and "model_tflops" in leg:
            peak, meas = (PEAK_BF16_TFLOPS, mp["mfma_bf16_tflops"]) if dtype == "bf16" else (PEAK_F32_TFLOPS, mp["mfma_f32_tflops"])
            leg["mfma_frac"] = leg["model_tflops"] / peak
            leg["mfma_frac_of_measured"] = leg["model_tflops"] / meas
    price(res, res.get("dtype"))
    for k in ("inference_one_chain", "training", "inference"):
        price(res.get(k), (res.get(k) or {}).get("dtype"))
    for k, leg in (res.get("fp32") or {}).items():
        price(leg, "fp32")


def main():
    global FUSE_BLOCKS_FLAG
    args = parse()
    FUSE_BLOCKS_FLAG = args.fuse_blocks
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU is visible and the package has no CPU path")
    world, rank, local = dist_setup(args.gpus)
    dev = torch.device("cuda", local)
    if world == 1 and args.mode == "auto":
        # The copy stream of the host-fed leg is created before the compute streams; a high-priority copy stream stalled the
        # graph replays for minutes (scripts/h2d_overlap_probe.py), so it has the default priority.
        # Buffers 2 vs 3 and the creation order made no difference there (1.41 ms per batch with two chains).
        COPY_STREAM[dev.index] = torch.cuda.Stream(dev)
    if args.mode == "imagenet":
        from erfnet_pytorch_amd import trainer
        res = trainer.bench_imagenet(args, world, rank, dev, barrier, max_over_ranks)
    elif args.mode == "multi":
        res = run_multi(args, world, rank, dev)
    elif args.mode == "train":
        res = run_train(args, world, rank, dev)
    elif args.mode == "infer":
        res = run_infer(args, world, rank, dev)
    elif world == 1:
        # auto, one GPU: inference is the headline; training, and both legs in fp32, ride in the same line
        res = run_infer(args, world, rank, dev)
        free_gpu()
        if args.infer_ways > 1:
            one = argparse.Namespace(**vars(args))
            one.infer_ways = 1
            res["inference_one_chain"] = leg(run_infer(one, world, rank, dev, steps=15, warmup=3, roofline=False))
            free_gpu()
        # The host-fed leg after the resident ones: as the first leg of a fresh process it reads 1.9-2.1 ms per batch whatever its
        # warm-up (scripts/h2d_bisect.py: 2.09 / 1.57 / 1.22 / 1.31 ms for four runs in a row in one process).
        res["inference_with_h2d"] = run_infer_h2d(args, dev, steps=60, warmup=60)
        free_gpu()
        if not args.no_train:
            res["training"] = leg(run_train(args, world, rank, dev, steps=args.train_steps, warmup=args.train_warmup, batch=args.train_batch))
            free_gpu()
        if not args.no_fp32:
            fp32 = {"inference": leg(run_infer(args, world, rank, dev, dtype="fp32", steps=10, warmup=3, roofline=False))}
            free_gpu()
            if not args.no_train:
                fp32["training"] = leg(run_train(args, world, rank, dev, dtype="fp32", steps=8, warmup=2, batch=args.train_batch))
                free_gpu()
            res["fp32"] = fp32
    else:
        # auto, several GPUs: the data-parallel training step is the headline (the leg with an exchange step); the inference
        # pipeline (independent shards) is reported beside it
        res = run_train(args, world, rank, dev, batch=args.train_batch)
        res["config"]["one_gpu_counterpart"] = ("`training.value` of the `--gpus 1` line (same step, same per-GPU batch); that line's own "
                                                 "`value` is the inference leg, which is not what this `value` scales from")
        free_gpu()
        res["inference"] = leg(run_infer(args, world, rank, dev, steps=args.infer_steps, warmup=args.infer_warmup, roofline=False))
    res.setdefault("world_size", world)
    if rank == 0:
        if not args.no_roofline and args.mode in ("auto", "infer", "train"):
            add_measured_peaks(res, dev)
        if not args.no_cpu_baseline and world == 1 and args.mode != "multi":
            res["cpu_baseline"] = cpu_baseline(args.height, args.width)
        if os.environ.get("ERF_BENCH_REHEARSAL") == "1" and world > 1:
            res["data"] = "synthetic; REHEARSAL: all ranks share one GPU over gloo, not a measurement"
        print(json.dumps(res), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
