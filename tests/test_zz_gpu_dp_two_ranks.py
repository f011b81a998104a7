"""The data-parallel Trainer with real kernels and two ranks.

Two processes share the one GPU of the test box and talk through gloo (RCCL refuses two ranks on one device), with the
collectives staged through host copies (dp.stage_collectives_through_host); everything else is the production path: gradient sinks, the three bucket all-reduces hung on the backward pass, deferred bias /
weight-gradient reductions flushed before a bucket leaves, the global-denominator loss, the fused Adam.  Expected values:
nn.DataParallel's semantics (train/main.py:196-204, :543) composed in ONE process from the library's own differentiable
modules -- each shard through the model with its own BatchNorm batch statistics, one weighted loss over both shards'
logits, gradients summed.
"""
import os
import queue

import numpy as np
import pytest
import torch

from erfnet_pytorch_amd import synthetic as S
from oracle import erfnet_oracle as O

pytestmark = pytest.mark.gpu
H, W, NC, NB = 64, 128, 20, 2


def _build(dtype):
    from erfnet_pytorch_amd import erfnet as E
    m = E.ERFNet(NC)
    m.load_state_dict(O.synthetic_state(O.erfnet_template(NC)))
    m = m.to("cuda:0").train()
    m.compute_dtype = torch.bfloat16 if dtype == "bf16" else None
    for mod in m.modules():
        if isinstance(mod, E.non_bottleneck_1d) and mod.dropout.p != 0:
            mod.__dict__["_erf_forced_mask"] = torch.ones(NB, mod.conv3x1_1.out_channels, device="cuda:0")   # dropout off: deterministic
    return m


def _shard(rank):
    x = S.synthetic_images(NB, H, W, seed=170 + rank).to("cuda:0")
    t = S.synthetic_labels(NB, H, W, NC, seed=170 + rank, ignore_frac=0.05 + 0.3 * rank).to("cuda:0")   # unequal weight sums
    return x, t


def _worker_steps_launched(rank, world, store_file, dtype, overlap, q):
    _worker(rank, world, store_file, dtype, overlap, q, graph=False, steps=3)


def _worker_steps_segments(rank, world, store_file, dtype, overlap, q):
    _worker(rank, world, store_file, dtype, overlap, q, graph="segments", steps=3)


def _worker(rank, world, store_file, dtype, overlap, q, graph=False, steps=1):
    import torch.distributed as dist
    from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
    dist.init_process_group("gloo", init_method=f"file://{store_file}", rank=rank, world_size=world)
    from erfnet_pytorch_amd.dp import stage_collectives_through_host
    # gloo's own device-tensor all-reduce stalls for tens of seconds when it is issued while this process still has kernels
    # queued and a second busy process shares the GPU (scripts/hang_probe.py: no stall without gloo, with these host-staged
    # collectives, or with the device drained first; the same stall with or without any particular kernel of the step)
    stage_collectives_through_host()
    try:
        torch.cuda.set_device(0)
        m = _build(dtype)
        tr = Trainer(m, CITYSCAPES_WEIGHTS, group=dist.group.WORLD, world_size=world, overlap=overlap, graph=graph)
        x, t = _shard(rank)
        for _ in range(steps):
            loss, _ = tr.step(x, t)
        torch.cuda.synchronize()
        assert tr.segments == (graph == "segments") and bool(tr._graphs) == bool(graph)
        q.put((rank, float(loss), tr.gflat.cpu().numpy().copy(), tr.flat.cpu().numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(world, dtype, overlap, tmp_path, worker=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    store_file = str(tmp_path / "store")
    procs = [ctx.Process(target=worker or _worker, args=(r, world, store_file, dtype, overlap, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r = q.get(timeout=90)
            got[r[0]] = r[1:]
    except queue.Empty:
        pass
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
            p.join()
    assert len(got) == world and all(p.exitcode == 0 for p in procs), "the ranks did not complete"
    return got


@pytest.mark.parametrize("dtype,overlap,tol", [("bf16", True, 8e-2), ("bf16", False, 8e-2), ("fp32", True, 1e-4), ("fp32", False, 1e-4)])
def test_two_rank_trainer_step_equals_dataparallel_semantics(dtype, overlap, tol, tmp_path):
    from erfnet_pytorch_amd.loss import CrossEntropyLoss2d
    from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
    world = 2
    got = _run(world, dtype, overlap, tmp_path)
    # replicas stay identical: same reduced gradients, same parameters after Adam, same (global) loss
    assert got[0][0] == got[1][0]
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][2], got[1][2])

    # one process: both shards through one set of parameters, per-shard BatchNorm statistics, one loss, summed gradients
    m = _build(dtype)
    outs, tgts = [], []
    for r in range(world):
        x, t = _shard(r)
        outs.append(m(x))
        tgts.append(t[:, 0])
    w = torch.tensor(CITYSCAPES_WEIGHTS, device="cuda:0")
    loss = CrossEntropyLoss2d(w)(torch.cat(outs), torch.cat(tgts))
    loss.backward()
    assert abs(got[0][0] - loss.item()) <= (1e-5 if dtype == "fp32" else 2e-3) * abs(loss.item())
    ref = Trainer(_build(dtype), CITYSCAPES_WEIGHTS)            # only for the arena's name -> offset map
    gflat = torch.from_numpy(got[0][1])
    worst = 0.0
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        o, n = ref.offsets[k]
        g = gflat[o:o + n].view_as(p)
        pg = p.grad.detach().cpu()
        if ".bias" in k and "bn" not in k and "output_conv" not in k and pg.abs().max() < 1e-3 * gflat.abs().max():
            continue                                             # conv bias in front of a batch-statistics BatchNorm: exactly 0 here, rounding noise there
        err = (g - pg).norm().item() / max(pg.norm().item(), 1e-12)
        worst = max(worst, err)
        assert err <= tol, (k, err)


def _rccl_graph_worker(rank, world, store_file, dtype, overlap, q):
    _rccl_worker(rank, world, store_file, dtype, overlap, q, graph=True, steps=4)


def _rccl_segments_worker(rank, world, store_file, dtype, overlap, q):
    _rccl_worker(rank, world, store_file, dtype, overlap, q, graph="segments", steps=5)


def _rccl_worker(rank, world, store_file, dtype, overlap, q, graph=False, steps=2):
    """One rank, backend nccl (= RCCL): the production collective path on a single GPU."""
    import torch.distributed as dist
    from erfnet_pytorch_amd.trainer import Trainer, CITYSCAPES_WEIGHTS
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"file://{store_file}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x, t = _shard(0)
        res = []
        for group in (dist.group.WORLD, None):
            tr = Trainer(_build(dtype), CITYSCAPES_WEIGHTS, group=group, world_size=1, overlap=overlap, force_collectives=True,
                         graph=graph if group is not None else False)
            assert tr.collective == (group is not None) and tr.overlap == (overlap and group is not None)
            losses = []
            for _ in range(steps):
                loss, _ = tr.step(x, t)
                losses.append(float(loss))
            torch.cuda.synchronize()
            assert bool(tr._graphs) == bool(graph and group is not None)
            assert tr.segments == (graph == "segments" and group is not None)
            res.append((losses, tr.gflat.cpu().numpy().copy(), tr.flat.cpu().numpy().copy()))
        q.put((0, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,overlap", [("bf16", True), ("bf16", False), ("fp32", True)])
def test_collective_path_on_rccl_with_one_rank_is_bit_identical_to_no_group(dtype, overlap, tmp_path):
    """The asynchronous device-tensor all-reduces of the production path (three gradient buckets launched from the autograd
    thread behind the kernels of the raw current stream, the 2-float loss all-reduce, Work.wait() before Adam) on RCCL with a
    one-rank group: sums over one rank change nothing, so two optimisation steps must match the group-less run bit for bit."""
    got = _run(1, dtype, overlap, tmp_path, worker=_rccl_worker)
    (l_a, g_a, p_a), (l_b, g_b, p_b) = got[0][0]
    assert l_a == l_b
    assert np.array_equal(g_a, g_b) and np.array_equal(p_a, p_b)


@pytest.mark.parametrize("dtype,overlap", [("bf16", True), ("bf16", False)])
def test_captured_step_with_rccl_collectives_is_bit_identical_to_the_launched_one(dtype, overlap, tmp_path):
    """Trainer(graph=True) over a one-rank RCCL group: the loss all-reduce and the three asynchronous gradient-bucket
    all-reduces are captured into the HIP graph with the kernels; four steps (two of them replays) must match the
    group-less, kernel-by-kernel run bit for bit."""
    got = _run(1, dtype, overlap, tmp_path, worker=_rccl_graph_worker)
    (l_a, g_a, p_a), (l_b, g_b, p_b) = got[0][0]
    assert l_a == l_b
    assert np.array_equal(g_a, g_b) and np.array_equal(p_a, p_b)


@pytest.mark.parametrize("dtype,overlap", [("bf16", True), ("fp32", False)])
def test_step_in_three_graphs_with_rccl_collectives_between_them_is_bit_identical_to_the_launched_one(dtype, overlap, tmp_path):
    """Trainer(graph="segments") over a one-rank RCCL group (what bench.py runs at N > 1): forward + local loss sums, backward and
    Adam are three HIP graphs, the loss all-reduce and the gradient-bucket all-reduces are issued between their replays and are
    never captured; five steps (three of them replays) must match the group-less, kernel-by-kernel run bit for bit."""
    got = _run(1, dtype, overlap, tmp_path, worker=_rccl_segments_worker)
    (l_a, g_a, p_a), (l_b, g_b, p_b) = got[0][0]
    assert l_a == l_b
    assert np.array_equal(g_a, g_b) and np.array_equal(p_a, p_b)


def test_two_rank_step_in_three_graphs_equals_the_launched_two_rank_step(tmp_path):
    """Two ranks (gloo rehearsal on one GPU), three optimisation steps each: Trainer(graph="segments") -- eager first step, capture,
    one replay, collectives between the graphs -- against the kernel-by-kernel two-rank run: loss, reduced gradients and parameters
    bit for bit on both ranks."""
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    seg = _run(2, "bf16", True, tmp_path / "a", worker=_worker_steps_segments)
    ref = _run(2, "bf16", True, tmp_path / "b", worker=_worker_steps_launched)
    for r in range(2):
        assert seg[r][0] == ref[r][0]
        assert np.array_equal(seg[r][1], ref[r][1]) and np.array_equal(seg[r][2], ref[r][2])
