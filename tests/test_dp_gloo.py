"""Data-parallel step logic on CPU, world_size 2, gloo backend.

The compute engine here is the CPU oracle (the GPU kernels cannot run in this container); what is
under test is the distributed algorithm the Trainer uses on RCCL: the global-denominator loss
(dp.global_weighted_mean), the flat gradient arena with its bucket ranges (dp.bucket_ranges) and the
sum all-reduce.  Expected values: nn.DataParallel's semantics emulated in ONE process -- each shard
through the model with its own BatchNorm batch statistics, one loss over the concatenated logits
(train/main.py:196-204, :543), gradients summed.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from erfnet_pytorch_amd import dp, synthetic as S
from oracle import erfnet_oracle as O

H, W, NC = 32, 64, 20


# Everything runs in float64: end-to-end fp32 gradients of this deep training-mode graph move by ~1e-2
# with the summation order alone (thread count), which would mask the exact identity checked here.
def _state():
    sd = O.synthetic_state(O.erfnet_template(NC))
    return {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}


def _shard_inputs(rank):
    x = S.synthetic_images(1, H, W, seed=70 + rank).double()
    t = S.synthetic_labels(1, H, W, NC, seed=70 + rank, ignore_frac=0.05 + 0.3 * rank)[:, 0]   # unequal weight sums
    return x, t


def _masks(sd, rank):
    out = {}
    for i, (kind, cin, _cout, p, _d) in enumerate(O.ENCODER_LAYERS):
        if kind == "nb" and p > 0:
            u = S.unit_uniform(cin, S.key_seed(f"mask{i}", rank))
            out[f"encoder.layers.{i}"] = torch.from_numpy(((u > 2 * p - 1) / (1 - p)).astype("float64")).reshape(1, cin, 1, 1)
    return out


def _leaf(sd):
    return {k: (v.clone().requires_grad_(True) if (v.dtype.is_floating_point and "running" not in k) else v) for k, v in sd.items()}


def _worker(rank, world, store_file, q):
    torch.set_num_threads(2)
    # rendezvous through a file store: no TCP port has to be guessed free in advance, so there is no bind race to retry
    dist.init_process_group("gloo", init_method=f"file://{store_file}", rank=rank, world_size=world)
    try:
        sd = _state()
        w = torch.tensor(O.CITYSCAPES_WEIGHTS, dtype=torch.float64)
        leaf = _leaf(sd)
        x, t = _shard_inputs(rank)
        st = O.RunState(training=True, dropout_masks=_masks(sd, rank))
        logits = O.erfnet(leaf, x, st)
        num, den = O.weighted_ce_parts(logits, t, w)
        gnum, gden = dp.global_weighted_mean(num.detach(), den.detach(), dist.group.WORLD)
        # every rank back-propagates its own numerator over the GLOBAL denominator
        (num / gden).backward()
        names = [k for k, v in leaf.items() if v.requires_grad]
        offsets, total = dp.arena_offsets([(k, leaf[k].shape) for k in names])
        buckets, active = dp.bucket_ranges(offsets)
        arena = torch.zeros(total, dtype=torch.float64)
        for k in names:
            if leaf[k].grad is not None:
                o, n = offsets[k]
                arena[o:o + n] = leaf[k].grad.reshape(-1)
        for lo, hi in buckets:                      # decoder first, as the backward pass finishes them
            dist.all_reduce(arena[lo:hi])
        if rank == 0:
            # plain numpy in the queue: a torch tensor travels as a file descriptor that the parent has to fetch from this
            # process, which fails when the worker has already exited
            q.put({"loss": float(gnum / gden), "arena": arena.numpy().copy(), "names": names, "buckets": buckets, "active": active,
                   "offsets": offsets})
    finally:
        dist.destroy_process_group()


def _run_world(world, tmp_path):
    """Spawn `world` gloo ranks and return rank 0's result.  Any worker that exits non-zero, or does not finish in time,
    fails the test: nothing is retried."""
    import queue
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    store_file = str(tmp_path / "gloo_store")
    procs = [ctx.Process(target=_worker, args=(r, world, store_file, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = q.get(timeout=300)
    except queue.Empty:
        got = None
    for p in procs:
        p.join(timeout=180)
        if p.is_alive():
            p.kill()
            p.join()
            pytest.fail("a gloo rank did not finish in time")
    codes = [p.exitcode for p in procs]
    assert codes == [0] * world, f"worker exit codes {codes}"
    assert got is not None, "rank 0 sent no result"
    return got


def test_two_rank_step_equals_dataparallel_semantics(tmp_path):
    world = 2
    got = _run_world(world, tmp_path)
    got["arena"] = torch.from_numpy(got["arena"])

    # single-process emulation of nn.DataParallel
    sd = _state()
    w = torch.tensor(O.CITYSCAPES_WEIGHTS, dtype=torch.float64)
    leaf = _leaf(sd)
    outs, tgts = [], []
    for r in range(world):
        x, t = _shard_inputs(r)
        outs.append(O.erfnet(leaf, x, O.RunState(training=True, dropout_masks=_masks(sd, r))))   # per-replica BN statistics
        tgts.append(t)
    loss = O.weighted_ce(torch.cat(outs), torch.cat(tgts), w)                                       # loss over the gathered batch
    loss.backward()
    assert abs(got["loss"] - loss.item()) <= 1e-10 * abs(loss.item())
    # a mean of per-rank means would NOT match (the shards' weight sums differ)
    per_rank = [O.weighted_ce(o.detach(), t, w).item() for o, t in zip(outs, tgts)]
    assert abs(sum(per_rank) / world - loss.item()) > 1e-4 * abs(loss.item())

    offsets = got["offsets"]
    for k in got["names"]:
        o, n = offsets[k]
        mine = got["arena"][o:o + n]
        if leaf[k].grad is None:
            assert k.startswith("encoder.output_conv") and float(mine.abs().sum()) == 0.0
            continue
        ref = leaf[k].grad.reshape(-1)
        scale = max(ref.abs().max().item(), 1e-3)
        assert (mine - ref).abs().max().item() <= 1e-8 * scale, k
    # buckets tile the arena except for the unused encoder head; the optimiser never touches it
    lo_head, n_head = offsets["encoder.output_conv.weight"][0], sum(offsets[k][1] for k in offsets if k.startswith("encoder.output_conv"))
    covered = sorted(got["buckets"])
    assert covered[0][0] == 0 and covered[0][1] == covered[1][0] and covered[1][1] == lo_head
    assert covered[2][0] == lo_head + n_head and covered[2][1] == got["arena"].numel()
    assert all(not (lo < lo_head + n_head and hi > lo_head) for lo, hi in got["active"])
