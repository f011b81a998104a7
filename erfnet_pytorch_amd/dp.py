"""Backend-independent pieces of the data-parallel step (no kernels, no GPU needed).

nn.DataParallel (train/main.py:543) computes ONE loss over the gathered global batch and sums
the replicas' gradients.  With one process per GPU the same result needs two rules, both
implemented here so that they can be exercised on CPU with the gloo backend:

  global_weighted_mean : loss = sum_r num_r / sum_r den_r  (not the mean of per-rank means:
                         the ranks' weight sums differ whenever their label mixes differ)
  bucket_ranges        : contiguous slices of the flat gradient arena in the order the backward
                         pass completes them (decoder, 128-channel encoder stage, the rest), plus
                         the slices the optimiser may touch (everything that took part in the
                         forward; `encoder.output_conv` did not).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch


def global_weighted_mean(num: torch.Tensor, den: torch.Tensor, group=None):
    """(sum_r num_r, sum_r den_r) over the process group; the loss is their quotient and every
    rank scales its dlogits by 1 / sum_r den_r."""
    if group is None:
        return num, den
    import torch.distributed as dist
    parts = torch.stack([num.reshape(()), den.reshape(())]).clone()
    dist.all_reduce(parts, group=group)
    return parts[0], parts[1]


def arena_offsets(named_shapes) -> Tuple[Dict[str, Tuple[int, int]], int]:
    """name -> (offset, numel) in registration order, and the total element count."""
    offsets, off = {}, 0
    for name, shape in named_shapes:
        n = 1
        for s in shape:
            n *= int(s)
        offsets[name] = (off, n)
        off += n
    return offsets, off


def _span(offsets, prefixes):
    keys = [k for k in offsets if any(k.startswith(p) for p in prefixes)]
    lo = min(offsets[k][0] for k in keys)
    hi = max(offsets[k][0] + offsets[k][1] for k in keys)
    return lo, hi


def bucket_ranges(offsets, only_encode: bool = False) -> Tuple[List[Tuple[int, int]], List[Tuple[int, int]]]:
    """(all-reduce buckets in backward-completion order, optimiser ranges) for an ERFNet arena.
    `only_encode`: the encoder-only pass of train/main.py:203 -- the decoder takes no part, the 1x1 head does."""
    decoder = _span(offsets, ["decoder."])
    enc128 = _span(offsets, [f"encoder.layers.{i}." for i in range(6, 15)])
    enc_lo = _span(offsets, ["encoder.initial_block."] + [f"encoder.layers.{i}." for i in range(0, 6)])
    head = _span(offsets, ["encoder.output_conv."])
    if not (enc_lo[1] == enc128[0] and enc128[1] == head[0] and head[1] == decoder[0]):
        raise ValueError("parameter registration order differs from ERFNet's")
    if only_encode:
        return [(enc128[0], head[1]), enc_lo], [(enc_lo[0], head[1])]
    return [decoder, enc128, enc_lo], [(enc_lo[0], enc128[1]), decoder]


def stage_collectives_through_host():
    """Rehearsal / test aid for boxes where several ranks share ONE GPU (gloo backend): torch.distributed.all_reduce on a
    device tensor is replaced by copy-to-host, all-reduce, copy-back, done synchronously by the caller.  The algorithm
    around the collectives (which ranges are reduced, when, with what result) stays what it is; only the transport changes.
    Never used on the RCCL path."""
    import torch.distributed as dist
    if getattr(dist.all_reduce, "_erf_staged", False):
        return
    orig = dist.all_reduce

    class _Done:
        def wait(self):
            return True

    def staged(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not t.is_cuda:
            return orig(t, op=op, group=group, async_op=async_op)
        h = t.detach().cpu()
        orig(h, op=op, group=group)
        t.copy_(h)
        return _Done() if async_op else None

    staged._erf_staged = True
    dist.all_reduce = staged
