"""Training step of /root/reference/train/main.py:162-208 on one process per GPU.

The reference wraps the model in nn.DataParallel (train/main.py:543): one process, batch split
over the GPUs, per-replica BatchNorm statistics, the loss taken over the gathered global batch
on GPU 0, gradients summed there, Adam on GPU 0, parameters re-broadcast every step.  Here every
rank owns a full replica and a shard of the batch:

  * parameters live in ONE flat fp32 arena (registration order), gradients in a second arena the
    backward kernels write into directly (autograd._Sink), Adam moments in two more;
  * the loss is  sum_ranks(num) / sum_ranks(den)  (loss.weighted_cross_entropy with the process
    group), so the summed gradients equal the reference's global-batch gradient;
  * the gradient arena is all-reduced (RCCL, sum) in three buckets that complete in backward order
    -- decoder, 128-channel encoder stage, the rest -- each launched as soon as its last block's
    backward is enqueued, overlapping the remaining backward; BatchNorm stays per-GPU;
  * one fused Adam launch per active parameter range; parameters that took no part in the forward
    (`encoder.output_conv` in the full network) are left untouched, like torch.optim.Adam skipping
    `grad is None` (no weight decay on them either);
  * replicas stay bit-identical because every rank applies the same reduced gradient.
"""
from __future__ import annotations

import math
import time

import torch

from . import autograd as ag
from . import dp
from . import erfnet as E
from ._lib import call
from .loss import weighted_cross_entropy

# train/main.py:92-111 (20-class Cityscapes recipe; class 19 = ignore)
CITYSCAPES_WEIGHTS = [
    2.8149201869965, 6.9850029945374, 3.7890393733978, 9.9428062438965, 9.7702074050903,
    9.5110931396484, 10.311357498169, 10.026463508606, 4.6323022842407, 9.5608062744141,
    7.8698215484619, 9.5168733596802, 10.373730659485, 6.6616044044495, 10.260489463806,
    10.287888526917, 10.289801597595, 10.405355453491, 10.138095855713, 0.0,
]


def poly_lr(base_lr, epoch, num_epochs):
    """lr schedule of train/main.py:156-157 (epoch is 1-based, stepped at :165)."""
    return base_lr * math.pow(1 - ((epoch - 1) / num_epochs), 0.9)


def _take_over_batch_counters(model):
    """BatchNorm layers append their num_batches_tracked to the returned list instead of launching one add each."""
    pending = []
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.__dict__["_erf_batch_counter"] = pending
    return pending


def _flush_batch_counters(pending):
    if pending:
        torch._foreach_add_(list(pending), 1)
        pending.clear()


def _prepare_masks(model, images, state=None):
    """All Dropout2d masks of the coming forward pass by one launch (the blocks with p > 0, in forward order)."""
    if model.training:
        mods = [m for m in model.modules() if isinstance(m, E.non_bottleneck_1d) and m.dropout.p != 0]
        ag.prepare_dropout_masks(mods, images.shape[0], images.device, state)


def _refresh_bank(bank, model, images):
    """(Re)build the model's PackedBank for the step's compute dtype and re-pack every factorized-conv weight."""
    from . import ops
    cd = getattr(model, "compute_dtype", None)
    code = ops.image_code(images) if cd is None else (ops.BF16 if cd == torch.bfloat16 else ops.F32)
    if bank is None or bank.code != code:
        bank = ops.PackedBank(model, code)
    bank.refresh()
    return bank


class _NoCollection:
    """No cyclic garbage collection while a capture is open: a collection that happens to free an older CUDAGraph object
    (a discarded Trainer's, say) destroys that graph in the middle of the capture, which HIP refuses ("operation not
    permitted when stream is capturing") -- from a destructor, i.e. the process aborts."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


class Trainer:
    def __init__(self, model: E.ERFNet, class_weights=None, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4,
                 group=None, world_size=1, overlap=True, only_encode=False, fused_head_loss=True, force_collectives=False,
                 graph=False, relabel=None):
        """`relabel` = (olabel, nlabel): uint8 targets (the decoded label images) become int64 class indices on the device
        with olabel mapped to nlabel -- ToLabel() + Relabel of the loaders (train/main.py:65-66, train/transform.py:53-68);
        without it uint8 targets are converted only.
        `only_encode`: the first stage of the reference's recipe (train/main.py:495 `train(args, model, True)`,
        :203 `model(inputs, only_encode=enc)`): logits from Encoder.output_conv at 1/8 resolution against targets of
        that size; the decoder's parameters take no part and are left untouched.
        `force_collectives`: issue the loss and gradient-bucket all-reduces even when world_size is 1 (a one-rank process
        group exercises the whole collective path -- hooks, side-stream ordering, waits -- on a single GPU).
        `graph`: step() captures the whole optimisation step (zero_grad .. Adam, collectives included) into a HIP graph
        the second time it sees an input shape and replays it from then on; the inputs are copied into the graph's
        static buffers.  graph="segments": the step is cut at its exchange points into five graphs (forward + local
        loss sums | head + decoder backward | 128-channel encoder stage backward | remaining backward | Adam); the loss
        all-reduce is issued between the first two, and each gradient bucket's asynchronous all-reduce right after the
        replay of the section that completes it, so that it runs under the next section's kernels (`overlap=False`: all
        three after the last section); Adam waits for them.  No collective is ever captured, and the host enqueues five
        graph launches instead of ~370 kernels.  In graph modes the returned loss is a copy, the returned logits (if
        any) alias a static buffer the next step overwrites.  Everything that changes from step to step lives on the device for this (dropout seed counter,
        Adam's step count and learning rate), graph or not, so both ways of running give the same bits."""
        self.model = model
        self.relabel = None if relabel is None else (int(relabel[0]), int(relabel[1]))
        self.only_encode = only_encode
        # Decoder.output_conv + loss in one pass each way (the logits stay in registers); step() then returns
        # (loss, None).  Set False to get the logits back, e.g. for an IoU over the training batches.
        self.fused_head_loss = fused_head_loss and not only_encode
        self.group, self.world = group, world_size
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self._step_dev = None
        self._step_host = 0
        self.collective = world_size > 1 or (force_collectives and group is not None)
        self.overlap = overlap and self.collective
        self.mute_collectives = False                   # timing aid (segments mode): replay the graphs, issue no all-reduce
        dev = next(model.parameters()).device
        self.dev = dev
        self.weight = None if class_weights is None else torch.tensor(class_weights, dtype=torch.float32, device=dev)

        named = list(model.named_parameters())
        total = sum(p.numel() for _, p in named)
        self._epoch = [0]                               # zero_grad() calls so far (see autograd._Sink)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = {}
        off = 0
        for name, p in named:
            n = p.numel()
            self.flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            self.offsets[name] = (off, n)
            off += n

        # gradient sinks: every block's kernels write into views of gflat
        for mname, mod in model.named_modules():
            own = {k: v for k, v in mod.named_parameters(recurse=True)} if isinstance(
                mod, (E.DownsamplerBlock, E.UpsamplerBlock, E.non_bottleneck_1d)) else None
            if own is None and (mod is model.decoder.output_conv or mod is model.encoder.output_conv):
                own = dict(mod.named_parameters())
            if own is None:
                continue
            views = {}
            for k, prm in own.items():
                o, n = self.offsets[f"{mname}.{k}"]
                views[k] = self.gflat[o:o + n].view_as(prm)
            mod.__dict__["_erf_grad_sink"] = views
            mod.__dict__["_erf_grad_sink_zeroed"] = True    # step() clears the arena before every backward pass
            mod.__dict__["_erf_grad_sink_epoch"] = self._epoch

        # per-step scalars on the device: a captured step must not depend on host values that change
        self._seed_state = ag.new_seed_state(dev)
        self._step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._lr_dev = torch.full((1,), float(lr), dtype=torch.float64, device=dev)
        self._lr_on_dev = float(lr)
        self._hyper = torch.zeros(2, dtype=torch.float32, device=dev)
        self.use_graph = bool(graph)
        self.segments = graph == "segments" and self.collective and self.fused_head_loss
        self._graphs = {}                               # input signature -> (graph, static images, static targets, loss, outputs)
        self._seen = set()
        self._bank = None                               # packed factorized-conv weights, refreshed once per step
        self._counted = _take_over_batch_counters(model)
        # all-reduce buckets (backward-completion order) and the ranges Adam may touch
        self.buckets, self.active = dp.bucket_ranges(self.offsets, only_encode)
        self._pending = []
        if self.overlap:
            last = len(self.buckets) - 1
            if not only_encode:
                model.decoder.layers[0].__dict__["_erf_after_backward"] = lambda: self._bucket_hook(0)
            model.encoder.layers[6].__dict__["_erf_after_backward"] = lambda: self._bucket_hook(last - 1)
            model.encoder.initial_block.__dict__["_erf_after_backward"] = lambda: self._bucket_hook(last)

    # -- Adam's step count: the device copy is what the kernels read, the host copy mirrors it --------
    @property
    def step_count(self):
        return self._step_host

    @step_count.setter
    def step_count(self, value):
        self._step_host = int(value)
        if getattr(self, "_step_dev", None) is not None:
            self._step_dev.fill_(int(value))

    def _sync_lr(self):
        if self.lr != self._lr_on_dev:                                # set_epoch() / a resumed optimiser changed it
            self._lr_dev.fill_(float(self.lr))
            self._lr_on_dev = float(self.lr)

    # -- collectives ------------------------------------------------------------------------
    def _bucket_hook(self, i):
        if self.overlap:                                              # off while a backward pass is captured without collectives
            self._launch_bucket(i)

    def _launch_bucket(self, i):
        import torch.distributed as dist
        lo, hi = self.buckets[i]
        self._pending.append(dist.all_reduce(self.gflat[lo:hi], group=self.group, async_op=True))

    def _reduce_gradients(self):
        if not self.collective:
            return
        if not self.overlap:
            for i in range(len(self.buckets)):
                self._launch_bucket(i)
        for w in self._pending:
            w.wait()
        self._pending = []

    def time_buckets(self, reps=5, barrier=lambda: None, reduce_max=lambda v: v):
        """Each gradient bucket's all-reduce ALONE (nothing to overlap with), mean of `reps` after one warm-up, slowest rank:
        what the collective costs when it is fully exposed.  Sums garbage into the gradient arena: call between steps."""
        import torch.distributed as dist
        n = dist.get_world_size(self.group)
        out = []
        for lo, hi in self.buckets:
            buf = self.gflat[lo:hi]
            dist.all_reduce(buf, group=self.group)
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                dist.all_reduce(buf, group=self.group)
            torch.cuda.synchronize()
            ms = reduce_max((time.perf_counter() - t0) * 1e3 / reps)
            nbytes = (hi - lo) * 4
            out.append({"elements": hi - lo, "bytes": nbytes, "ms": ms, "algbw_GBs": nbytes / ms / 1e6,
                        "busbw_GBs": nbytes / ms / 1e6 * 2 * (n - 1) / n})
        self.gflat.zero_()
        return {"buckets": out, "order": "backward completion: decoder, 128-channel encoder stage, the rest",
                "overlapped_with_backward": self.overlap, "ranks": n}

    # -- one optimisation step ----------------------------------------------------------------
    def set_epoch(self, epoch, num_epochs, base_lr=5e-4):
        self.lr = poly_lr(base_lr, epoch, num_epochs)

    def zero_grad(self):
        self._epoch[0] += 1
        self.gflat.zero_()

    def step(self, images, targets):
        """images [N,3,H,W] (fp32 / bf16, or uint8 [N,3,H,W] / [N,H,W,3]: ToTensor()'s / 255 is then applied by the initial
        block); targets [N,1,H,W] or [N,H,W] int64, or uint8 (see `relabel`).  Returns (global loss tensor, logits or None).  In graph modes the loss is a
        copy; returned logits alias a static buffer that the next step overwrites (clone them to keep them)."""
        if targets.dim() == 4:
            targets = targets[:, 0]                                   # criterion(outputs, targets[:, 0]), train/main.py:202
        if targets.dtype == torch.uint8:
            from . import ops
            targets = ops.labels_from_u8(targets, self.relabel)
        self._sync_lr()
        if not self.use_graph or self._host_masks():
            return self._step_eager(images, targets)
        # betas, eps, weight decay and the class weights are constants of the captured launches: part of the key, and graphs
        # captured with other values are dropped (their private pools hold a whole step's activations)
        hyper = (tuple(self.betas), float(self.eps), float(self.wd), None if self.weight is None else self.weight.data_ptr())
        key = (tuple(images.shape), images.dtype, tuple(targets.shape), getattr(self.model, "compute_dtype", None), hyper)
        for stale in [k for k in self._graphs if k[4] != hyper]:
            del self._graphs[stale]
        entry = self._graphs.get(key)
        if entry is None:
            if key not in self._seen:                                 # first sight of a shape: run it eagerly (sizes caches, workspaces)
                self._seen.add(key)
                return self._step_eager(images, targets)
            entry = self._capture_segments(key, images, targets) if self.segments else self._capture(key, images, targets)
        graph, gx, gt, loss, outputs = entry
        gx.copy_(images)
        gt.copy_(targets)
        if self.segments:
            import torch.distributed as dist
            g1, sections, g3, sums = graph
            quiet = self.mute_collectives
            g1.replay()
            if not quiet:
                dist.all_reduce(sums[:2], group=self.group)           # (sum w*nll, sum w) over the global batch
            for i, g2 in enumerate(sections):
                g2.replay()
                if self.overlap and not quiet:                        # bucket i is complete: it travels under section i + 1
                    self._launch_bucket(i)
            if not quiet:
                self._reduce_gradients()                              # (all buckets now if not overlapped) + wait
            g3.replay()
        else:
            graph.replay()
        self._after_replayed_step()
        return loss.clone(), outputs

    def _host_masks(self):
        """Forced dropout masks (a test hook) that live in host memory would be uploaded inside the step: not capturable."""
        for m in self.model.modules():
            f = m.__dict__.get("_erf_forced_mask")
            if f is not None and (f.device != self.dev or f.dtype != torch.float32):
                return True
        return False

    def _capture(self, key, images, targets):
        gx, gt = images.clone(), targets.clone()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # everything the step changed on the HOST during capture is undone below: the capture launches nothing
        host = (self._step_host, self._epoch[0])
        # thread_local: only this thread is held to the capture rules.  With the default (global) mode an event query by
        # another thread while the capture is open -- the process group's watchdog polling the work of an earlier step --
        # fails with "operation not permitted when stream is capturing" and takes the process down.
        with _NoCollection(), torch.cuda.graph(graph, capture_error_mode="thread_local"):   # anything that cannot be captured raises here
            loss, outputs = self._step_eager(gx, gt)
        self._step_host, self._epoch[0] = host
        entry = (graph, gx, gt, loss, outputs)
        self._graphs[key] = entry
        return entry

    def _capture_segments(self, key, images, targets):
        """Five graphs around the exchange points of a step; see __init__.  The capture itself launches nothing and
        issues no collective (the first, eager, step of this shape has already run them once).  The backward pass is cut
        where the gradient buckets end (autograd.erfnet_backward_sections), one graph per bucket."""
        gx, gt = images.clone(), targets.clone()
        torch.cuda.synchronize()
        host = (self._step_host, self._epoch[0])
        overlap, self.overlap = self.overlap, False                   # the bucket hooks stay silent: buckets go out between the graphs
        g1, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        sections = [torch.cuda.CUDAGraph() for _ in self.buckets]
        try:
            with _NoCollection():
                with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                    self.zero_grad()
                    self._refresh_packed(gx)
                    _prepare_masks(self.model, gx, self._seed_state)
                    feats, sums, state = ag.erfnet_forward_sums(self.model, gx, gt, self.weight, cuts=True)
                    _flush_batch_counters(self._counted)
                walk = ag.erfnet_backward_sections(feats, sums, state)
                loss = None
                for i, g2 in enumerate(sections):
                    with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
                        got = next(walk, None)
                        if i == 0:
                            loss = got.detach()
                if next(walk, "done") != "done":
                    raise RuntimeError("the backward pass has more sections than gradient buckets")
                with torch.cuda.graph(g3, pool=g1.pool(), capture_error_mode="thread_local"):
                    self.optimizer_step()
        finally:
            self.overlap = overlap
        self._step_host, self._epoch[0] = host
        entry = ((g1, sections, g3, sums), gx, gt, loss, None)
        self._graphs[key] = entry
        return entry

    def _after_replayed_step(self):
        """Host-side mirrors of what a replayed step changed on the device."""
        from . import ops
        self._epoch[0] += 1
        self._step_host += 1
        ops.bump_param_epoch()

    def _step_eager(self, images, targets):
        self.zero_grad()                                              # optimizer.zero_grad(), train/main.py:200
        self._refresh_packed(images)
        _prepare_masks(self.model, images, self._seed_state)
        group = self.group if self.collective else None
        if self.fused_head_loss:
            from . import ops
            ops.require_gpu(images, "Trainer")
            outputs = None
            loss = ag.erfnet_forward_loss(self.model, images, targets, self.weight, group)
            _flush_batch_counters(self._counted)
        else:
            outputs = self.model(images, only_encode=True) if self.only_encode else self.model(images)
            _flush_batch_counters(self._counted)
            loss = weighted_cross_entropy(outputs, targets, self.weight, group)
        loss.backward()
        ag.flush_bias_sums()
        self._reduce_gradients()
        self.optimizer_step()
        return loss.detach(), None if outputs is None else outputs.detach()

    def _refresh_packed(self, images):
        self._bank = _refresh_bank(self._bank, self.model, images)

    # -- checkpoint / resume (train/main.py:143-153 loads, :320-326 saves 'optimizer': optimizer.state_dict()) --------
    def _is_active(self, name):
        o, _ = self.offsets[name]
        return any(lo <= o < hi for lo, hi in self.active)

    def optimizer_state_dict(self):
        """The arena's Adam state in torch.optim.Adam.state_dict() form (parameter index = registration order), so a
        checkpoint written here resumes under the reference's optimiser and vice versa.  Parameters that never took
        part in a step carry no state, exactly like torch's lazily created per-parameter state."""
        names = [k for k, _ in self.model.named_parameters()]
        params = dict(self.model.named_parameters())
        state = {}
        if self.step_count > 0:
            for i, k in enumerate(names):
                if not self._is_active(k):
                    continue
                o, n = self.offsets[k]
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + n].view_as(params[k]).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(params[k]).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        names = [k for k, _ in self.model.named_parameters()]
        group = sd["param_groups"][0]
        if len(sd["param_groups"]) != 1 or list(group["params"]) != list(range(len(names))):
            raise ValueError("optimizer state does not describe this model's parameter list")
        if group.get("amsgrad") or group.get("maximize"):
            raise ValueError("amsgrad / maximize are not part of the reference recipe and are not supported")
        self.lr, self.betas, self.eps, self.wd = group["lr"], tuple(group["betas"]), group["eps"], group["weight_decay"]
        self._graphs.clear()                                          # captured steps hold the old betas / eps / weight decay
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, st in sd["state"].items():
            k = names[int(i)]
            if not self._is_active(k):
                raise ValueError(f"optimizer state for {k}, which this training stage does not update")
            o, n = self.offsets[k]
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; the arena keeps one step counter")
        self.step_count = steps.pop() if steps else 0

    def optimizer_step(self):
        from . import ops
        ops.bump_param_epoch()                          # parameters change below through raw pointers
        self._sync_lr()
        self._step_host += 1
        st = ops._stream()
        # the step count and the learning rate are read from device memory (see __init__): count += 1, then the two
        # step-dependent factors, then the update of every active range
        call("erf_adam_advance", self._step_dev.data_ptr(), self._lr_dev.data_ptr(), self._hyper.data_ptr(),
             float(self.betas[0]), float(self.betas[1]), st)
        for lo, hi in self.active:
            call("erf_adam_fused_arena_hyper", self.flat[lo:hi].data_ptr(), self.gflat[lo:hi].data_ptr(),
                 self.exp_avg[lo:hi].data_ptr(), self.exp_avg_sq[lo:hi].data_ptr(), hi - lo, self._hyper.data_ptr(),
                 float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd), 1.0, st)


class ArenaSGD:
    """Training step of the ImageNet pre-training recipe (imagenet/main.py:81-85 of the reference:
    CrossEntropyLoss + SGD(lr, momentum 0.9, weight_decay 1e-4)) for any model built from this package's blocks.

    Same arrangement as Trainer: parameters, gradients and momentum buffers live in flat fp32 arenas; the
    convolution blocks' backward kernels write into the gradient arena directly, every other parameter's
    `.grad` is a view of it; with world_size > 1 the loss is the global-batch mean and the arena is summed
    over the ranks (RCCL) in one all-reduce before one fused SGD launch."""

    def __init__(self, model, lr=0.1, momentum=0.9, weight_decay=1e-4, group=None, world_size=1, graph=False):
        """`graph` (one GPU): from the third step with one input shape on, the whole step is replayed from a HIP graph (the
        step is launch-bound at 224x224).  The Dropout2d seed counter, the learning rate and the weight decay live on the
        device in both modes (imagenet/main.py:276-286 changes the last two from epoch to epoch: assign `.lr` / `.wd`
        and the next step, replayed or launched, uses the new values); the momentum is a constant of the captured launch
        and part of the graph key."""
        self.model, self.lr, self.momentum, self.wd = model, lr, momentum, weight_decay
        self.group, self.world = group, world_size
        self.step_count = 0
        self._epoch = [0]
        named = list(model.named_parameters())
        dev = named[0][1].device
        self.use_graph = bool(graph) and world_size == 1
        self._graphs, self._seen = {}, set()
        self._seed_state = ag.new_seed_state(dev)
        self._hyper = torch.tensor([lr, weight_decay], dtype=torch.float32, device=dev)
        self._hyper_host = (float(lr), float(weight_decay))
        total = sum(p.numel() for _, p in named)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buf = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets, off = {}, 0
        for name, p in named:
            n = p.numel()
            self.flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            self.offsets[name] = (off, n)
            off += n
        sunk = set()
        for mname, mod in model.named_modules():
            if not isinstance(mod, (E.DownsamplerBlock, E.UpsamplerBlock, E.non_bottleneck_1d)):
                continue
            views = {}
            for k, prm in mod.named_parameters(recurse=True):
                o, n = self.offsets[f"{mname}.{k}"]
                views[k] = self.gflat[o:o + n].view_as(prm)
                sunk.add(f"{mname}.{k}")
            mod.__dict__["_erf_grad_sink"] = views
            mod.__dict__["_erf_grad_sink_zeroed"] = True    # step() clears the arena before every backward pass
            mod.__dict__["_erf_grad_sink_epoch"] = self._epoch
        self._views = [(p, self.gflat[o:o + n].view_as(p)) for (name, p), (o, n) in
                       ((np_, self.offsets[np_[0]]) for np_ in named) if name not in sunk]

    def zero_grad(self):
        self._epoch[0] += 1
        self.gflat.zero_()
        for p, v in self._views:
            p.grad = v                                  # autograd accumulates in place into the arena

    def _sync_hyper(self):
        now = (float(self.lr), float(self.wd))
        if now != self._hyper_host:                                   # the schedule moved lr / wd: one small upload, no new capture
            self._hyper.copy_(torch.tensor(now, dtype=torch.float32))
            self._hyper_host = now

    def step(self, images, targets):
        self._sync_hyper()
        if not self.use_graph or self.step_count == 0:                # the first step of a run also initialises the momentum buffers
            return self._step_eager(images, targets)
        key = (tuple(images.shape), images.dtype, tuple(targets.shape), getattr(self.model, "compute_dtype", None), float(self.momentum))
        entry = self._graphs.get(key)
        if entry is None:
            if key not in self._seen:                                 # first sight: run it eagerly (sizes caches, workspaces)
                self._seen.add(key)
                return self._step_eager(images, targets)
            gx, gt = images.clone(), targets.clone()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            host = (self.step_count, self._epoch[0])
            with _NoCollection(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                loss, outputs = self._step_eager(gx, gt)
            self.step_count, self._epoch[0] = host
            entry = self._graphs[key] = (graph, gx, gt, loss, outputs)
        graph, gx, gt, loss, outputs = entry
        gx.copy_(images)
        gt.copy_(targets)
        graph.replay()
        from . import ops
        self._epoch[0] += 1
        self.step_count += 1
        ops.bump_param_epoch()
        return loss.clone(), outputs                                  # the logits alias a static buffer the next replay overwrites

    def _step_eager(self, images, targets):
        self.zero_grad()
        self._bank = _refresh_bank(getattr(self, "_bank", None), self.model, images)
        _prepare_masks(self.model, images, self._seed_state)
        outputs = self.model(images)
        loss = weighted_cross_entropy(outputs, targets, None, self.group if self.world > 1 else None)
        loss.backward()
        ag.flush_bias_sums()
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.gflat, group=self.group)
        from . import ops
        ops.bump_param_epoch()
        call("erf_sgd_fused_arena_hyper", self.flat.data_ptr(), self.gflat.data_ptr(), self.buf.data_ptr(), self.flat.numel(),
             self._hyper.data_ptr(), float(self.momentum), int(self.step_count == 0), 1.0, ops._stream())
        self.step_count += 1
        return loss.detach(), outputs.detach()


# ------------------------------------------------------------------------------------------ benchmark leg
def bench_train(args, world, rank, dev, barrier, max_over_ranks):
    """BASELINE.json configs[2]: ERFNet(20) training, batch 6 per GPU, 3x512x1024 synthetic images,
    weighted cross-entropy (20-class weights, w[19] = 0), Adam, dropout on, data parallel over `world` ranks."""
    import torch.distributed as dist
    from . import synthetic as S
    batch = args.batch or 6
    model = E.ERFNet(20)
    S.load_synthetic(model)
    model = model.to(dev).train()
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else None
    group = dist.group.WORLD if world > 1 else None
    # One GPU: the step is replayed from one HIP graph.  Several GPUs: five graphs (forward | three backward sections | Adam)
    # with the loss all-reduce and one gradient bucket's all-reduce issued after each replay -- launched kernel by kernel the
    # step is host-bound (6.5 ms against 5.4 ms of kernels); --train-graph captures the RCCL all-reduces into one graph
    # instead (verified with a one-rank group only).
    use_graph = False
    if not getattr(args, "no_graph", False):
        use_graph = True if (world == 1 or getattr(args, "train_graph", False)) else "segments"
    tr = Trainer(model, CITYSCAPES_WEIGHTS, group=group, world_size=world, graph=use_graph)
    torch.manual_seed(1234 + rank)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(batch, 3, args.height, args.width, device=dev, generator=gen)
    t = torch.randint(0, 20, (batch, 1, args.height, args.width), device=dev, generator=gen)
    t[torch.rand(t.shape, device=dev, generator=gen) < 0.05] = 19

    def step():
        return tr.step(x, t)                  # zero_grad + forward + loss + backward + all-reduce + Adam

    for _ in range(max(args.warmup, 3)):        # the first step sizes caches, the second is captured, replays from the third
        loss, _ = step()
    torch.cuda.synchronize()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step()
    torch.cuda.synchronize()
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    flop_img = 2 * (3 * 26_612_727_808 - 46_006_272) * (args.height * args.width) / (512 * 1024)
    value = world * batch * args.steps / elapsed
    extra = {"world_size": dist.get_world_size(group) if world > 1 else 1}
    if tr.collective:
        final_loss = float(loss)
        ar = tr.time_buckets(barrier=lambda: barrier(world), reduce_max=lambda v: max_over_ranks(v, world, dev))
        if tr.segments:
            # what the collectives cost the step: the same replays with every all-reduce muted (each rank then steps on its own
            # gradients; nothing after this uses the parameters)
            tr.mute_collectives = True
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            barrier(world)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            barrier(world)
            muted = max_over_ranks(time.perf_counter() - t1, world, dev)
            tr.mute_collectives = False
            ar["step_ms_without_collectives"] = 1e3 * muted / args.steps
            ar["exposed_ms"] = 1e3 * (elapsed - muted) / args.steps
        extra["allreduce"] = ar
        loss = torch.tensor(final_loss)
    return {
        **extra,
        "metric": "img/s ERFNet 512x1024: bf16 inference (1 GPU) + training (1/2/4/8 MI355X)",
        "value": value, "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"ERFNet(20) training step (fwd + weighted CE + bwd + RCCL grad all-reduce + Adam), batch {batch}/GPU, "
                               f"3x{args.height}x{args.width}, data parallel x{world}",
                   "global_batch": batch * world, "final_loss": float(loss),
                   "hip_graph": ("segments" if tr.segments else True) if tr._graphs else False},
        "model_tflops": value * flop_img / 1e12,
    }


def bench_imagenet(args, world, rank, dev, barrier, max_over_ranks):
    """BASELINE.json configs[3]: erfnet_imagenet.ERFNet(1000) training, batch 32 per GPU, 3x224x224 synthetic images,
    CrossEntropyLoss, SGD(0.1, 0.9, 1e-4), data parallel over `world` ranks."""
    import torch.distributed as dist
    from . import erfnet_imagenet as EI, synthetic as S
    batch = args.batch or 32
    model = EI.ERFNet(1000)
    S.load_synthetic(model)
    model = model.to(dev).train()
    model.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else None
    tr = ArenaSGD(model, 0.1, 0.9, 1e-4, group=dist.group.WORLD if world > 1 else None, world_size=world,
                  graph=not getattr(args, "no_graph", False))
    gen = torch.Generator(device=dev).manual_seed(4321 + rank)
    x = torch.randn(batch, 3, 224, 224, device=dev, generator=gen)
    t = torch.randint(0, 1000, (batch,), device=dev, generator=gen)
    for _ in range(max(args.warmup, 4)):        # steps 0 and 1 run launched, step 2 is captured, replays from then on
        loss, _ = tr.step(x, t)
    torch.cuda.synchronize()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = tr.step(x, t)
    torch.cuda.synchronize()
    barrier(world)
    elapsed = max_over_ranks(time.perf_counter() - t0, world, dev)
    value = world * batch * args.steps / elapsed
    return {
        "metric": "img/s erfnet_imagenet.ERFNet(1000) 224x224 training (config 4)",
        "value": value, "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"erfnet_imagenet.ERFNet(1000) training step (fwd + CE + bwd + RCCL all-reduce + SGD), batch {batch}/GPU, "
                               f"3x224x224, data parallel x{world}", "global_batch": batch * world, "final_loss": float(loss),
                   "hip_graph": bool(tr._graphs)},
        "model_tflops": value * 12.35e9 / 1e12,
    }
