"""Checkpoint interchange with the reference's scripts.

* `load_my_state_dict(model, state_dict)` -- the tolerant loader of eval/eval_iou.py:54-65: every entry whose name (with or
  without the `module.` prefix that nn.DataParallel adds) exists in the model is copied in place, others are reported and
  skipped, nothing is required to be complete.
* `checkpoint_dict` / `save_checkpoint` / `resume` -- the dictionary train/main.py:320-326 writes after every epoch
  ({'epoch', 'arch', 'state_dict', 'best_acc', 'optimizer'}) and :143-153 reads back under --resume.  `data_parallel_keys`
  writes the parameter names the way the reference's DataParallel-wrapped model does (`module.` in front), so that a file
  written here opens under the reference's `model.load_state_dict` and the reverse.

Files are read with `torch.load(..., weights_only=True)`: tensors, numbers, strings and containers only.
"""
from __future__ import annotations

import shutil

import torch

PREFIX = "module."


def strip_module_prefix(state_dict):
    """Names without the DataParallel prefix (first occurrence only, as a wrapper adds exactly one)."""
    return {(k[len(PREFIX):] if k.startswith(PREFIX) else k): v for k, v in state_dict.items()}


def load_my_state_dict(model, state_dict, report=print):
    """eval/eval_iou.py:54-65.  Returns the model; `report(name, " not loaded")` for every entry without a home."""
    own = model.state_dict()
    ownp = {PREFIX + k: v for k, v in own.items()}
    with torch.no_grad():
        for name, param in state_dict.items():
            dst = own.get(name)
            if dst is None and name.startswith(PREFIX):
                dst = own.get(name.split(PREFIX)[-1])
            if dst is None:
                dst = ownp.get(name)                      # a bare checkpoint into a wrapped model
            if dst is None:
                if report is not None:
                    report(name, " not loaded")
                continue
            dst.copy_(param)
    from . import ops
    ops.bump_param_epoch()                                # cached packed weights are stale now
    return model


def checkpoint_dict(model, trainer=None, epoch=0, best_acc=0.0, data_parallel_keys=True):
    """train/main.py:320-326.  `epoch` is the one to start from on resume (the reference stores epoch + 1)."""
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    if data_parallel_keys:
        sd = {PREFIX + k: v for k, v in sd.items()}
    out = {"epoch": int(epoch), "arch": str(model), "state_dict": sd, "best_acc": float(best_acc)}
    if trainer is not None:
        opt = trainer.optimizer_state_dict()
        for st in opt["state"].values():
            for k, v in list(st.items()):
                if torch.is_tensor(v):
                    st[k] = v.detach().cpu()
        out["optimizer"] = opt
    return out


def save_checkpoint(state, is_best, filename_checkpoint, filename_best):
    """train/main.py:505-509: always the running checkpoint, and a copy of it when it is the best so far."""
    torch.save(state, filename_checkpoint)
    if is_best:
        shutil.copyfile(filename_checkpoint, filename_best)


def resume(path, model, trainer=None, map_location="cpu"):
    """train/main.py:143-153.  Returns (start_epoch, best_acc)."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    sd = ck["state_dict"]
    own = model.state_dict()
    if not all(k in own for k in sd):
        sd = strip_module_prefix(sd)
    model.load_state_dict(sd)                             # strict, like the reference's resume
    from . import ops
    ops.bump_param_epoch()
    if trainer is not None and "optimizer" in ck:          # (a Trainer's parameters are views of its arena: already updated)
        trainer.load_optimizer_state_dict(ck["optimizer"])
    return int(ck.get("epoch", 1)), float(ck.get("best_acc", 0.0))
