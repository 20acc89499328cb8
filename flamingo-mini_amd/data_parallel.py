"""Data-parallel gradient exchange for the trainable part of a FlamingoModel (resampler, gated xattn blocks, token
embedding) — one process per GPU, RCCL (`backend="nccl"`) over xGMI.

The reference gets this implicitly from HF Trainer's DistributedDataParallel (training/train.sh:26,36).  Here the fused
modules already emit all of their parameter gradients into ONE flat buffer per module (functional._flat_grads), so a
bucket is simply that buffer: as soon as a block's backward kernels are enqueued its buffer is all-reduced (mean) on a
side stream while the backward of the layers below keeps running.  Bucket sizes at config B: 31 MB per xattn block
(bf16), 126 MB resampler, 129 MB embedding — large enough to run the 7 xGMI links at bandwidth, small enough to overlap.
Only the un-fused trainable parameters (the token embedding) need a post-accumulate hook.

Gradient accumulation: run every micro-batch but the last under `reducer.no_sync()` (as with DistributedDataParallel); the last
backward then finds existing `.grad`s, lets autograd accumulate, and the reducer all-reduces the accumulated `.grad` of those
parameters after backward (no overlap on that step).  `finish()` verifies that every early-reduced flat slice really is the
parameter's `.grad` (autograd adopts an incoming gradient only while nobody else references it - otherwise it clones) and repairs
the ones that are not.
"""
from __future__ import annotations

import contextlib
from typing import List, Optional

import torch
import torch.distributed as dist

from . import functional as F


class GradientAllReducer:
    def __init__(self, model: torch.nn.Module, process_group: Optional[dist.ProcessGroup] = None, force_collectives: bool = False):
        """force_collectives: issue the collectives even in a 1-rank group (exercises the RCCL path on a single GPU)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self.cuda = self.backend == "nccl"
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.pending: List = []
        self.late: List[torch.Tensor] = []          # parameters whose gradient is being accumulated: reduced after backward
        self._early = set()                         # ids of parameters already reduced in this step
        self._sync = True
        fused = {id(p) for m in model.modules() if hasattr(m, "fused_params") for p in m.fused_params()}
        self.loose = [p for p in model.parameters() if p.requires_grad and id(p) not in fused]
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_param) for p in self.loose]
        F.add_grad_ready_callback(self._on_bucket)

    def close(self):
        F.remove_grad_ready_callback(self._on_bucket)
        for h in self._hooks:
            h.remove()

    @contextlib.contextmanager
    def no_sync(self):
        """Micro-batches whose gradients are only accumulated locally (every one but the last of an optimizer step)."""
        self._sync = False
        try:
            yield
        finally:
            self._sync = True

    # -- called from inside backward --
    def _on_param(self, p: torch.Tensor):
        """post-accumulate hook of an un-fused parameter: p.grad is final for this backward (accumulated or not)."""
        if self.active and self._sync:
            self._reduce_async(p.grad, [])

    def _on_bucket(self, flat: torch.Tensor, owners=()):
        if not (self.active and self._sync):
            return
        es = flat.element_size()
        if any(p.grad is not None and p.grad.data_ptr() != flat.data_ptr() + off * es for p, off, _ in owners):     # (a deferred gradient that autograd
            # has already adopted IS its slice of `flat`: that is not accumulation)
            # accumulation: autograd is about to ADD these slices to existing .grad tensors - the flat buffer is not the gradient
            for p, _, _ in owners:
                if id(p) in self._early:
                    raise RuntimeError("a fused module ran backward twice in one step after its gradients were all-reduced; wrap all but "
                                       "the last micro-batch in GradientAllReducer.no_sync()")
                self.late.append(p)
            return
        self._early.update(id(p) for p, _, _ in owners)
        self._reduce_async(flat, list(owners))

    def _reduce_async(self, flat: torch.Tensor, owners):
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record()                                   # after the kernels producing `flat` on the compute stream
            if not torch.cuda.is_current_stream_capturing():
                flat.record_stream(self.stream)              # (inside a graph capture all memory is the graph's own static pool)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self.pending.append((flat, work, False, owners))
        else:  # gloo (CPU tests): no AVG, divide afterwards
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((flat, work, True, owners))

    def finish(self):
        """Call after backward(), before optimizer.step(): the compute stream waits for the outstanding collectives."""
        for flat, work, divide, owners in self.pending:
            work.wait()
            if divide:
                flat.div_(self.world)
            for p, off, n in owners:     # the reduced slice must BE the parameter's gradient (see the module docstring)
                if p.grad is not None and p.grad.data_ptr() != flat.data_ptr() + off * flat.element_size():
                    p.grad.copy_(flat[off:off + n].view(p.shape))
        self.pending.clear()
        seen = set()
        for p in self.late:
            if id(p) in seen or p.grad is None:
                continue
            seen.add(id(p))
            if self.cuda:
                dist.all_reduce(p.grad, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
                p.grad.div_(self.world)
        self.late.clear()
        self._early.clear()
