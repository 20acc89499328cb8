"""Data-parallel gradient exchange for the trainable part of a FlamingoModel (resampler, gated xattn blocks, token
embedding) — one process per GPU, RCCL (`backend="nccl"`) over xGMI.

The reference gets this implicitly from HF Trainer's DistributedDataParallel (training/train.sh:26,36).  Here the fused
modules already emit all of their parameter gradients into ONE flat buffer per module (functional._flat_grads), so a
bucket is simply that buffer: as soon as a block's backward kernels are enqueued its buffer is all-reduced (mean) on a
side stream while the backward of the layers below keeps running.  Bucket sizes at config B: 31 MB per xattn block
(bf16), 126 MB resampler, 129 MB embedding — large enough to run the 7 xGMI links at bandwidth, small enough to overlap.
Only the un-fused trainable parameters (the token embedding) need a post-accumulate hook.

One backward per optimizer step is assumed (the flat buffers ARE `param.grad`; use zero_grad(set_to_none=True)).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist

from . import functional as F


class GradientAllReducer:
    def __init__(self, model: torch.nn.Module, process_group: Optional[dist.ProcessGroup] = None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self.cuda = self.backend == "nccl"
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.pending: List = []
        fused = {id(p) for m in model.modules() if hasattr(m, "fused_params") for p in m.fused_params()}
        self.loose = [p for p in model.parameters() if p.requires_grad and id(p) not in fused]
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_param) for p in self.loose]
        F.add_grad_ready_callback(self._on_bucket)

    def close(self):
        F.remove_grad_ready_callback(self._on_bucket)
        for h in self._hooks:
            h.remove()

    # -- called from inside backward --
    def _on_param(self, p: torch.Tensor):
        self._on_bucket(p.grad)

    def _on_bucket(self, flat: torch.Tensor):
        if self.world == 1:
            return
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record()                                   # after the kernels producing `flat` on the compute stream
            flat.record_stream(self.stream)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            self.pending.append((flat, work, False))
        else:  # gloo (CPU tests): no AVG, divide afterwards
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((flat, work, True))

    def finish(self):
        """Call after backward(), before optimizer.step(): the compute stream waits for the outstanding collectives."""
        for flat, work, divide in self.pending:
            work.wait()
            if divide:
                flat.div_(self.world)
        self.pending.clear()
