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


def _split_kv_buckets(model: torch.nn.Module, layers_per_bucket: int = 4) -> None:
    """With collectives in play the hoisted K / V projection runs as one call per `layers_per_bucket` layers (FlamingoBaseModel.
    kv_project_group): each call is its own autograd node with its own gradient bucket, ready as soon as its layers' backward is done."""
    for m in model.modules():
        if hasattr(m, "kv_project_group") and m.kv_project_group == 0:
            m.kv_project_group = layers_per_bucket


def _bucket_is_ours(owners, ids) -> bool:
    """The gradient-ready callbacks are process-wide: a second model in the process (an evaluation copy, a test's reference model)
    announces its buckets too.  A bucket belongs to a reducer iff all of its parameters are that reducer's model's."""
    mine = [id(p) in ids for p, _, _ in owners]
    if all(mine):
        return True
    if any(mine):
        raise RuntimeError("a gradient bucket mixes parameters of the wrapped model with foreign ones")
    return False


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
        self._fused_ids = fused
        self.loose = [p for p in model.parameters() if p.requires_grad and id(p) not in fused]
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_param) for p in self.loose]
        if self.active:
            _split_kv_buckets(model)
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
        if not (self.active and self._sync) or not owners or not _bucket_is_ours(owners, self._fused_ids):
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


class ShardedAdamW:
    """Data-parallel AdamW with the optimizer state sharded over the ranks (reduce-scatter -> sharded update ->
    all-gather), pipelined per gradient bucket on the side stream while backward continues.

    A bucket is the flat gradient buffer of one fused module (functional._flat_grads).  On its first appearance the bucket's
    parameters are moved into ONE flat parameter buffer with the same layout (each `p.data` becomes a view of it), and this rank
    allocates moments (and fp32 master copies, `master_dtype=torch.float32`) for its 1/world slice only.  Every step, as soon as a
    bucket's gradients are final:   reduce_scatter(AVG) -> ff_adamw_step on the rank's slice -> all_gather of the updated
    parameters - all on the reducer's stream, so communication AND the update overlap with the backward of the layers below
    (nothing that is still to run in this step reads those weights).  Un-fused parameters (the token embedding) are all-reduced
    and updated replicated, as in GradientAllReducer + FusedAdamW.  Call `finish_step()` after backward() - it replaces
    `reducer.finish(); optimizer.step()`.

    xGMI arithmetic (8 GPUs, 7 links x ~153 GB/s each): reduce-scatter + all-gather move 2 * (S / 8) per link pair instead of a
    ring's 2 * (7/8) * S over one link, and the update touches 1/8 of the state per rank.
    """

    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 master_dtype=None, process_group: Optional[dist.ProcessGroup] = None, force_collectives: bool = False, update_fn=None):
        from .optim import FusedAdamW
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self.cuda = self.backend == "nccl"
        self.collectives = self.world > 1 or (force_collectives and dist.is_initialized())
        self.hp = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.master_dtype = master_dtype
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.step_count = 0
        self.buckets = {}            # position of the bucket in the backward pass -> state
        self._work: List = []
        self._update_fn = update_fn or self._hip_update
        fused = {id(p) for m in model.modules() if hasattr(m, "fused_params") for p in m.fused_params()}
        self._fused_ids = fused
        self._arrival = 0                                           # buckets are identified by their position in the backward pass
        self.loose = [p for p in model.parameters() if p.requires_grad and id(p) not in fused]
        self._loose_work: List = []
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_loose) for p in self.loose]
        self._loose_opt = FusedAdamW(self.loose, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, master_dtype=master_dtype) if update_fn is None else None
        self._loose_update_fn = update_fn
        self._loose_state = {}
        if self.collectives:
            _split_kv_buckets(model)
        F.add_grad_ready_callback(self._on_bucket)

    def close(self):
        F.remove_grad_ready_callback(self._on_bucket)
        for h in self._hooks:
            h.remove()

    # ---- the kernel call (tests substitute a torch implementation through update_fn on CPU ranks) ----
    def _hip_update(self, p, g, m, v, master, step):
        from . import ffi
        import ctypes as C
        lib = ffi.lib()
        desc = ffi.AdamWDesc(ffi.dtype_code(p.dtype), 1, step, self.hp["lr"], self.hp["betas"][0], self.hp["betas"][1], self.hp["eps"],
                             self.hp["weight_decay"], 1.0, None)
        one = lambda t: ffi.ptr_array([t])
        ffi.check(lib.ff_adamw_step_mixed(desc, ffi.dtype_code(m.dtype), one(p), one(g), one(m), one(v), None if master is None else one(master),
                                          None, (C.c_longlong * 1)(p.numel()), ffi.stream_handle(p.device)), "ff_adamw_step_mixed")

    def _bucket_state(self, flat, owners):
        key = self._arrival                      # the order of the buckets within a backward pass is the same every step (static graph)
        self._arrival += 1
        sig = (flat.numel(), tuple((off, cnt) for _, off, cnt in owners))
        st = self.buckets.get(key)
        if st is not None and st["sig"] != sig:
            raise RuntimeError("ShardedAdamW: the gradient buckets of this backward pass do not arrive in the order of the first one")
        if st is None:
            n = flat.numel()
            assert n % self.world == 0, "flat gradient buffers are padded to a multiple of 1024 elements (functional._flat_offsets)"
            shard = n // self.world
            pflat = torch.zeros(n, dtype=flat.dtype, device=flat.device)
            with torch.no_grad():
                for p, off, cnt in owners:       # parameters move into the flat buffer; the modules keep seeing them under their own names
                    pflat[off:off + cnt].copy_(p.detach().reshape(-1))
                    p.data = pflat[off:off + cnt].view(p.shape)
            lo = self.rank * shard
            sdt = torch.float32 if (self.master_dtype is not None and flat.dtype == torch.bfloat16) else flat.dtype
            st = dict(sig=sig, params=[p for p, _, _ in owners], pflat=pflat, shard=shard, lo=lo, m=torch.zeros(shard, dtype=sdt, device=flat.device), v=torch.zeros(shard, dtype=sdt, device=flat.device),
                      master=pflat[lo:lo + shard].to(torch.float32) if sdt != flat.dtype else None, gshard=torch.empty(shard, dtype=flat.dtype, device=flat.device))
            self.buckets[key] = st
        return st

    def _on_bucket(self, flat: torch.Tensor, owners=()):
        if not owners or not _bucket_is_ours(owners, self._fused_ids):
            return
        st = self._bucket_state(flat, owners)
        step = self.step_count + 1
        lo, shard = st["lo"], st["shard"]

        def pipeline():
            if self.collectives:
                if self.cuda:
                    dist.reduce_scatter_tensor(st["gshard"], flat, op=dist.ReduceOp.AVG, group=self.group)
                else:       # gloo has no reduce-scatter: all-reduce and keep this rank's slice
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                    st["gshard"].copy_(flat[lo:lo + shard]).div_(self.world)
                g = st["gshard"]
            else:
                g = flat[lo:lo + shard]
            self._update_fn(st["pflat"][lo:lo + shard], g, st["m"], st["v"], st["master"], step)
            if self.collectives:
                dist.all_gather_into_tensor(st["pflat"], st["pflat"][lo:lo + shard].clone() if not self.cuda else st["pflat"][lo:lo + shard], group=self.group)

        if self.cuda:
            ready = torch.cuda.Event()
            ready.record()
            if not torch.cuda.is_current_stream_capturing():
                flat.record_stream(self.stream)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                pipeline()
                done = torch.cuda.Event()
                done.record()
            self._work.append(done)
        else:
            pipeline()

    def _on_loose(self, p: torch.Tensor):
        if self.collectives:
            if self.cuda:
                dist.all_reduce(p.grad, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
                p.grad.div_(self.world)

    def finish_step(self):
        """After backward(): wait for the per-bucket pipelines, update the un-fused parameters, advance the step count."""
        for ev in self._work:
            torch.cuda.current_stream().wait_event(ev)
        self._work.clear()
        self._arrival = 0
        self.step_count += 1
        if self._loose_opt is not None:
            self._loose_opt.step()
        else:
            for p in self.loose:
                if p.grad is None:
                    continue
                s = self._loose_state.setdefault(id(p), dict(m=torch.zeros_like(p), v=torch.zeros_like(p)))
                self._loose_update_fn(p.data.view(-1), p.grad.view(-1), s["m"].view(-1), s["v"].view(-1), None, self.step_count)

    def zero_grad(self, set_to_none: bool = True):
        """Gradients are re-created by every backward (flat buffers), so they are simply dropped."""
        for st in self.buckets.values():
            for p in st["params"]:
                p.grad = None
        for p in self.loose:
            p.grad = None
