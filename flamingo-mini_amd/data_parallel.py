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


def _bucket_launch_structure(model: torch.nn.Module, layers_per_bucket: int = 4) -> list:
    """With collectives in play the hoisted K / V projection runs as one call per `layers_per_bucket` layers (FlamingoBaseModel.
    kv_project_group): each call is its own autograd node with its own gradient bucket, ready as soon as its layers' backward is done;
    and the deferred weight gradients of the blocks are flushed every `layers_per_bucket` layers as well (single-GPU default: 12, which is
    faster per launch but would hold back every bucket until a third of backward has passed).  These are settings OF THE MODEL
    (set_launch_structure), not of the process.  Returns what close() needs to take back exactly what was changed here and nothing else:
    [(owner object, attribute, value before, value set)] - the K / V group of the model and `wgrad_group` of EVERY block (per block: blocks
    may differ), so that a setting the user changed after constructing the reducer, or one this function never touched (hoist_kv,
    defer_wgrad), survives close()."""
    undo = []
    for m in model.modules():
        if hasattr(m, "set_launch_structure") and hasattr(m, "kv_project_group") and hasattr(m, "get_modified_layers"):
            group = m.kv_project_group if m.kv_project_group > 0 else layers_per_bucket
            if m.kv_project_group != group:
                undo.append((m, "kv_project_group", m.kv_project_group, group))
                m.kv_project_group = group
            for hook in m.get_modified_layers():
                blk = hook.xattn_block
                if blk.wgrad_group != layers_per_bucket:
                    undo.append((blk, "wgrad_group", blk.wgrad_group, layers_per_bucket))
                    blk.wgrad_group = layers_per_bucket
        if hasattr(m, "layerwise") and hasattr(m, "fused_params") and not m.layerwise:
            # the resampler layer by layer (ff_resampler_layer_*): one gradient bucket per layer, final when that layer's backward is done,
            # instead of all of the resampler's gradients at the very end of backward
            undo.append((m, "layerwise", False, True))
            m.layerwise = True
    return undo


def _restore_launch_structure(undo: list) -> None:
    for obj, name, before, set_to in undo:
        if getattr(obj, name) == set_to:        # still what the reducer set: nobody has changed it since
            setattr(obj, name, before)


def _bucket_is_ours(owners, ids) -> bool:
    """The gradient-ready callbacks are process-wide: a second model in the process (an evaluation copy, a test's reference model)
    announces its buckets too.  A bucket belongs to a reducer iff all of its parameters are that reducer's model's."""
    mine = [id(p) in ids for p, _, _ in owners]
    if all(mine):
        return True
    if any(mine):
        raise RuntimeError("a gradient bucket mixes parameters of the wrapped model with foreign ones")
    return False


class _StreamWork:
    """`work.wait()` of a collective that was issued synchronously on the side stream: the current stream waits for the recorded event."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)


class GradientAllReducer:
    def __init__(self, model: torch.nn.Module, process_group: Optional[dist.ProcessGroup] = None, force_collectives: bool = False,
                 reduce_dtype: Optional[torch.dtype] = None):
        """force_collectives: issue the collectives even in a 1-rank group (exercises the RCCL path on a single GPU).
        reduce_dtype=torch.float32: bf16 buckets are widened to fp32 for the exchange (the sum over the ranks is accumulated in fp32 and
        rounded to bf16 once, at twice the bytes on the links); None = exchange in the gradients' own dtype (ReduceOp.AVG on bf16)."""
        self.reduce_dtype = reduce_dtype
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
        self._fused_params = [p for p in model.parameters() if p.requires_grad and id(p) in fused]
        self._names = dict(model.named_parameters())
        self.loose = [p for p in model.parameters() if p.requires_grad and id(p) not in fused]
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_param) for p in self.loose]
        self._undo = _bucket_launch_structure(model) if self.active else []
        self._collect: Optional[list] = None      # graphs.PiecewiseGraphedTrainStep: buckets are recorded while a segment is captured ...
        self.defer_loose = False                  # ... and un-fused parameters wait for finish() when backward runs in several segments
        self.timeline: Optional[list] = None      # record_timeline(): (label, bytes, ready event, done event) per exchanged bucket
        F.add_grad_ready_callback(self._on_bucket)

    def close(self):
        """Detach from the model: callbacks and hooks removed; the launch-structure settings this reducer changed (and only those, and only
        where they still hold the value it set) go back to what they were."""
        F.remove_grad_ready_callback(self._on_bucket)
        for h in self._hooks:
            h.remove()
        _restore_launch_structure(self._undo)
        self._undo = []

    # -- piecewise capture: record instead of exchanging --
    def begin_collect(self) -> None:
        self._collect = []

    def end_collect(self) -> list:
        out, self._collect = self._collect or [], None
        return out

    def reduce_bucket(self, flat: torch.Tensor, owners) -> None:
        """Exchange one recorded bucket now (its producers are already enqueued on the current stream)."""
        self.reduce_buckets([(flat, owners)])

    def reduce_buckets(self, buckets, producers_done: bool = False) -> None:
        """Exchange the recorded buckets of one backward segment, all of whose producers are already enqueued on the current stream: ONE
        ready / done event pair for the segment (the piecewise step used to pay a pair per bucket: 94 events and as many cross-stream waits
        per step), the collectives back to back on the side stream.
        producers_done=True: the caller has SEEN the producers finish (it synchronised with an event behind them), so the side stream needs
        no wait - see PiecewiseGraphedTrainStep(pace="host") for why that is worth a host round trip."""
        if not self.active or not buckets:
            return
        for _, owners in buckets:
            self._early.update(id(p) for p, _, _ in owners)
        if not self.cuda:
            for flat, owners in buckets:
                self._reduce_async(flat, list(owners))
            return
        timed = self.timeline is not None
        ready = torch.cuda.Event(enable_timing=timed)
        if not producers_done:
            ready.record()
        with torch.cuda.stream(self.stream):
            if producers_done:
                ready.record()                               # (timeline only: when the exchange was issued)
            else:
                self.stream.wait_event(ready)
            for flat, _ in buckets:
                flat.record_stream(self.stream)
                self._mean_in_place(flat)
            done = torch.cuda.Event(enable_timing=timed)
            done.record()
        work = _StreamWork(done)
        for i, (flat, owners) in enumerate(buckets):
            if timed:
                label = next((n for n, q in self._names.items() if owners and q is owners[0][0]), None) or ("loose" if not owners else "bucket")
                self.timeline.append((label, flat.numel() * flat.element_size(), ready, done))
            self.pending.append((flat, work if i == 0 else _StreamWork(None), list(owners)))

    def record_timeline(self, on: bool = True) -> None:
        """GPU only: keep (label, bytes, ready, done) event pairs of every bucket exchanged from now on - `timeline_ms()` after a
        synchronised step says when each bucket became ready and how long its collective was in flight."""
        self.timeline = [] if on else None

    def timeline_ms(self, origin: "torch.cuda.Event"):
        rows = []
        for label, nbytes, ready, done in self.timeline or []:
            rows.append(dict(bucket=label, mb=round(nbytes / 1e6, 1), ready_ms=round(origin.elapsed_time(ready), 3), done_ms=round(origin.elapsed_time(done), 3)))
        return rows

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
            if self._collect is not None:
                self._collect.append((p.grad, []))
            elif self.defer_loose:
                self.late.append(p)
            else:
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
        if self._collect is not None:
            self._collect.append((flat, list(owners)))
            return
        self._early.update(id(p) for p, _, _ in owners)
        self._reduce_async(flat, list(owners))

    def _mean_in_place(self, t: torch.Tensor) -> None:
        """t <- mean over the ranks, on the current stream / thread, honouring reduce_dtype (the one place that decides how a gradient
        tensor is exchanged; the early buckets and the accumulated-gradient path both end here)."""
        widen = self.reduce_dtype is not None and t.dtype != self.reduce_dtype
        x = t.to(self.reduce_dtype) if widen else t
        if self.cuda:
            # (one rank - force_collectives on a single GPU: the mean IS the sum.  RCCL turns AVG into a pre-multiplied sum, which even at one rank
            # is a kernel that reads and rewrites the whole bucket - `oneRankReduce<FuncPreMulSum>`: 0.39 ms per 115 MB segment on a side queue,
            # ~4.4 ms of HBM-bound work per step beside the backward (profiles/r05_rccl_presence_trace.txt); an in-place SUM over one rank is nothing.)
            dist.all_reduce(x, op=dist.ReduceOp.AVG if self.world > 1 else dist.ReduceOp.SUM, group=self.group)
        else:            # gloo (CPU tests): no AVG
            dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
            x.div_(self.world)
        if widen:
            t.copy_(x)

    def _reduce_async(self, flat: torch.Tensor, owners):
        if self.cuda:
            timed = self.timeline is not None and not torch.cuda.is_current_stream_capturing()
            ready = torch.cuda.Event(enable_timing=timed)
            ready.record()                                   # after the kernels producing `flat` on the compute stream
            if not torch.cuda.is_current_stream_capturing():
                flat.record_stream(self.stream)              # (inside a graph capture all memory is the graph's own static pool)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                self._mean_in_place(flat)                    # enqueued on the side stream
                done = torch.cuda.Event(enable_timing=timed)
                done.record()
            if timed:
                label = next((n for n, q in getattr(self, "_names", {}).items() if owners and q is owners[0][0]), None) or ("loose" if not owners else "bucket")
                self.timeline.append((label, flat.numel() * flat.element_size(), ready, done))
            self.pending.append((flat, _StreamWork(done), owners))
        else:
            self._mean_in_place(flat)                        # CPU ranks: synchronous
            self.pending.append((flat, _StreamWork(None), owners))

    def finish(self):
        """Call after backward(), before optimizer.step(): the compute stream waits for the outstanding collectives."""
        for flat, work, owners in self.pending:
            work.wait()
            for p, off, n in owners:     # the reduced slice must BE the parameter's gradient (see the module docstring)
                if p.grad is not None and p.grad.data_ptr() != flat.data_ptr() + off * flat.element_size():
                    p.grad.copy_(flat[off:off + n].view(p.shape))
        self.pending.clear()
        seen = set()
        for p in self.late:
            if id(p) in seen or p.grad is None:
                continue
            seen.add(id(p))
            self._mean_in_place(p.grad)
        if self.active and self._sync and self._collect is None:
            # fused parameters whose gradient did not arrive as a bucket of this model (torch.autocast over fp32 parameters: the fused modules ran
            # on CASTS of them, whose buckets are not ours; the gradients reached the parameters through the casts' backward): exchanged here
            for p in self._fused_params:
                if p.grad is not None and id(p) not in self._early and id(p) not in seen:
                    self._mean_in_place(p.grad)
        self.late.clear()
        self._early.clear()


class ShardedAdamW(torch.optim.Optimizer):
    """Data-parallel AdamW with the optimizer state sharded over the ranks (reduce-scatter -> sharded update ->
    all-gather), pipelined per gradient bucket on the side stream while backward continues.

    A bucket is the flat gradient buffer of one fused module (functional._flat_grads).  On its first appearance the bucket's
    parameters are moved into ONE flat parameter buffer with the same layout (each `p.data` becomes a view of it), and this rank
    allocates moments (and fp32 master copies, `master_dtype=torch.float32`) for its 1/world slice only.  Every step, as soon as a
    bucket's gradients are final:   reduce_scatter(AVG) -> ff_adamw_step on the rank's slice -> all_gather of the updated
    parameters - all on the reducer's stream, so communication AND the update overlap with the backward of the layers below
    (nothing that is still to run in this step reads those weights).  Un-fused parameters (the token embedding) are all-reduced
    and updated replicated, as in GradientAllReducer + FusedAdamW.  Call `step()` (= `finish_step()`) after backward() - it replaces
    `reducer.finish(); optimizer.step()`.

    It is a torch.optim.Optimizer: `param_groups[0]["lr"]` is the learning rate the next backward's updates use, so torch / HF LR
    schedulers attach as usual (the reference trains with constant_with_warmup); `state_dict()` / `load_state_dict()` save and restore
    THIS RANK's shards (every rank saves its own file; world size and rank must match on load).  Gradient accumulation: every
    micro-batch but the last under `no_sync()`; the last backward then finds accumulated `.grad`s and the bucket's pipeline runs in
    `step()` on the accumulated values.  A second backward without no_sync() after a bucket was already updated in this step raises.
    `capturable=True` keeps the step count and the learning rate in device scalars (graphs.GraphedTrainStep replays the step).

    xGMI arithmetic (8 GPUs, 7 links x ~153 GB/s each): reduce-scatter + all-gather move 2 * (S / 8) per link pair instead of a
    ring's 2 * (7/8) * S over one link, and the update touches 1/8 of the state per rank.
    """

    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 master_dtype=None, process_group: Optional[dist.ProcessGroup] = None, force_collectives: bool = False, update_fn=None,
                 capturable: bool = False, overlap: bool = False):
        from .optim import FusedAdamW
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self.cuda = self.backend == "nccl"
        self.collectives = self.world > 1 or (force_collectives and dist.is_initialized())
        self.master_dtype = master_dtype
        self.capturable = capturable
        # The per-bucket pipeline runs on a side stream next to backward whenever there are collectives to hide.  overlap=True does the same
        # on a single GPU, where the pipeline is just the update - measured SLOWER there (config B, graph replay: 35.4 -> 37.2 ms per step):
        # 48 small AdamW launches compete with the backward kernels for the CUs' load path instead of filling idle HBM time.
        on_gpu = torch.cuda.is_available() and any(p.is_cuda for p in model.parameters())
        self.stream = torch.cuda.Stream() if (self.cuda or (overlap and on_gpu and update_fn is None)) else None
        self.step_count = 0
        self.buckets = {}            # ids of the bucket's parameters -> state (independent of the order the buckets arrive in)
        self._work: List = []
        self._late: List = []        # buckets whose gradients are being accumulated: their pipeline runs in step()
        self._updated = set()        # buckets already updated in the running step
        self._sync = True
        self._loaded = None          # a state_dict loaded before the buckets exist: applied as they are created
        self._dev_scalars = {}       # capturable: device -> (step counter, learning rate)
        self._update_fn = update_fn or self._hip_update
        self._model = model
        self._names = {id(p): n for n, p in model.named_parameters()}
        fused = {id(p) for m in model.modules() if hasattr(m, "fused_params") for p in m.fused_params()}
        self._fused_ids = fused
        self._arrival = 0                                           # buckets seen so far in the running backward pass
        self.loose = [p for p in model.parameters() if p.requires_grad and id(p) not in fused]
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_loose) for p in self.loose]
        trainable = [p for p in model.parameters() if p.requires_grad]
        super().__init__(trainable, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._loose_opt = FusedAdamW(self.loose, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, master_dtype=master_dtype,
                                     capturable=capturable) if update_fn is None else None
        self._loose_update_fn = update_fn
        self._loose_state = {}
        self._undo = _bucket_launch_structure(model) if self.collectives else []
        F.add_grad_ready_callback(self._on_bucket)

    @property
    def hp(self):
        return self.param_groups[0]

    def close(self):
        F.remove_grad_ready_callback(self._on_bucket)
        for h in self._hooks:
            h.remove()
        _restore_launch_structure(self._undo)
        self._undo = []

    @contextlib.contextmanager
    def no_sync(self):
        """Micro-batches whose gradients are only accumulated locally (every one but the last of an optimizer step)."""
        self._sync = False
        try:
            yield
        finally:
            self._sync = True

    def set_lr(self, lr: float) -> None:
        for g in self.param_groups:
            g["lr"] = lr
        self.sync_device_hyperparams()

    def sync_device_hyperparams(self) -> None:
        """capturable mode: copy the host learning rate into the device scalars the captured kernels read (before a graph replay)."""
        for dev, (_, lr_dev, seen) in self._dev_scalars.items():
            if seen[0] != self.hp["lr"]:
                lr_dev.fill_(float(self.hp["lr"]))
                seen[0] = self.hp["lr"]
        if self._loose_opt is not None:
            for g in self._loose_opt.param_groups:
                g["lr"] = self.hp["lr"]
            self._loose_opt.sync_device_hyperparams()

    def _scalars(self, device):
        sc = self._dev_scalars.get(device)
        if sc is None:
            sc = self._dev_scalars[device] = (torch.full((), float(self.step_count), dtype=torch.float32, device=device),
                                              torch.full((), float(self.hp["lr"]), dtype=torch.float32, device=device), [self.hp["lr"]])
        return sc

    # ---- the kernel call (tests substitute a torch implementation through update_fn on CPU ranks) ----
    def _hip_update(self, p, g, m, v, master, step):
        from . import ffi
        import ctypes as C
        lib = ffi.lib()
        step_dev = lr_dev = None
        if self.capturable:
            sc = self._scalars(p.device)
            step, step_dev, lr_dev = 0, sc[0].data_ptr(), sc[1].data_ptr()
        desc = ffi.AdamWDesc(ffi.dtype_code(p.dtype), 1, step, self.hp["lr"], self.hp["betas"][0], self.hp["betas"][1], self.hp["eps"],
                             self.hp["weight_decay"], 1.0, step_dev)
        one = lambda t: ffi.ptr_array([t])
        ffi.check(lib.ff_adamw_step_mixed(desc, ffi.dtype_code(m.dtype), one(p), one(g), one(m), one(v), None if master is None else one(master),
                                          lr_dev, (C.c_longlong * 1)(p.numel()), ffi.stream_handle(p.device)), "ff_adamw_step_mixed")

    def _bucket_name(self, owners) -> str:
        return self._names.get(id(owners[0][0]), f"bucket{len(self.buckets)}")

    def _bucket_state(self, flat, owners):
        key = tuple(id(p) for p, _, _ in owners)
        self._arrival += 1
        st = self.buckets.get(key)
        if st is None:
            n = flat.numel()
            assert n % self.world == 0, "flat gradient buffers are padded to a multiple of 1024 elements (functional._flat_offsets)"
            shard = n // self.world
            pflat = torch.zeros(n, dtype=flat.dtype, device=flat.device)
            with torch.no_grad():
                for p, off, cnt in owners:       # parameters move into the flat buffer; the modules keep seeing them under their own names
                    pflat[off:off + cnt].copy_(p.detach().reshape(-1))
                    p.data = pflat[off:off + cnt].view(p.shape)
            for m in self._model.modules():      # a captured decode graph still reads the old parameter storage
                if hasattr(m, "reset_decode_sessions"):
                    m.reset_decode_sessions()
            lo = self.rank * shard
            sdt = torch.float32 if (self.master_dtype is not None and flat.dtype == torch.bfloat16) else flat.dtype
            st = dict(name=self._bucket_name(owners), owners=[(p, off, cnt) for p, off, cnt in owners], params=[p for p, _, _ in owners], pflat=pflat,
                      shard=shard, lo=lo, m=torch.zeros(shard, dtype=sdt, device=flat.device), v=torch.zeros(shard, dtype=sdt, device=flat.device),
                      master=pflat[lo:lo + shard].to(torch.float32) if sdt != flat.dtype else None, gshard=torch.empty(shard, dtype=flat.dtype, device=flat.device))
            if self._loaded is not None and st["name"] in self._loaded:
                self._restore_bucket(st, self._loaded.pop(st["name"]))
            self.buckets[key] = st
        return st

    def _pipeline(self, st, flat):
        """reduce-scatter -> update of this rank's slice -> all-gather, on the current stream."""
        lo, shard = st["lo"], st["shard"]
        if self.collectives:
            if self.cuda:
                dist.reduce_scatter_tensor(st["gshard"], flat, op=dist.ReduceOp.AVG, group=self.group)
            else:       # gloo has no reduce-scatter: all-reduce and keep this rank's slice
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                st["gshard"].copy_(flat[lo:lo + shard]).div_(self.world)
            g = st["gshard"]
        else:
            g = flat[lo:lo + shard]
        self._update_fn(st["pflat"][lo:lo + shard], g, st["m"], st["v"], st["master"], self.step_count + 1)
        if self.collectives:
            dist.all_gather_into_tensor(st["pflat"], st["pflat"][lo:lo + shard].clone() if not self.cuda else st["pflat"][lo:lo + shard], group=self.group)

    def _on_bucket(self, flat: torch.Tensor, owners=()):
        if not owners or not _bucket_is_ours(owners, self._fused_ids):
            return
        first = self._arrival == 0
        st = self._bucket_state(flat, owners)
        key = tuple(id(p) for p in st["params"])
        es = flat.element_size()
        accumulating = any(p.grad is not None and p.grad.data_ptr() != flat.data_ptr() + off * es for p, off, _ in owners)
        if key in self._updated:
            raise RuntimeError("ShardedAdamW: a fused module ran backward twice in one step after its parameters were already updated; "
                               "wrap all but the last micro-batch in ShardedAdamW.no_sync()")
        if not self._sync:
            return                               # autograd accumulates into .grad; nothing is exchanged or updated
        if accumulating:                         # the gradient of this step is .grad AFTER autograd has added `flat` to it: see step()
            if st not in self._late:
                self._late.append(st)
            return
        if self.capturable and first and flat.is_cuda:
            self._scalars(flat.device)[0].add_(1.0)       # the step counter the captured updates read, advanced on the device
        self._updated.add(key)
        if self.stream is not None:
            ready = torch.cuda.Event()
            ready.record()
            if not torch.cuda.is_current_stream_capturing():
                flat.record_stream(self.stream)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ready)
                self._pipeline(st, flat)
                done = torch.cuda.Event()
                done.record()
            self._work.append(done)
        else:
            self._pipeline(st, flat)

    def _on_loose(self, p: torch.Tensor):
        if self.collectives and self._sync:
            if self.cuda:
                dist.all_reduce(p.grad, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group)
                p.grad.div_(self.world)

    def finish_step(self):
        """After backward(): wait for the per-bucket pipelines, run the pipelines of buckets whose gradients were accumulated over several
        micro-batches, update the un-fused parameters, advance the step count."""
        for ev in self._work:
            torch.cuda.current_stream().wait_event(ev)
        self._work.clear()
        covered = {id(p) for st in self.buckets.values() for p in st["params"]}
        stray = [self._names.get(id(p), "?") for p in self._model.parameters()
                 if p.requires_grad and p.grad is not None and id(p) in self._fused_ids and id(p) not in covered]
        if stray:       # (torch.autocast over fp32 parameters: the fused modules ran on casts, no bucket of THIS model ever arrived)
            raise RuntimeError(f"ShardedAdamW: {len(stray)} fused parameter(s) (e.g. {stray[0]}) have a gradient that did not arrive as a gradient bucket "
                               "of their module - under torch.autocast the fused modules run on casts of fp32 parameters. Use bf16 parameters with "
                               "master_dtype=torch.float32 here, or GradientAllReducer + FusedAdamW under autocast.")
        for st in self._late:                    # accumulated gradients: gather .grad into the bucket layout, then the same pipeline
            flat = torch.zeros_like(st["pflat"])
            for p, off, cnt in st["owners"]:
                if p.grad is not None:
                    flat[off:off + cnt].copy_(p.grad.reshape(-1))
            if self.capturable and not self._updated and flat.is_cuda:
                self._scalars(flat.device)[0].add_(1.0)
            self._updated.add(tuple(id(p) for p in st["params"]))
            self._pipeline(st, flat)
        self._late.clear()
        self._updated.clear()
        self._arrival = 0
        self.step_count += 1
        if self._loose_opt is not None:
            for g in self._loose_opt.param_groups:
                g["lr"] = self.hp["lr"]
            self._loose_opt.step()
        else:
            for p in self.loose:
                if p.grad is None:
                    continue
                s = self._loose_state.setdefault(id(p), dict(m=torch.zeros_like(p), v=torch.zeros_like(p)))
                self._loose_update_fn(p.data.view(-1), p.grad.view(-1), s["m"].view(-1), s["v"].view(-1), None, self.step_count)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            from . import functional as _F
            _F.poll_sync_exchange("ShardedAdamW.step")      # non-blocking check of the fused cross-attention kernels' hand-off error word
        self.finish_step()
        return loss

    def zero_grad(self, set_to_none: bool = True):
        """Gradients are re-created by every backward (flat buffers), so they are simply dropped."""
        for st in self.buckets.values():
            for p in st["params"]:
                p.grad = None
        for p in self.loose:
            p.grad = None

    # ---- checkpointing: this rank's shards --------------------------------------------------------------------------------
    def state_dict(self):
        """This rank's optimizer state: per bucket (named after its first parameter) the moment shards and, in mixed precision, the
        fp32 master shard; plus the replicated state of the un-fused parameters.  Every rank saves its own."""
        if self.capturable:
            for sc in self._dev_scalars.values():
                self.step_count = max(self.step_count, int(float(sc[0])))
        buckets = {st["name"]: dict(lo=st["lo"], shard=st["shard"], exp_avg=st["m"], exp_avg_sq=st["v"], master=st["master"])
                   for st in self.buckets.values()}
        loose = self._loose_opt.state_dict() if self._loose_opt is not None else \
            {self._names.get(i, str(i)): dict(exp_avg=s["m"], exp_avg_sq=s["v"]) for i, s in self._loose_state.items()}
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        return dict(step=self.step_count, world=self.world, rank=self.rank, buckets=buckets, loose=loose, param_groups=groups)

    def _restore_bucket(self, st, src):
        if src["lo"] != st["lo"] or src["shard"] != st["shard"]:
            raise RuntimeError(f"ShardedAdamW.load_state_dict: bucket {st['name']} was saved with another sharding")
        st["m"].copy_(src["exp_avg"])
        st["v"].copy_(src["exp_avg_sq"])
        if st["master"] is not None:
            if src.get("master") is not None:
                st["master"].copy_(src["master"])
            else:
                st["master"].copy_(st["pflat"][st["lo"]:st["lo"] + st["shard"]])

    def load_state_dict(self, state):
        if state["world"] != self.world or state["rank"] != self.rank:
            raise RuntimeError(f"ShardedAdamW.load_state_dict: saved by rank {state['rank']} of {state['world']}, this is rank {self.rank} of {self.world}")
        self.step_count = int(state["step"])
        for g, src in zip(self.param_groups, state.get("param_groups", [])):
            g.update({k: v for k, v in src.items() if k != "params"})
        pending = dict(state["buckets"])
        for st in self.buckets.values():
            if st["name"] in pending:
                self._restore_bucket(st, pending.pop(st["name"]))
        self._loaded = pending                   # buckets are created by the first backward: their state is applied then
        if self._loose_opt is not None:
            self._loose_opt.load_state_dict(state["loose"])
        else:
            by_name = {n: i for i, n in self._names.items()}
            for n, s in state["loose"].items():
                if n in by_name:
                    self._loose_state[by_name[n]] = dict(m=s["exp_avg"].clone(), v=s["exp_avg_sq"].clone())
        for sc in self._dev_scalars.values():
            sc[0].fill_(float(self.step_count))
        self.sync_device_hyperparams()
