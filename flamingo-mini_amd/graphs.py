"""HIP-graph replay of a whole training step.

At flamingo-mini's benchmark size one step is ~3100 kernel launches of 5-40 us each; launched one by one from Python the
GPU idles ~10 % of the time between them.  Every launch of this library only enqueues work on the caller's stream (no
allocation, no host synchronisation), so forward + backward + optimizer can be captured once into a HIP graph
(torch.cuda.CUDAGraph on ROCm) and replayed with a single launch per step.

    step = GraphedTrainStep(model, optimizer, example_batch)         # warms up, captures
    loss = step(batch)                                                # copies the batch into the static inputs, replays

PiecewiseGraphedTrainStep is the variant that does not depend on collectives being capturable: the step is cut into sub-graphs
(forward | one backward segment per few LM layers | resampler backward | optimizer) that are replayed one after the other, and the
gradient collectives are issued EAGERLY between the replays, on the reducer's side stream, overlapping the next segment.

Requirements: fixed shapes, an optimizer whose step is capture-safe (FusedAdamW(capturable=True) or
torch.optim.AdamW(capturable=True)), and no data-dependent host control flow in the model (true for FlamingoModel's
training forward).  With `reducer=` (data_parallel.GradientAllReducer) the gradient all-reduces are part of the capture: they are
issued on the reducer's side stream, which forks from and rejoins the capturing stream through events, so the replayed graph
contains the RCCL kernels and their overlap with backward (PyTorch captures NCCL / RCCL collectives; the communicator must already
exist, which the eager warm-up steps guarantee).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import contextlib
import time

import torch


def _refuse_live_autograd_graphs(model: torch.nn.Module) -> None:
    """A parameter's AccumulateGrad node lives as long as some autograd graph refers to it and remembers the stream it was created on.  If
    outputs of an earlier eager forward (built on the default stream) are still alive, the captured backward would hand its gradients to
    those nodes: the engine then synchronises the capturing stream with the default stream in the middle of the capture - on ROCm 7 a
    segmentation fault in the runtime at the end of the capture (tools/capture_after_eager.py), not an error message.  Refuse up front.
    The probe: a mark left in the node's metadata survives only if somebody else keeps the node alive."""
    def accumulator(p):
        return p.view_as(p).grad_fn.next_functions[0][0]

    stale = []
    try:
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            accumulator(p).metadata["ff_capture_probe"] = True
            node = accumulator(p)
            if node.metadata.pop("ff_capture_probe", False):
                stale.append(name)
    except (AttributeError, IndexError, TypeError):      # a torch build whose autograd nodes cannot be probed this way: capture unguarded
        return
    if stale:
        raise RuntimeError(
            f"GraphedTrainStep: an autograd graph from an earlier forward still refers to {len(stale)} parameter(s) of the model (e.g. "
            f"{stale[0]}). Delete the outputs / losses of earlier forward passes (or run them under torch.no_grad()) before capturing: "
            "their gradient-accumulation nodes are bound to the stream of that forward and cannot take part in a stream capture.")


_WATCHDOG_PERIOD_S = 0.1        # ProcessGroupNCCL.hpp: kWatchdogThreadSleepMillis = 100 (a compile-time constant of the torch build, no environment knob)


def _let_the_watchdog_drain() -> None:
    """Called between the eager warm-up steps and a stream capture.  ProcessGroupNCCL's watchdog thread keeps every collective of the warm-up in
    a list and polls its events (hipEventQuery) once per period until it has seen them complete; on ROCm such a query from that thread while
    THIS thread captures has aborted the process now and then even in "thread_local" capture mode (round 5: no Python frame, no message).
    The list has no Python accessor (the process group exposes neither its length nor the watchdog's heartbeat), so the drain cannot be
    observed - but it can be bounded: once the DEVICE is idle every listed collective is complete, each watchdog pass visits the whole list
    and erases what is complete, and a pass starts at most one period after the previous one ended.  Device idle + three periods (one may be
    in progress, one full pass, one of slack for a descheduled thread) therefore leaves nothing to query during the capture; collectives
    issued INSIDE a capture are never listed (ProcessGroupNCCL skips workEnqueue while capturing)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        torch.cuda.synchronize()
        deadline = time.monotonic() + 3 * _WATCHDOG_PERIOD_S + 0.05
        while time.monotonic() < deadline:
            time.sleep(_WATCHDOG_PERIOD_S / 2)


def _autocast(dtype: Optional[torch.dtype]):
    return torch.autocast("cuda", dtype=dtype) if dtype is not None else contextlib.nullcontext()


def _model_device(model: torch.nn.Module) -> torch.device:
    for p in model.parameters():
        if p.is_cuda:
            return p.device
    return torch.device("cuda", torch.cuda.current_device())


def _prepared_capture_stream(model: torch.nn.Module) -> torch.cuda.Stream:
    """The stream a step object captures on - its own, so that the arrival counters its fused cross-attention launches exchange through
    (functional.ensure_sync_buffer: one buffer per device AND stream) are shared with nobody else's launches - with that buffer allocated
    and zeroed NOW: inside a capture nothing can be allocated, and a capture nobody prepared silently keeps the separate launches."""
    from . import functional as F
    device = _model_device(model)
    stream = torch.cuda.Stream(device=device)
    stream.wait_stream(torch.cuda.current_stream(device))
    if F.use_sync_exchange:
        F.ensure_sync_buffer(device, stream)
    return stream


def _check_sync_exchange(where: str) -> None:
    from . import functional as F
    F.check_sync_exchange(where)


class GraphedTrainStep:
    """check_every: every that many replays (and after the warm-up, and in close()) the step reads the error word of the in-launch hand-offs
    of the fused cross-attention kernels (functional.check_sync_exchange - one device synchronisation) and raises SyncExchangeTimeout if a
    launch was ever denied co-residency: a replayed graph cannot train through a timed-out hand-off unnoticed.  0 = only at those two points."""

    def __init__(self, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer], example_batch: Dict[str, torch.Tensor],
                 warmup: int = 3, loss_fn: Optional[Callable] = None, reducer=None, check_every: int = 128,
                 autocast: Optional[torch.dtype] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedTrainStep needs a GPU")
        self.model, self.optimizer, self.reducer = model, optimizer, reducer
        self.check_every, self._replays = int(check_every), 0
        self.autocast = autocast        # torch.bfloat16 / torch.float16: the forward runs under torch.autocast (fp32 parameters, the reference's recipe)
        _refuse_live_autograd_graphs(model)
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        self._loss_fn = loss_fn or (lambda out: out.loss)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up off the default stream: lazy init, autotuning, allocator pools
            for _ in range(max(warmup, 1)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _check_sync_exchange("GraphedTrainStep: eager warm-up steps")
        self.graph = torch.cuda.CUDAGraph()
        self._capture_stream = _prepared_capture_stream(model)
        model.zero_grad(set_to_none=True)
        # With collectives in the step the process group's watchdog THREAD is alive and polls the events of the warm-up steps' collectives
        # (hipEventQuery) whenever it wakes up; under the default "global" capture mode such a call from another thread during the capture
        # aborts the process (seen once in ~10 runs of the 1-rank RCCL test).  "thread_local" restricts the check to this thread's own calls.
        import torch.distributed as dist
        mode = "thread_local" if reducer is not None or (dist.is_available() and dist.is_initialized()) else "global"
        _let_the_watchdog_drain()
        with torch.cuda.graph(self.graph, stream=self._capture_stream, capture_error_mode=mode):
            self.loss = self._eager().detach()
        torch.cuda.synchronize()

    def close(self) -> None:
        """Final check of the in-launch hand-offs (raises SyncExchangeTimeout); idempotent."""
        if self._replays:
            self._replays = 0
            _check_sync_exchange("GraphedTrainStep.close")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.close()
        return False

    def _eager(self) -> torch.Tensor:
        self.model.zero_grad(set_to_none=True)              # gradients are re-created (not accumulated) by every backward
        with _autocast(self.autocast):
            loss = self._loss_fn(self.model(**self.static))
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()                           # the capturing stream waits for the collectives before the optimizer reads .grad
        if self.optimizer is not None:
            self.optimizer.step()
        return loss

    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        """Replay one step.  `batch` (same keys / shapes / dtypes as the example) is copied into the static inputs; None reuses them.
        Returns the static loss tensor (overwritten by the next replay)."""
        if batch is not None:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self.static[k].copy_(v, non_blocking=True)
        if self.optimizer is not None and hasattr(self.optimizer, "sync_device_hyperparams"):
            self.optimizer.sync_device_hyperparams()        # an LR scheduler may have changed group["lr"] since the capture
        if self.check_every > 0 and self._replays and self._replays % self.check_every == 0:
            _check_sync_exchange(f"GraphedTrainStep: replays {self._replays - self.check_every + 1}..{self._replays}")
        self._replays += 1
        self.graph.replay()
        return self.loss


class AutogradCuts:
    """Cut points of the autograd graph.  `cut(t)` returns a detached copy-free twin of `t` that requires grad and remembers the pair;
    `backward(loss)` then runs the backward pass segment by segment: from the loss down to the nearest cuts, then from each cut's attached
    tensor (with the gradient its twin collected) down to the next ones, newest cut first.  Every segment is its own autograd call - its own
    graph task, its own flush of the deferred weight gradients - so it can be captured into its own HIP graph, and whatever must happen
    between segments (a collective, here) happens between two calls instead of inside one.  Valid because a cut created earlier in the
    forward can only be consumed by work that flows into LATER cuts or the loss: by the time a pair is processed its twin's gradient is final."""

    def __init__(self):
        self.pairs = []
        self.armed = False      # cut() only cuts while the owning step runs a forward of its own (armed()): an ordinary training-mode
                                # model(**batch).loss.backward() on a model that still has the cuts installed sees the whole graph

    def reset(self) -> None:
        self.pairs = []

    @contextlib.contextmanager
    def arm(self):
        prev, self.armed = self.armed, True
        try:
            yield self
        finally:
            self.armed = prev

    def cut(self, t: torch.Tensor) -> torch.Tensor:
        if not (self.armed and torch.is_grad_enabled() and t.requires_grad):
            return t
        twin = t.detach().requires_grad_(True)
        self.pairs.append((t, twin))
        return twin

    def segments(self, loss: torch.Tensor):
        """The backward pass as a list of thunks, one per segment, in execution order."""
        def top():
            loss.backward()

        def below(pair):
            attached, twin = pair
            if twin.grad is not None:                       # (a twin nothing consumed - e.g. features unused under cached K / V - has no gradient)
                torch.autograd.backward(attached, twin.grad)
        return [top] + [lambda pair=pair: below(pair) for pair in reversed(self.pairs)]

    def backward(self, loss: torch.Tensor) -> None:
        for seg in self.segments(loss):
            seg()


def _merge_arena_buckets(buckets, arena):
    """The recorded buckets of one segment with those that are slices of the segment's arena replaced by ONE bucket spanning the used part of
    the arena (owners' offsets rebased); buckets outside the arena - an un-fused parameter's own gradient - stay as they are."""
    if arena is None or arena.buf is None or arena.used == 0:
        return buckets
    base, es = arena.buf.data_ptr(), arena.buf.element_size()
    end = base + arena.used * es
    inside = [b for b in buckets if b[1] and b[0].dtype == arena.buf.dtype and base <= b[0].data_ptr() < end]
    if len(inside) < 2:
        return buckets
    owners = []
    for flat, own in inside:
        shift = (flat.data_ptr() - base) // es
        owners.extend((p, off + shift, n) for p, off, n in own)
    merged, out, placed = (arena.buf[:arena.used], owners), [], False
    for b in buckets:
        if any(b is i for i in inside):
            if not placed:
                out.append(merged)
                placed = True
        else:
            out.append(b)
    return out


class PiecewiseGraphedTrainStep:
    """A training step replayed from SEVERAL HIP graphs with the gradient collectives issued eagerly between them.

        forward | backward of the top `segment_layers` gated layers (+ head) | ... | backward of the bottom ones (+ embedding) |
        resampler backward | optimizer

    Each `|` is a graph boundary.  After a backward segment has been launched, the buckets that became final in it (recorded once, while
    that segment was captured: the flat gradient buffers are static) are handed to the reducer, which all-reduces them on its side
    stream while the next segment's graph runs; the optimizer graph is launched after reducer.finish().  Nothing here needs RCCL kernels to
    be capturable, and a rank that cannot capture a piece can run that piece eagerly without changing the protocol between ranks.

    pace="host" (default): ALL sub-graphs of the step are enqueued first; then the host waits for segment k's event and only then issues
    segment k's collectives (the last segment's stay stream-ordered, so that finish() and the optimizer graph are enqueued without waiting
    for the GPU).  pace="stream" issues every segment's collectives right behind its graph launch, ordered by a cross-stream event wait.
    The stream-ordered form looks free and is not: a barrier packet that sits unsatisfied in the side queue for the length of a segment slows
    the dispatch of the compute queue's thousands of short kernels - measured on one MI355X with the exchange going through a 1-rank RCCL
    group (tools/sessions/r4/rccl_variants.py): a wait alone, nothing behind it, 41.2 -> 42.9 ms per step, independent of how many
    segments wait; host-paced 41.4.  The price is that __call__ returns when the GPU is in the last backward segment instead of
    immediately; the next step's forward graph is enqueued while the last segment and the optimizer run.
    `capture=False` runs the same segmented step with eager launches (CPU / gloo tests of the segmentation; a debugging aid on the GPU).

    segment_arena=True (default, with a reducer): the flat gradient buffers a segment produces (one per gated block, one per K / V
    projection group, the resampler's) are carved from ONE buffer per segment (functional.GradArena; sized by a pass over the warm-up
    step), and the segment's buckets travel as one collective: 11 all-reduces per step instead of 47 at the benchmark's geometry.

    overlap_optimizer=True (host pacing only; the optimizer must accept `step(only=ids, advance=bool)`, FusedAdamW with capturable=True does):
    the optimizer is captured as one sub-graph per backward segment - the parameters whose gradient became FINAL in that segment (the tied
    token embedding, which two segments accumulate into, belongs to the later one) - and segment k's update is launched on the side stream,
    behind segment k's collectives, as soon as the host has seen segment k finish: the HBM-bound AdamW sweep runs beside the latency-bound
    backward of the layers below instead of after it.  Nothing that still runs in the step reads those weights (a backward segment reads
    the weights of its own layers only; the K / V projection of a layer group has its backward inside that group's segment).  The last
    segment's update stays on the calling stream, and the calling stream waits for the side stream at the end of the call, so the usual
    stream semantics hold for whoever reads the parameters next.

    The model must offer `install_autograd_cuts(cuts, segment_layers)` (FlamingoModel / FlamingoBaseModel do: the visual features and the
    hidden state in front of every `segment_layers`-th gated layer become cut points).  Segment boundaries should coincide with the bucket
    structure the reducer sets (4 layers per weight-gradient group and per K / V projection call), which is the default."""

    def __init__(self, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer], example_batch: Dict[str, torch.Tensor],
                 warmup: int = 3, loss_fn: Optional[Callable] = None, reducer=None, segment_layers: int = 4, capture: bool = True,
                 pace: str = "host", overlap_optimizer: bool = False, segment_arena: bool = True, check_every: int = 128,
                 autocast: Optional[torch.dtype] = None):
        self.check_every, self._replays = int(check_every), 0          # see GraphedTrainStep
        self.autocast = autocast
        if pace not in ("host", "stream"):
            raise ValueError("pace must be 'host' or 'stream'")
        if overlap_optimizer and (pace != "host" or not capture):
            raise ValueError("overlap_optimizer needs pace='host' and capture=True")
        self.model, self.optimizer, self.reducer = model, optimizer, reducer
        self.pace = pace
        self.overlap_optimizer = bool(overlap_optimizer) and optimizer is not None
        self.segment_arena = bool(segment_arena) and capture and reducer is not None
        self._arenas = []              # segment_arena: one buffer per backward segment holding every flat gradient buffer the segment produces
        self._opt_pieces = []          # overlap_optimizer: one optimizer sub-graph per backward segment (None where nothing became final)
        self._side = None
        self.capture = bool(capture)
        self._loss_fn = loss_fn or (lambda out: out.loss)
        self.cuts = AutogradCuts()
        if not hasattr(model, "install_autograd_cuts"):
            raise TypeError("PiecewiseGraphedTrainStep: the model has no install_autograd_cuts(cuts, segment_layers)")
        model.install_autograd_cuts(self.cuts, segment_layers)
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        self.graphs = []
        self.segment_buckets = []
        self.host_timing: Optional[dict] = None
        if not self.capture:
            self.loss = None
            return
        try:
            self._capture(model, optimizer, reducer, warmup)
        except BaseException:
            # A capture that raised must leave nothing behind: the caller falls back to another launch mode ON THE SAME model and reducer
            # (bench.py does), where a reducer still in recording mode would never exchange a bucket again and installed cuts would make an
            # ordinary backward stop at the top segment.
            if reducer is not None and hasattr(reducer, "end_collect"):
                reducer.end_collect()
            self.close()
            raise

    def close(self) -> None:
        """Take the cut points out of the model (idempotent) and check the in-launch hand-offs of the steps that ran one last time (raises
        SyncExchangeTimeout).  The step cannot run afterwards."""
        if getattr(self.model, "install_autograd_cuts", None) is not None:
            self.model.install_autograd_cuts(None)
        self.cuts.reset()
        self.graphs, self._opt_pieces, self._opt_graph = [], [], None
        if self._replays and torch.cuda.is_available():
            self._replays = 0
            _check_sync_exchange("PiecewiseGraphedTrainStep.close")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _capture(self, model, optimizer, reducer, warmup) -> None:
        if not torch.cuda.is_available():
            raise RuntimeError("PiecewiseGraphedTrainStep(capture=True) needs a GPU")
        _refuse_live_autograd_graphs(model)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        _check_sync_exchange("PiecewiseGraphedTrainStep: eager warm-up steps")
        arena_sizes = self._size_arenas() if self.segment_arena else None
        import torch.distributed as dist
        mode = "thread_local" if reducer is not None or (dist.is_available() and dist.is_initialized()) else "global"
        _let_the_watchdog_drain()
        pool = torch.cuda.graph_pool_handle()
        cap_stream = _prepared_capture_stream(model)        # every piece is captured on this one stream: one set of arrival counters

        def piece(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, stream=cap_stream, capture_error_mode=mode):
                out = fn()
            self.graphs.append(g)
            return out

        model.zero_grad(set_to_none=True)
        self.cuts.reset()
        def fwd():
            with _autocast(self.autocast):
                return self._loss_fn(self.model(**self.static))

        with self.cuts.arm():
            loss = piece(fwd)
        self.loss = loss.detach()
        trainable = [p for p in model.parameters() if p.requires_grad]
        last_touched, seen = {}, {}
        from . import functional as F
        for k, seg in enumerate(self.cuts.segments(loss)):
            arena = None
            if arena_sizes is not None and k < len(arena_sizes) and arena_sizes[k] is not None:
                (dt, dev), n = arena_sizes[k]
                arena = F.GradArena()
                arena.buf = torch.zeros(n, dtype=dt, device=dev)     # a plain tensor this object keeps alive: static for the graphs; pads stay zero
            self._arenas.append(arena)
            if reducer is not None:
                reducer.begin_collect()
            prev = F.set_grad_arena(arena)
            buckets = []
            try:
                piece(seg)
            finally:
                F.set_grad_arena(prev)
                if reducer is not None:
                    buckets = reducer.end_collect()      # also when the capture raised: the reducer must not stay in recording mode
            self.segment_buckets.append(_merge_arena_buckets(buckets, arena))
            for p in trainable:                 # which segment wrote (or accumulated into) which gradient
                if p.grad is not None:
                    mark = (p.grad.data_ptr(), p.grad._version)
                    if seen.get(id(p)) != mark:
                        seen[id(p)] = mark
                        last_touched[id(p)] = k
        if reducer is not None:        # an un-fused parameter (the tied token embedding) takes part in two segments: exchange it after the last one
            seen = set()
            for buckets in reversed(self.segment_buckets):
                keep = [b for b in buckets if b[1] or b[0].data_ptr() not in seen]
                seen.update(b[0].data_ptr() for b in buckets if not b[1])
                buckets[:] = keep
        self._opt_graph = None
        if self.overlap_optimizer:
            n_seg = len(self.segment_buckets)
            final_in = [frozenset(i for i, k in last_touched.items() if k == seg) for seg in range(n_seg)]
            first = True
            for ids in final_in:
                if not ids:
                    self._opt_pieces.append(None)
                    continue
                piece(lambda ids=ids, first=first: optimizer.step(only=ids, advance=first))
                self._opt_pieces.append(self.graphs.pop())
                first = False
            self._side = reducer.stream if (reducer is not None and getattr(reducer, "cuda", False)) else torch.cuda.Stream()
        elif optimizer is not None:
            piece(optimizer.step)
            self._opt_graph = self.graphs.pop()
        del loss
        self.cuts.reset()               # the pairs' memory belongs to the graphs' pool; the Python references are not needed any more
        torch.cuda.synchronize()

    def _size_arenas(self):
        """One more eager pass over the segments with a sizing arena installed: what each segment asks `functional._flat_grads` for.
        Returns, per segment, ((dtype, device), elements) of its dominant request class, or None.  (No collectives, no optimizer: the
        reducer is told to skip this backward.)"""
        from . import functional as F
        self.model.zero_grad(set_to_none=True)
        self.cuts.reset()
        sizes = []
        rng = torch.cuda.get_rng_state()                    # (the pass must not shift the dropout stream of the steps that follow)
        ctx = self.reducer.no_sync() if self.reducer is not None and hasattr(self.reducer, "no_sync") else contextlib.nullcontext()
        with ctx:
            with self.cuts.arm(), _autocast(self.autocast):
                loss = self._loss_fn(self.model(**self.static))
            for seg in self.cuts.segments(loss):
                arena = F.GradArena()
                prev = F.set_grad_arena(arena)
                try:
                    seg()
                finally:
                    F.set_grad_arena(prev)
                sizes.append(max(arena.need.items(), key=lambda kv: kv[1]) if arena.need else None)
        self.cuts.reset()
        self.model.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        torch.cuda.set_rng_state(rng)
        return sizes

    def _eager(self) -> torch.Tensor:
        """The segmented step with eager launches (warm-up, capture=False): collectives are issued from inside backward as usual, except
        that un-fused parameters, whose gradient is accumulated by two segments, are exchanged after the last one (reducer.finish())."""
        self.model.zero_grad(set_to_none=True)
        self.cuts.reset()
        with self.cuts.arm(), _autocast(self.autocast):
            loss = self._loss_fn(self.model(**self.static))
        if self.reducer is not None:
            self.reducer.defer_loose = True
        try:
            self.cuts.backward(loss)
        finally:
            if self.reducer is not None:
                self.reducer.defer_loose = False
        self.cuts.reset()
        if self.reducer is not None:
            self.reducer.finish()
        if self.optimizer is not None:
            self.optimizer.step()
        return loss.detach()

    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        if batch is not None:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self.static[k].copy_(v, non_blocking=True)
        if not self.capture:
            self.loss = self._eager()
            return self.loss
        if self.optimizer is not None and hasattr(self.optimizer, "sync_device_hyperparams"):
            self.optimizer.sync_device_hyperparams()
        if self.check_every > 0 and self._replays and self._replays % self.check_every == 0:
            _check_sync_exchange(f"PiecewiseGraphedTrainStep: replays {self._replays - self.check_every + 1}..{self._replays}")
        self._replays += 1
        ht = self.host_timing                               # None, or a dict that accumulates the HOST seconds of each kind of call
        t0 = time.perf_counter() if ht is not None else 0.0
        if self.overlap_optimizer:
            return self._replay_overlapped(ht, t0)
        self.graphs[0].replay()
        host_paced = self.pace == "host" and self.reducer is not None and getattr(self.reducer, "cuda", False) and getattr(self.reducer, "active", False)
        if host_paced:
            events = []
            for g in self.graphs[1:]:
                g.replay()
                e = torch.cuda.Event()
                e.record()
                events.append(e)
            if ht is not None:
                t1 = time.perf_counter(); ht["graph_launches"] = ht.get("graph_launches", 0.0) + t1 - t0; t0 = t1
            with_buckets = [i for i, b in enumerate(self.segment_buckets) if b]
            for i in with_buckets:
                if i != with_buckets[-1]:
                    events[i].synchronize()                 # the host sees segment i finish; nothing waits inside a hardware queue
                    if ht is not None:
                        t1 = time.perf_counter(); ht["host_waits"] = ht.get("host_waits", 0.0) + t1 - t0; t0 = t1
                    self.reducer.reduce_buckets(self.segment_buckets[i], producers_done=True)
                else:
                    self.reducer.reduce_buckets(self.segment_buckets[i])
                if ht is not None:
                    t1 = time.perf_counter(); ht["collectives"] = ht.get("collectives", 0.0) + t1 - t0; t0 = t1
        else:
            for g, buckets in zip(self.graphs[1:], self.segment_buckets):
                g.replay()
                if ht is not None:
                    t1 = time.perf_counter(); ht["graph_launches"] = ht.get("graph_launches", 0.0) + t1 - t0; t0 = t1
                if buckets:                                     # eager collectives on the reducer's side stream, behind this segment
                    self.reducer.reduce_buckets(buckets)
                if ht is not None:
                    t1 = time.perf_counter(); ht["collectives"] = ht.get("collectives", 0.0) + t1 - t0; t0 = t1
        if self.reducer is not None:
            self.reducer.finish()
        if ht is not None:
            t1 = time.perf_counter(); ht["finish"] = ht.get("finish", 0.0) + t1 - t0; t0 = t1
        if self._opt_graph is not None:
            self._opt_graph.replay()
        if ht is not None:
            ht["graph_launches"] = ht.get("graph_launches", 0.0) + time.perf_counter() - t0
            ht["steps"] = ht.get("steps", 0) + 1
        return self.loss

    def _replay_overlapped(self, ht, t0) -> torch.Tensor:
        """overlap_optimizer=True: forward | all backward segments enqueued | per segment, once the host has seen it finish: its collectives,
        then its optimizer sub-graph, on the side stream | last segment: collectives, finish(), update on the calling stream."""
        red = self.reducer if (self.reducer is not None and getattr(self.reducer, "active", False)) else None      # (a gloo reducer exchanges synchronously)
        main = torch.cuda.current_stream()
        self.graphs[0].replay()                              # (the previous call ended with this stream waiting for the side stream)
        events = []
        for g in self.graphs[1:]:
            g.replay()
            e = torch.cuda.Event()
            e.record()
            events.append(e)
        if ht is not None:
            t1 = time.perf_counter(); ht["graph_launches"] = ht.get("graph_launches", 0.0) + t1 - t0; t0 = t1
        work = [i for i in range(len(events)) if self._opt_pieces[i] is not None or (red is not None and self.segment_buckets[i])]
        progress = None
        for i in work[:-1]:
            events[i].synchronize()                         # the host sees segment i finish; nothing waits inside a hardware queue
            if ht is not None:
                t1 = time.perf_counter(); ht["host_waits"] = ht.get("host_waits", 0.0) + t1 - t0; t0 = t1
            if red is not None and self.segment_buckets[i]:
                if getattr(red, "cuda", False):
                    red.reduce_buckets(self.segment_buckets[i], producers_done=True)        # on the reducer's stream = self._side
                else:
                    with torch.cuda.stream(self._side):     # a synchronous (gloo) exchange orders itself against the CURRENT stream
                        red.reduce_buckets(self.segment_buckets[i], producers_done=True)
            if self._opt_pieces[i] is not None:
                with torch.cuda.stream(self._side):         # behind segment i's collectives: same stream
                    self._opt_pieces[i].replay()
                    progress = torch.cuda.Event()
                    progress.record()
            if ht is not None:
                t1 = time.perf_counter(); ht["collectives"] = ht.get("collectives", 0.0) + t1 - t0; t0 = t1
        if work:
            i = work[-1]
            if red is not None and self.segment_buckets[i]:
                red.reduce_buckets(self.segment_buckets[i])
            if red is not None:
                red.finish()
            if progress is not None:
                main.wait_event(progress)                    # the first partial update advanced the step counters; and: end-of-call stream semantics
            if self._opt_pieces[i] is not None:
                self._opt_pieces[i].replay()
        elif red is not None:
            red.finish()
        if ht is not None:
            ht["finish"] = ht.get("finish", 0.0) + time.perf_counter() - t0
            ht["steps"] = ht.get("steps", 0) + 1
        return self.loss
