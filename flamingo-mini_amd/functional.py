"""torch.autograd.Function wrappers around the C ABI (one Function per fwd/bwd pair) plus thin wrappers of the
primitive kernels used by the parity tests.

Nothing here computes: every op below is a call into libflamingo_fusion.so on the current HIP stream; CPU tensors or a
missing library raise FusionLibraryError.  (The CPU plumbing tests monkeypatch `resampler`, `xattn_block` and
`text_time` of this module from tests/oracle_backend.py - there is no alternative path in the package itself.)
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch

from . import ffi

# called as cb(flat, owners) right after the kernels that fill a fused module's flat gradient buffer are enqueued;
# owners = [(parameter, offset in elements, numel)]: which slice of `flat` autograd was handed as that parameter's gradient
_grad_ready_callbacks: list = []


def _owners(params, flat=None):
    offs, _ = _flat_offsets(params)
    return [(p, o, p.numel()) for p, o in zip(params, offs)]


def _announce(flat, params) -> None:
    if _grad_ready_callbacks:
        owners = _owners(params)
        for cb in _grad_ready_callbacks:
            cb(flat, owners)


def add_grad_ready_callback(fn: Callable[[torch.Tensor], None]) -> None:
    _grad_ready_callbacks.append(fn)


def remove_grad_ready_callback(fn) -> None:
    if fn in _grad_ready_callbacks:
        _grad_ready_callbacks.remove(fn)


def _graph_task_id() -> int:
    """Id of the autograd engine's running backward pass; -1 outside any backward."""
    get = getattr(torch._C, "_current_graph_task_id", None)
    return get() if get is not None else -1


class _WgradQueue:
    """Deferred, grouped weight gradients of the cross-attention blocks (ff_xattn_block_bwd_kv_data / ff_xattn_wgrad_grouped).

    A block's backward enqueues only its data-gradient chain and parks the operands of its four weight-gradient GEMMs in a `stash`;
    whenever `group` same-shaped blocks are waiting (and once more at the end of the backward pass, through the autograd
    engine's queue_callback) their weight gradients are computed by grouped launches that fill the chip.  The gradient tensors were
    already handed to autograd (views of the block's flat buffer): they are FILLED by the flush, which is enqueued on the same stream
    before backward() returns, so everything that consumes .grad afterwards is ordered behind it.  Blocks whose parameters already
    hold a .grad (gradient accumulation: autograd would ADD the returned tensor right away) do not defer.
    The queue keeps raw addresses, never the gradient tensors themselves: AccumulateGrad adopts an incoming gradient without a copy
    only while nobody else references it (otherwise it clones - at that moment, i.e. before the flush has filled it).  The final
    callback verifies that adoption really happened for every deferred gradient and repairs the ones autograd copied after all.
    The queue's state belongs to ONE backward pass (the autograd engine's graph task, torch._C._current_graph_task_id()): the engine
    does not run queue_callback callbacks of a pass that raised (OOM, a bad label, an exception in a hook), so whatever such a pass
    left behind - raw gradient addresses of flat buffers that are gone by now - must never be run or grouped with fresh entries.
    State is therefore kept per graph task, and the leftovers of dead passes are dropped by the next forward that runs outside any
    backward (forget_dead_passes).  A block whose parameters take part TWICE in one pass (module reuse, activation-checkpoint
    recompute) must not defer the second time: the engine sums the two contributions the moment the second one is returned, so the
    first is completed on the spot and the second runs the plain, non-deferred backward (seen_in_this_pass).
    Whether a block defers, and how many blocks a grouped launch batches, is the CALLER's choice (xattn_block(..., wgrad=(defer, group)):
    GatedCrossAttentionBlock.defer_wgrad / .wgrad_group, set for a whole model through FlamingoBaseModel.set_launch_structure); `group`
    below is only the default for callers that pass none, `enabled = False` switches the deferral off for everybody.  No environment
    variable is read."""

    class _Pass:
        def __init__(self):
            self.pending: list = []
            self.done: list = []
            self.deferred_ids: set = set()   # parameters with a deferred gradient in this pass
            self.summed_ids: set = set()     # ... that received a second contribution: autograd holds their SUM, nothing to repair

    def __init__(self):
        # the per-pass bookkeeping needs the engine's graph-task id; a torch build without that (private) accessor runs the plain,
        # non-deferred backward instead of guessing which pass an entry belongs to
        self.enabled = hasattr(torch._C, "_current_graph_task_id")
        # Blocks per grouped launch.  At flamingo-mini's size a block contributes 400 tiles of 128 x 128 per product and the chip holds 512
        # workgroups at a time: 4 blocks = 3.1 rounds (the last one a quarter full), 12 blocks = 9.4 - measured 35.9 -> 35.5 ms per step
        # (weight-gradient launches 704 -> 830 TFLOP/s).  Data-parallel reducers set 4 ON THEIR MODEL so that gradient buckets keep becoming
        # final - and their exchange keeps starting - every four layers of backward (data_parallel._bucket_launch_structure).
        self.group = ffi.WGRAD_GROUP_MAX
        self._passes: dict = {}              # graph-task id -> _Pass

    @property
    def pending(self) -> list:
        return [e for st in self._passes.values() for e in st.pending]

    def forget_dead_passes(self) -> None:
        """Called from a forward: outside any backward pass, whatever the queue still holds belongs to passes that raised."""
        if self._passes and _graph_task_id() < 0:
            self._passes.clear()

    def seen_in_this_pass(self, params) -> bool:
        """True if one of `params` already has a deferred gradient in the running pass (the caller must then not defer again)."""
        st = self._passes.get(_graph_task_id())
        if st is None:
            return False
        hit = [id(p) for p in params if id(p) in st.deferred_ids]
        if hit:
            st.summed_ids.update(hit)
            self._flush_pending(st)          # the first contribution must be complete before autograd adds the second one to it
        return bool(hit)

    def push(self, entry) -> None:
        task = _graph_task_id()
        st = self._passes.get(task)
        if st is None:
            st = self._passes[task] = self._Pass()
            torch.autograd.Variable._execution_engine.queue_callback(lambda: self.flush(task))
        st.pending.append(entry)
        st.deferred_ids.update(id(p) for p in entry["wparams"])
        same = [e for e in st.pending if e["key"] == entry["key"]]
        if len(same) >= entry["group"]:
            self._run(st, same)

    def _flush_pending(self, st) -> None:
        while st.pending:
            key = st.pending[0]["key"]
            self._run(st, [e for e in st.pending if e["key"] == key][: st.pending[0]["group"]])

    def flush(self, task) -> None:
        st = self._passes.get(task)
        if st is None:
            return
        try:
            self._flush_pending(st)
            for e in st.done:        # every AccumulateGrad of this backward pass has run by now
                for p, (off, n) in zip(e["wparams"], e["wslices"]):
                    if p.grad is None or id(p) in st.summed_ids:
                        continue                                 # retain_graph / autograd.grad without accumulation; or a sum of two uses
                    filled = e["flat"][off:off + n].view(p.shape)
                    if p.grad.data_ptr() != filled.data_ptr():   # autograd cloned the (then unfilled) tensor instead of adopting it
                        p.grad.copy_(filled)
        finally:
            self._passes.pop(task, None)

    def _run(self, st, group) -> None:
        lib = ffi.lib()
        ids = {id(e) for e in group}
        st.pending = [e for e in st.pending if id(e) not in ids]
        e0 = group[0]
        desc, dev = e0["desc"], e0["device"]
        ws = _empty_bytes(lib.ff_xattn_wgrad_workspace_bytes(desc), dev)
        params = [p for e in group for p in e["params"]]
        n = ffi.XATTN_PARAMS
        grad_ptrs = (ffi.C.c_void_p * (n * len(group)))()
        for i, e in enumerate(group):
            for j in range(n):
                grad_ptrs[i * n + j] = e["grad_ptrs"][j]
        ffi.check(lib.ff_xattn_wgrad_grouped(desc, len(group), ffi.ptr_array([e["dout"] for e in group]), ffi.ptr_array([e["saved"] for e in group]),
                                             e0["saved"].numel(), ffi.ptr_array([e["stash"] for e in group]), e0["stash"].numel(),
                                             ffi.ptr_array(params), grad_ptrs, ws.data_ptr(), ws.numel(), ffi.stream_handle(dev)),
                  "ff_xattn_wgrad_grouped")
        for e in group:
            _announce(e["flat"], e["own"])
        st.done.extend(group)


_wgrad_queue = _WgradQueue()


def _flat_offsets(params: Sequence[torch.Tensor]):
    align = 16 // params[0].element_size()
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + align - 1) // align * align
    return offs, (total + 1023) // 1024 * 1024       # whole buffer: a multiple of 1024 elements, so it splits evenly over 1/2/4/8 ranks (reduce-scatter)


class GradArena:
    """One buffer for ALL the flat gradient buffers a stretch of backward produces (graphs.PiecewiseGraphedTrainStep: one backward segment),
    so that the stretch's gradients travel in ONE collective instead of one per fused module.  Without a buffer it only adds up what the
    stretch asks for (`need`, by dtype and device): the sizing pass.  With one, `take` hands out consecutive slices; a request that does
    not fit (or is of another dtype) falls back to its own allocation."""

    def __init__(self):
        self.need, self.buf, self.used = {}, None, 0

    def take(self, total: int, dtype, device):
        if self.buf is None:
            self.need[(dtype, device)] = self.need.get((dtype, device), 0) + total
            return None
        if self.buf.dtype != dtype or self.buf.device != device or self.used + total > self.buf.numel():
            return None
        out = self.buf[self.used:self.used + total]
        self.used += total
        return out


_grad_arena: Optional[GradArena] = None


def set_grad_arena(arena: Optional[GradArena]) -> Optional[GradArena]:
    """Install (or remove, None) the arena the flat gradient buffers are carved from; returns the previous one.  Process-wide on purpose:
    backward runs on the autograd engine's thread."""
    global _grad_arena
    prev, _grad_arena = _grad_arena, arena
    return prev


def _flat_grads(params: Sequence[torch.Tensor]):
    """One contiguous buffer holding every parameter gradient of a fused module (each slice 16-byte aligned).  Autograd
    adopts the slices as `param.grad` without copying, so the buffer doubles as a ready-made all-reduce bucket."""
    offs, total = _flat_offsets(params)
    flat = _grad_arena.take(total, params[0].dtype, params[0].device) if _grad_arena is not None else None
    if flat is None:
        flat = torch.empty(total, dtype=params[0].dtype, device=params[0].device)   # alignment pads are never read
    return flat, [flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, params)]


def _empty_bytes(n: int, device) -> torch.Tensor:
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)


def autocast_compute_dtype(ref: torch.Tensor) -> Optional[torch.dtype]:
    """Mixed-precision training the way the reference's recipe does it (`training/train.sh:24`: HF Trainer --fp16 = torch.autocast over fp32
    parameters): inside a CUDA autocast region the drop-in modules hand the library copies of their parameters in the dtype its kernels
    compute in - bfloat16 under bf16 autocast; float32 under fp16 autocast (the library has no fp16 kernels: that region runs its exact
    fp32 kernels, correct and slow) - through differentiable casts, so the gradients reach the fp32 parameters.  None outside autocast."""
    if not ref.is_cuda:
        return None
    try:                                    # torch >= 2.4 spells it per device type
        on, adt = torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda")
    except TypeError:
        on, adt = torch.is_autocast_enabled(), torch.get_autocast_gpu_dtype()
    if not on:
        return None
    return torch.bfloat16 if adt == torch.bfloat16 else torch.float32


def autocast_params(params: Sequence[torch.Tensor], dtype: torch.dtype):
    return [p if p.dtype == dtype else p.to(dtype) for p in params]


def _same_dtype(ref: torch.Tensor, tensors: Sequence[torch.Tensor], what: str) -> None:
    for t in tensors:
        if t.dtype != ref.dtype:
            raise ffi.FusionLibraryError(f"{what}: parameters are {t.dtype} but activations are {ref.dtype}; "
                                         "cast the module (module.to(dtype)) — the kernels compute in one dtype")


# ----------------------------------------------------------------------------------------------------
# text_time
# ----------------------------------------------------------------------------------------------------
def text_time(media_locations: torch.Tensor) -> torch.Tensor:
    """cumsum(media_locations, -1) as int32 (gated_cross_attention.py:97), computed once per step."""
    ffi.require_cuda(media_locations)
    ml = media_locations
    if ml.dtype == torch.bool:
        ml = ml.view(torch.uint8)
    if ml.dtype not in (torch.int64, torch.int32, torch.uint8):
        ml = ml.to(torch.int64)
    ml = ml.contiguous()
    b, n = ml.shape
    out = torch.empty((b, n), dtype=torch.int32, device=ml.device)
    ffi.check(ffi.lib().ff_text_time(b, n, ml.data_ptr(), ml.element_size(), out.data_ptr(), ffi.stream_handle(ml.device)), "ff_text_time")
    return out


# ----------------------------------------------------------------------------------------------------
# PerceiverResampler
# ----------------------------------------------------------------------------------------------------
# cfg = (depth, heads, dim_head, num_latents, num_time_embeds, ff_mult, act)
def _resampler_desc(x_f: torch.Tensor, cfg) -> ffi.ResamplerDesc:
    depth, heads, dim_head, num_latents, nte, ff_mult, act = cfg
    b, T, v, d = x_f.shape
    return ffi.ResamplerDesc(ffi.dtype_code(x_f.dtype), b, T, v, d, depth, heads, dim_head, num_latents, nte, ff_mult, ffi.ACTS[act])


class _ResamplerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_f: torch.Tensor, cfg, *params: torch.Tensor):
        lib = ffi.lib()
        x_f = x_f.contiguous()
        params = tuple(p.contiguous() for p in params)
        desc = _resampler_desc(x_f, cfg)
        dev = x_f.device
        saved = _empty_bytes(lib.ff_resampler_saved_bytes(desc), dev)
        scratch = _empty_bytes(lib.ff_resampler_scratch_bytes(desc), dev)
        out = torch.empty((x_f.shape[0], cfg[3], x_f.shape[3]), dtype=x_f.dtype, device=dev)
        ffi.check(lib.ff_resampler_fwd(desc, x_f.data_ptr(), ffi.ptr_array(params), out.data_ptr(), saved.data_ptr(), saved.numel(),
                                       scratch.data_ptr(), scratch.numel(), ffi.stream_handle(dev)), "ff_resampler_fwd")
        ctx.cfg = cfg
        ctx.save_for_backward(x_f, saved, *params)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        lib = ffi.lib()
        x_f, saved, *params = ctx.saved_tensors
        desc = _resampler_desc(x_f, ctx.cfg)
        dev = x_f.device
        dout = dout.contiguous()
        flat, grads = _flat_grads(params)
        dx_f = torch.empty_like(x_f) if ctx.needs_input_grad[0] else None
        scratch = _empty_bytes(lib.ff_resampler_scratch_bytes(desc), dev)
        ffi.check(lib.ff_resampler_bwd(desc, x_f.data_ptr(), ffi.ptr_array(params), dout.data_ptr(), saved.data_ptr(), saved.numel(),
                                       ffi.ptr_array(grads), ffi.ptr(dx_f), scratch.data_ptr(), scratch.numel(), ffi.stream_handle(dev)),
                  "ff_resampler_bwd")
        _announce(flat, params)
        return (dx_f, None, *grads)


def resampler(x_f: torch.Tensor, params: Sequence[torch.Tensor], cfg) -> torch.Tensor:
    """x_f (b, T, v, d) -> (b, num_latents, d).  `params` in the order documented in flamingo_fusion.h."""
    ffi.require_cuda(x_f, *params)
    _same_dtype(x_f, params, "PerceiverResampler")
    assert len(params) == ffi.RESAMPLER_GLOBAL_PARAMS + ffi.RESAMPLER_LAYER_PARAMS * cfg[0]
    return _ResamplerFn.apply(x_f, tuple(cfg), *params)


# ---- the resampler driven LAYER BY LAYER (ff_resampler_layer_* + prologue / epilogue; perceiver_resampler.py:181-183) ----
# One autograd node per layer: a layer's twelve parameter gradients are final - and announced as a bucket of their own - when that layer's
# backward has been enqueued, so under data parallelism layer 5's gradients leave while layer 4 runs backward (the stack-level call hands
# over all 63 M resampler gradients at once, at the very end of backward).  d x_f is summed over the layers in ONE buffer that the layers'
# backward calls accumulate into; it travels outside autograd (`_RsPass`), and the prologue node - whose token LAYER 0 consumes: layer 0's
# backward is the last of the layers' (d x flows 5 -> 0), also when every layer is a backward segment of its own (graphs.AutogradCuts between
# the layers: one autograd call per segment, each with its own dependency count) - turns it into d time_pos_emb (and hands it on as d x_f),
# and the batch-sum of layer 0's input gradient into d latents.
class _RsPass:
    """What the layer-wise backward of ONE resampler forward shares: the d x_f accumulator and layer 0's input gradient."""

    def __init__(self):
        self.dxf: Optional[torch.Tensor] = None
        self.dx0: Optional[torch.Tensor] = None


class _RsPrologueFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_f, tpe, latents, cfg, rs_pass):
        lib = ffi.lib()
        desc = _resampler_desc(x_f, cfg)
        pro = _empty_bytes(lib.ff_resampler_prologue_saved_bytes(desc), x_f.device)
        ffi.check(lib.ff_resampler_prologue_fwd(desc, x_f.data_ptr(), tpe.data_ptr(), pro.data_ptr(), pro.numel(), ffi.stream_handle(x_f.device)),
                  "ff_resampler_prologue_fwd")
        ctx.cfg, ctx.rs_pass = cfg, rs_pass
        ctx.save_for_backward(x_f, tpe, latents)
        token = torch.zeros(1, dtype=torch.float32, device=x_f.device)       # what the layers consume: orders this node's backward behind theirs
        ctx.mark_non_differentiable(pro)
        return token, pro

    @staticmethod
    def backward(ctx, _dtoken, _dpro):
        lib = ffi.lib()
        x_f, tpe, latents = ctx.saved_tensors
        rp = ctx.rs_pass
        if rp.dxf is None or rp.dx0 is None:
            raise ffi.FusionLibraryError("resampler (layer-wise): the prologue's backward ran before the layers' - the autograd graph was cut between them")
        desc = _resampler_desc(x_f, ctx.cfg)
        flat, grads = _flat_grads([latents, tpe])
        scratch = _empty_bytes(lib.ff_resampler_layer_scratch_bytes(desc), x_f.device)
        ffi.check(lib.ff_resampler_prologue_bwd(desc, rp.dx0.data_ptr(), rp.dxf.data_ptr(), grads[0].data_ptr(), grads[1].data_ptr(), scratch.data_ptr(),
                                                scratch.numel(), ffi.stream_handle(x_f.device)), "ff_resampler_prologue_bwd")
        _announce(flat, [latents, tpe])
        dxf = rp.dxf if ctx.needs_input_grad[0] else None
        rp.dxf = rp.dx0 = None
        return dxf, grads[1], grads[0], None, None


class _RsLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, token, x_f, tpe, pro, cfg, rs_pass, first, last, *params):
        lib = ffi.lib()
        desc = _resampler_desc(x_f, cfg)
        dev = x_f.device
        x = x.contiguous()
        params = tuple(p.contiguous() for p in params)
        saved = _empty_bytes(lib.ff_resampler_layer_saved_bytes(desc), dev)
        scratch = _empty_bytes(lib.ff_resampler_layer_scratch_bytes(desc), dev)
        out = torch.empty((x_f.shape[0], cfg[3], x_f.shape[3]), dtype=x_f.dtype, device=dev)
        ffi.check(lib.ff_resampler_layer_fwd(desc, x_f.data_ptr(), tpe.data_ptr(), pro.data_ptr(), pro.numel(), x.data_ptr(), 1 if first else 0,
                                             ffi.ptr_array(params), out.data_ptr(), saved.data_ptr(), saved.numel(), scratch.data_ptr(), scratch.numel(),
                                             ffi.stream_handle(dev)), "ff_resampler_layer_fwd")
        ctx.cfg, ctx.rs_pass, ctx.first, ctx.last = cfg, rs_pass, first, last
        ctx.save_for_backward(x, x_f, tpe, pro, saved, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = ffi.lib()
        x, x_f, tpe, pro, saved, *params = ctx.saved_tensors
        desc = _resampler_desc(x_f, ctx.cfg)
        dev = x_f.device
        rp = ctx.rs_pass
        dout = dout.contiguous()
        flat, grads = _flat_grads(params)
        accumulate = rp.dxf is not None                      # the first layer_bwd call of a pass (the LAST layer) overwrites
        if rp.dxf is None:
            rp.dxf = torch.empty_like(x_f)
        dx_in = torch.empty_like(dout)
        scratch = _empty_bytes(lib.ff_resampler_layer_scratch_bytes(desc), dev)
        ffi.check(lib.ff_resampler_layer_bwd(desc, x_f.data_ptr(), tpe.data_ptr(), pro.data_ptr(), pro.numel(), x.data_ptr(), 1 if ctx.first else 0,
                                             ffi.ptr_array(params), dout.data_ptr(), saved.data_ptr(), saved.numel(), ffi.ptr_array(grads), dx_in.data_ptr(),
                                             rp.dxf.data_ptr(), 1 if accumulate else 0, scratch.data_ptr(), scratch.numel(), ffi.stream_handle(dev)),
                  "ff_resampler_layer_bwd")
        _announce(flat, params)
        if ctx.first:                                        # d (broadcast latents): summed over the batch by the prologue's backward, which
            rp.dx0 = dx_in                                   # this layer's token gradient sets off
            return (None, torch.zeros(1, dtype=torch.float32, device=dev), None, None, None, None, None, None, None, *grads)
        return (dx_in, None, None, None, None, None, None, None, None, *grads)


class _RsEpilogueFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, x_f, cfg):
        lib = ffi.lib()
        desc = _resampler_desc(x_f, cfg)
        x = x.contiguous()
        epi = _empty_bytes(lib.ff_resampler_epilogue_saved_bytes(desc), x.device)
        out = torch.empty_like(x)
        ffi.check(lib.ff_resampler_epilogue_fwd(desc, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), epi.data_ptr(), epi.numel(),
                                                ffi.stream_handle(x.device)), "ff_resampler_epilogue_fwd")
        ctx.cfg, ctx.xf_shape = cfg, tuple(x_f.shape)
        ctx.save_for_backward(x, gamma, beta, epi)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = ffi.lib()
        x, gamma, beta, epi = ctx.saved_tensors
        b, T, v, d = ctx.xf_shape
        depth, heads, dim_head, num_latents, nte, ff_mult, act = ctx.cfg
        desc = ffi.ResamplerDesc(ffi.dtype_code(x.dtype), b, T, v, d, depth, heads, dim_head, num_latents, nte, ff_mult, ffi.ACTS[act])
        dout = dout.contiguous()
        flat, grads = _flat_grads([gamma, beta])
        dx = torch.empty_like(x)
        scratch = _empty_bytes(lib.ff_resampler_layer_scratch_bytes(desc), x.device)
        ffi.check(lib.ff_resampler_epilogue_bwd(desc, dout.data_ptr(), x.data_ptr(), gamma.data_ptr(), epi.data_ptr(), epi.numel(), dx.data_ptr(),
                                                grads[0].data_ptr(), grads[1].data_ptr(), scratch.data_ptr(), scratch.numel(), ffi.stream_handle(x.device)),
                  "ff_resampler_epilogue_bwd")
        _announce(flat, [gamma, beta])
        return dx, grads[0], grads[1], None, None


def resampler_layerwise(x_f: torch.Tensor, params: Sequence[torch.Tensor], cfg, cut: Optional[Callable] = None) -> torch.Tensor:
    """The same function as `resampler`, one library call - and one autograd node, one gradient bucket - per layer.  `cut`
    (graphs.AutogradCuts.cut) is applied to the latents between the layers, which makes every layer its own backward segment."""
    ffi.require_cuda(x_f, *params)
    _same_dtype(x_f, params, "PerceiverResampler")
    depth = cfg[0]
    assert len(params) == ffi.RESAMPLER_GLOBAL_PARAMS + ffi.RESAMPLER_LAYER_PARAMS * depth
    cfg = tuple(cfg)
    x_f = x_f.contiguous()
    latents, tpe, gamma, beta = (p.contiguous() for p in params[:4])
    rp = _RsPass()
    token, pro = _RsPrologueFn.apply(x_f, tpe, latents, cfg, rp)
    x = latents
    for i in range(depth):
        lp = params[4 + 12 * i: 4 + 12 * (i + 1)]
        x = _RsLayerFn.apply(x, token if i == 0 else None, x_f, tpe, pro, cfg, rp, i == 0, i == depth - 1, *lp)
        if cut is not None and i + 1 < depth:
            x = cut(x)
    return _RsEpilogueFn.apply(x, gamma, beta, x_f, cfg)


# ----------------------------------------------------------------------------------------------------
# GatedCrossAttentionBlock
# ----------------------------------------------------------------------------------------------------
# ff_xattn_desc.sync: the arrival counters through which the (sample, head) workgroups of the fused cross-attention kernels exchange their
# tiles inside a launch (to_out / d LN(y) without a launch of their own).  The counters are per sample, not per call, so every launch that
# shares a buffer must be ordered behind the previous one: a buffer therefore belongs to ONE stream of one device - the stream that was
# current when it was first asked for - and a call on any other stream gets its own (eval or decode beside training, a second model, a
# side-stream experiment: no interleaved arrivals).  The zero fill is enqueued on the owning stream, i.e. ahead of the first launch that
# counts on it.  A forward call records the buffer it used in its autograd context: the backward pass (the autograd engine restores the
# forward's stream) and the deferred weight-gradient flush see the same decision the forward made, whatever the switch says by then.
_sync_buffers: dict = {}        # (device index, stream handle) -> uint8 tensor of ff_xattn_sync_bytes()
_sync_probes: dict = {}         # (device index, stream handle) -> (pinned int32 tensor, event) of the last non-blocking status probe
use_sync_exchange = True        # False: every block keeps its separate to_out / d LN(y) launches (A/B timing, debugging)
_STATUS_WORD = None             # element index of the status word in an int32 view of a sync buffer


class SyncExchangeTimeout(ffi.FusionLibraryError):
    """An arrival wait inside a fused cross-attention launch gave up (a sample's workgroups were denied co-residency): the outputs of
    that launch - and of everything computed from them - are invalid."""


def _status_word() -> int:
    global _STATUS_WORD
    if _STATUS_WORD is None:
        words = int(ffi.lib().ff_xattn_sync_bytes()) // 4          # include/flamingo_fusion.h: (4 * FF_XATTN_SYNC_SLOTS + 64) words,
        _STATUS_WORD = 2 * ((words - 64) // 4)                      # the status word follows the first two counter banks
    return _STATUS_WORD


def _sync_key(device, stream=None):
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(index) if stream is None else stream
    return index, int(stream.cuda_stream)


def ensure_sync_buffer(device, stream=None) -> Optional[torch.Tensor]:
    """The sync buffer of (`device`, `stream`; default: the current stream), allocated and zeroed on first use.  Nothing can be allocated
    while that stream is capturing (the zero fill must not become a graph node, the memory must not come from a graph pool): code that
    captures HIP graphs calls this for its capture stream BEFORE the capture begins (graphs.GraphedTrainStep and
    PiecewiseGraphedTrainStep do); a capture nobody prepared gets None and keeps the separate launches."""
    index, handle = key = _sync_key(device, stream)
    buf = _sync_buffers.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        stream = torch.cuda.current_stream(index) if stream is None else stream
        with torch.cuda.stream(stream):
            buf = torch.zeros(int(ffi.lib().ff_xattn_sync_bytes()), dtype=torch.uint8, device=torch.device("cuda", index))
        _sync_buffers[key] = buf
    return buf


def _sync_buffer(device) -> Optional[torch.Tensor]:
    return ensure_sync_buffer(device) if use_sync_exchange else None


def sync_exchange_status(device=None) -> int:
    """1 if an in-launch arrival wait of the fused cross-attention kernels ever timed out on `device` (None: any device) - that launch's
    output was invalid -, else 0.  Blocking: synchronises the streams that own sync buffers."""
    bad = 0
    for (index, handle), buf in list(_sync_buffers.items()):
        if device is None or torch.device(device).index in (None, index):
            rc = int(ffi.lib().ff_xattn_sync_status(buf.data_ptr(), handle))
            if rc < 0:
                ffi.check(rc, "ff_xattn_sync_status")
            bad |= rc
    return bad


def check_sync_exchange(where: str = "", device=None) -> None:
    """Blocking check; raises SyncExchangeTimeout if any in-launch hand-off ever timed out.  Called by the graph-replay steps (after the
    warm-up, after the capture, every `check_every` replays, in close()) and by bench.py after its timed region."""
    if _sync_buffers and sync_exchange_status(device):
        raise SyncExchangeTimeout(
            f"{where or 'check_sync_exchange'}: an in-launch hand-off of the fused cross-attention kernels timed out (ff_xattn_sync_status = 1): "
            "a sample's eight workgroups were not resident together (CU mask, partitioned or shared GPU?). Results since the last clean check are "
            "invalid; set flamingo_mini_amd.functional.use_sync_exchange = False to run the separate launches.")


def poll_sync_exchange(where: str = "") -> None:
    """Non-blocking form for eager training loops (FusedAdamW.step / ShardedAdamW.step call it): looks at the status word that the PREVIOUS
    call copied to pinned memory - raising SyncExchangeTimeout if it was set - and enqueues the next copy behind the work issued so far.
    Never synchronises; does nothing for a stream that is capturing (a captured step is checked by its graph-step object)."""
    for key, buf in list(_sync_buffers.items()):
        index, handle = key
        probe = _sync_probes.get(key)
        if probe is not None:
            host, event = probe
            if not event.query():
                continue                                     # the previous probe is still in flight: look again next time
            if int(host.item()):
                raise SyncExchangeTimeout(f"{where or 'poll_sync_exchange'}: an in-launch hand-off of the fused cross-attention kernels timed out "
                                          "(see functional.check_sync_exchange)")
        stream = torch.cuda.ExternalStream(handle, device=torch.device("cuda", index)) if handle else torch.cuda.default_stream(index)
        with torch.cuda.stream(stream):
            if torch.cuda.is_current_stream_capturing():
                continue
            host = probe[0] if probe is not None else torch.zeros(1, dtype=torch.int32).pin_memory()
            host.copy_(buf.view(torch.int32)[_status_word(): _status_word() + 1], non_blocking=True)
            event = torch.cuda.Event()
            event.record(stream)
        _sync_probes[key] = (host, event)


_AUTO = object()


def _xattn_desc(y, n_media, n_visual, dim_visual, cfg, tt, tt_offset=0, ck=None, cv=None, sync=_AUTO) -> ffi.XattnDesc:
    """sync: _AUTO = the current stream's buffer if the switch is on and the shape can use one (forward calls); a tensor or None = what the
    forward call of this block decided (backward calls pass ctx.sync); the tensor is kept alive on the descriptor."""
    heads, dim_head, ff_mult, act = cfg
    b, L, d = y.shape
    desc = ffi.XattnDesc(ffi.dtype_code(y.dtype), b, L, d, dim_visual, n_media, n_visual, heads, dim_head, ff_mult, ffi.ACTS[act],
                         tt.shape[1], tt_offset)
    if ck is not None:
        desc.cached_k = ffi.Strides(ck.stride(0), ck.stride(2), ck.stride(1))   # (b, h, n, d) tensor -> (sb, sr, sh)
        desc.cached_v = ffi.Strides(cv.stride(0), cv.stride(2), cv.stride(1))
    if sync is _AUTO:
        sync = _sync_buffer(y.device) if (y.is_cuda and y.dtype == torch.bfloat16 and heads == 8 and dim_head == 64 and L <= 32) else None
    desc.sync = None if sync is None else sync.data_ptr()
    desc.sync_tensor = sync
    return desc


class _XattnBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, vf, tt, cfg, n_visual, *params):
        lib = ffi.lib()
        y = y.contiguous()
        vf = vf.contiguous()
        params = tuple(p.contiguous() for p in params)
        desc = _xattn_desc(y, vf.shape[1], n_visual, vf.shape[3], cfg, tt)
        dev = y.device
        saved = _empty_bytes(lib.ff_xattn_saved_bytes(desc), dev)
        scratch = _empty_bytes(lib.ff_xattn_scratch_bytes(desc), dev)
        out = torch.empty_like(y)
        ffi.check(lib.ff_xattn_block_fwd(desc, y.data_ptr(), vf.data_ptr(), tt.data_ptr(), ffi.ptr_array(params), None, None,
                                         out.data_ptr(), saved.data_ptr(), saved.numel(), scratch.data_ptr(), scratch.numel(),
                                         ffi.stream_handle(dev)), "ff_xattn_block_fwd")
        ctx.cfg, ctx.n_visual, ctx.sync = cfg, n_visual, desc.sync_tensor
        ctx.save_for_backward(y, vf, tt, saved, *params)
        ctx.mark_non_differentiable(saved)
        ctx.set_materialize_grads(False)     # no zero-filled "gradient" of the saved-activation buffer (a multi-MB uint8 fill per block)
        return out, saved

    @staticmethod
    def backward(ctx, dout, _dsaved):
        lib = ffi.lib()
        y, vf, tt, saved, *params = ctx.saved_tensors
        desc = _xattn_desc(y, vf.shape[1], ctx.n_visual, vf.shape[3], ctx.cfg, tt, sync=ctx.sync)
        dev = y.device
        dout = torch.zeros_like(y) if dout is None else dout.contiguous()
        flat, grads = _flat_grads(params)
        dy = torch.empty_like(y)
        dvf = torch.empty_like(vf) if ctx.needs_input_grad[1] else None
        scratch = _empty_bytes(lib.ff_xattn_scratch_bytes(desc), dev)
        ffi.check(lib.ff_xattn_block_bwd(desc, y.data_ptr(), vf.data_ptr(), tt.data_ptr(), ffi.ptr_array(params), dout.data_ptr(),
                                         saved.data_ptr(), saved.numel(), ffi.ptr_array(grads), dy.data_ptr(), ffi.ptr(dvf),
                                         scratch.data_ptr(), scratch.numel(), ffi.stream_handle(dev)), "ff_xattn_block_bwd")
        _announce(flat, params)
        return (dy, dvf, None, None, None, *grads)


# ---- keys / values of ALL layers projected up front (ff_kv_project_*), blocks consume them and hand d K / d V back ----
class _KvProjectFn(torch.autograd.Function):
    """(vf (b, N, q, dv), W_0 .. W_{n-1} (kv_dim, dv)) -> KV_0 .. KV_{n-1} (b, N*q, kv_dim), K = [..., :kv_dim/2], V = the rest."""

    @staticmethod
    def forward(ctx, vf, *weights):
        lib = ffi.lib()
        vf = vf.contiguous()
        weights = tuple(w.contiguous() for w in weights)
        b, N, q, dv = vf.shape
        kv_dim = weights[0].shape[0]
        desc = ffi.KvProjDesc(ffi.dtype_code(vf.dtype), len(weights), b * N * q, dv, kv_dim)
        kvs = tuple(torch.empty((b, N * q, kv_dim), dtype=vf.dtype, device=vf.device) for _ in weights)
        ws = _empty_bytes(lib.ff_kv_project_workspace_bytes(desc, 0), vf.device)
        ffi.check(lib.ff_kv_project_fwd(desc, vf.data_ptr(), ffi.ptr_array(weights), ffi.ptr_array(kvs), ws.data_ptr(), ws.numel(),
                                        ffi.stream_handle(vf.device)), "ff_kv_project_fwd")
        ctx.desc = desc
        ctx.save_for_backward(vf, *weights)
        ctx.set_materialize_grads(False)
        return kvs

    @staticmethod
    def backward(ctx, *dkvs):
        lib = ffi.lib()
        vf, *weights = ctx.saved_tensors
        desc = ctx.desc
        b, N, q, _ = vf.shape
        dkvs = [torch.zeros((b, N * q, desc.kv_dim), dtype=vf.dtype, device=vf.device) if g is None else g.contiguous() for g in dkvs]
        flat, grads = _flat_grads(weights)
        dvf = torch.empty_like(vf) if ctx.needs_input_grad[0] else None
        ws = _empty_bytes(lib.ff_kv_project_workspace_bytes(desc, 1 if dvf is not None else 0), vf.device)
        ffi.check(lib.ff_kv_project_bwd(desc, vf.data_ptr(), ffi.ptr_array(weights), ffi.ptr_array(dkvs), ffi.ptr_array(grads), ffi.ptr(dvf),
                                        ws.data_ptr(), ws.numel(), ffi.stream_handle(vf.device)), "ff_kv_project_bwd")
        _announce(flat, weights)
        return (dvf, *grads)


def kv_project(visual_features: torch.Tensor, to_kv_weights: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, ...]:
    """`to_kv` of every cross-attention layer applied to the same visual features (gated_cross_attention.py:84-86) in grouped
    launches; returns one (b, N*q, 2*inner) tensor per layer for xattn_block(..., hoisted_kv=...)."""
    ffi.require_cuda(visual_features, *to_kv_weights)
    _same_dtype(visual_features, to_kv_weights, "kv_project")
    return _KvProjectFn.apply(visual_features, *to_kv_weights)


_KV_PARAM = 5        # position of attn.to_kv.weight in a block's parameter list


class _XattnBlockKvFn(torch.autograd.Function):
    """Block forward / backward with externally projected K / V: consumes `kv` (b, n_kv, 2*inner), returns d kv."""

    @staticmethod
    def _desc(y, kv, n_visual, dim_visual, cfg, tt, sync=_AUTO):
        inner = cfg[0] * cfg[1]
        desc = _xattn_desc(y, kv.shape[1] // n_visual, n_visual, dim_visual, cfg, tt, sync=sync)
        desc.cached_k = desc.cached_v = ffi.Strides(kv.shape[1] * 2 * inner, 2 * inner, cfg[1])   # (batch, row, head) strides
        return desc, inner

    @staticmethod
    def forward(ctx, y, kv, tt, cfg, n_visual, wgrad, *params):
        lib = ffi.lib()
        _wgrad_queue.forget_dead_passes()
        y, kv = y.contiguous(), kv.contiguous()
        params = tuple(p.contiguous() for p in params)
        desc, inner = _XattnBlockKvFn._desc(y, kv, n_visual, params[_KV_PARAM].shape[1], cfg, tt)
        dev = y.device
        saved = _empty_bytes(lib.ff_xattn_saved_bytes(desc), dev)
        scratch = _empty_bytes(lib.ff_xattn_scratch_bytes(desc), dev)
        out = torch.empty_like(y)
        ffi.check(lib.ff_xattn_block_fwd(desc, y.data_ptr(), None, tt.data_ptr(), ffi.ptr_array(params), kv.data_ptr(),
                                         kv.data_ptr() + inner * kv.element_size(), out.data_ptr(), saved.data_ptr(), saved.numel(),
                                         scratch.data_ptr(), scratch.numel(), ffi.stream_handle(dev)), "ff_xattn_block_fwd(hoisted kv)")
        ctx.cfg, ctx.n_visual, ctx.sync = cfg, n_visual, desc.sync_tensor
        defer, group = wgrad if wgrad is not None else (True, None)
        ctx.defer = bool(defer)
        ctx.group = max(1, min(ffi.WGRAD_GROUP_MAX, int(group if group else _wgrad_queue.group)))
        ctx.save_for_backward(y, kv, tt, saved, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = ffi.lib()
        y, kv, tt, saved, *params = ctx.saved_tensors
        desc, inner = _XattnBlockKvFn._desc(y, kv, ctx.n_visual, params[_KV_PARAM].shape[1], ctx.cfg, tt, sync=ctx.sync)
        dev = y.device
        dout = dout.contiguous()
        own = [p for i, p in enumerate(params) if i != _KV_PARAM]            # d to_kv.weight comes from _KvProjectFn
        flat, own_grads = _flat_grads(own)
        grads = own_grads[:_KV_PARAM] + [None] + own_grads[_KV_PARAM:]
        dy, dkv = torch.empty_like(y), torch.empty_like(kv)
        scratch = _empty_bytes(lib.ff_xattn_scratch_bytes(desc), dev)
        aligned = all(t.data_ptr() % 16 == 0 for t in (y, dout, params[2], params[7]))     # the deferred entry point requires it
        if _wgrad_queue.enabled and ctx.defer and aligned and all(p.grad is None for p in own) and not _wgrad_queue.seen_in_this_pass(own):
            # data gradients now; d ffw.3 / d ffw.1 / d to_out / d to_q (and the final reductions of the LayerNorm / gate gradients)
            # later, grouped with the neighbouring layers' (see _WgradQueue)
            stash = _empty_bytes(lib.ff_xattn_wgrad_stash_bytes(desc), dev)
            ffi.check(lib.ff_xattn_block_bwd_kv_data(desc, y.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * kv.element_size(), tt.data_ptr(),
                                                     ffi.ptr_array(params), dout.data_ptr(), saved.data_ptr(), saved.numel(), ffi.ptr_array(grads),
                                                     dy.data_ptr(), dkv.data_ptr(), stash.data_ptr(), stash.numel(), scratch.data_ptr(),
                                                     scratch.numel(), ffi.stream_handle(dev)), "ff_xattn_block_bwd_kv_data")
            key = (dev, y.dtype, tuple(y.shape), kv.shape[1], ctx.n_visual, tuple(ctx.cfg), params[_KV_PARAM].shape[1])
            offs, _ = _flat_offsets(own)
            deferred = tuple(i for i in range(len(params)) if i != _KV_PARAM)      # every gradient of the block is completed by the flush
            own_index = {i: (i if i < _KV_PARAM else i - 1) for i in deferred}
            _wgrad_queue.push(dict(key=key, group=ctx.group, desc=desc, device=dev, dout=dout, saved=saved, stash=stash, params=params, flat=flat, own=own,
                                   grad_ptrs=[None if g is None else g.data_ptr() for g in grads],
                                   wparams=[params[i] for i in deferred],
                                   wslices=[(offs[own_index[i]], params[i].numel()) for i in deferred]))
            return (dy, dkv, None, None, None, None, *grads)
        ffi.check(lib.ff_xattn_block_bwd_kv(desc, y.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * kv.element_size(), tt.data_ptr(),
                                            ffi.ptr_array(params), dout.data_ptr(), saved.data_ptr(), saved.numel(), ffi.ptr_array(grads),
                                            dy.data_ptr(), dkv.data_ptr(), scratch.data_ptr(), scratch.numel(), ffi.stream_handle(dev)),
                  "ff_xattn_block_bwd_kv")
        _announce(flat, own)
        return (dy, dkv, None, None, None, None, *grads)


def _kv_views(saved: torch.Tensor, y: torch.Tensor, n_kv: int, heads: int, dim_head: int):
    """K / V as (b, h, n_kv, dim_head) views of the library's (b, n_kv, 2, h, dim_head) buffer (no copy)."""
    b = y.shape[0]
    n = b * n_kv * 2 * heads * dim_head
    kv = saved[: n * y.element_size()].view(y.dtype).view(b, n_kv, 2, heads, dim_head)
    return kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)


def xattn_block(y: torch.Tensor, visual_features: Optional[torch.Tensor], tt: torch.Tensor, params: Sequence[torch.Tensor], cfg,
                n_visual: int, previous_kv: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, output_kv: bool = False,
                hoisted_kv: Optional[torch.Tensor] = None, wgrad: Optional[Tuple[bool, Optional[int]]] = None):
    """GatedCrossAttentionBlock forward.  cfg = (heads, dim_head, ff_mult, act); tt = text_time int32 (b, L_total).
    hoisted_kv: this layer's output of kv_project (then visual_features is not read).  wgrad = (defer, group): whether this block's weight
    gradients are deferred into grouped launches and how many same-shaped blocks a launch batches (None: deferred, the queue's default).
    Returns (y_out, (k, v) or None)."""
    ffi.require_cuda(y, tt, *params)
    _same_dtype(y, params, "GatedCrossAttentionBlock")
    heads, dim_head = cfg[0], cfg[1]
    if previous_kv is None and hoisted_kv is not None:
        ffi.require_cuda(hoisted_kv)
        out = _XattnBlockKvFn.apply(y, hoisted_kv, tt, tuple(cfg), n_visual, wgrad, *params)
        kv = None
        if output_kv:
            split = hoisted_kv.detach().view(hoisted_kv.shape[0], hoisted_kv.shape[1], 2, heads, dim_head)
            kv = (split[:, :, 0].permute(0, 2, 1, 3), split[:, :, 1].permute(0, 2, 1, 3))
        return out, kv
    if previous_kv is None:
        ffi.require_cuda(visual_features)
        if visual_features.dtype != y.dtype:
            visual_features = visual_features.to(y.dtype)
        out, saved = _XattnBlockFn.apply(y, visual_features, tt, tuple(cfg), n_visual, *params)
        kv = _kv_views(saved, y, visual_features.shape[1] * n_visual, heads, dim_head) if output_kv else None
        return out, kv
    # cached decode (gated_cross_attention.py:88-92,102-104): inference only, K/V reused, last n_token rows of text_time
    lib = ffi.lib()
    k, v = previous_kv
    if k.dtype != y.dtype:
        k, v = k.to(y.dtype), v.to(y.dtype)
    if k.stride(3) != 1 or v.stride(3) != 1:
        k, v = k.contiguous(), v.contiguous()
    y = y.contiguous()
    n_kv = k.shape[2]
    L = y.shape[1]
    desc = _xattn_desc(y, n_kv // n_visual, n_visual, params[5].shape[1], cfg, tt, tt_offset=tt.shape[1] - L, ck=k, cv=v)
    dev = y.device
    saved = _empty_bytes(lib.ff_xattn_saved_bytes(desc), dev)
    scratch = _empty_bytes(lib.ff_xattn_scratch_bytes(desc), dev)
    out = torch.empty_like(y)
    with torch.no_grad():
        ffi.check(lib.ff_xattn_block_fwd(desc, y.data_ptr(), None, tt.data_ptr(), ffi.ptr_array([p.contiguous() for p in params]),
                                         k.data_ptr(), v.data_ptr(), out.data_ptr(), saved.data_ptr(), saved.numel(),
                                         scratch.data_ptr(), scratch.numel(), ffi.stream_handle(dev)), "ff_xattn_block_fwd(cached)")
    return out, ((k, v) if output_kv else None)


# ----------------------------------------------------------------------------------------------------
# QuickGELU of the CLIP tower
# ----------------------------------------------------------------------------------------------------
class _QuickGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        y = torch.empty_like(x)
        ffi.check(ffi.lib().ff_quick_gelu_fwd(ffi.dtype_code(x.dtype), x.numel(), x.data_ptr(), y.data_ptr(), ffi.stream_handle(x.device)),
                  "ff_quick_gelu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        ffi.check(ffi.lib().ff_quick_gelu_bwd(ffi.dtype_code(x.dtype), x.numel(), x.data_ptr(), dy.data_ptr(), dx.data_ptr(),
                                              ffi.stream_handle(x.device)), "ff_quick_gelu_bwd")
        return dx


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """x * sigmoid(1.702 x) in one pass (float32 / bfloat16 on the GPU)."""
    ffi.require_cuda(x)
    return _QuickGeluFn.apply(x)


# ----------------------------------------------------------------------------------------------------
# shifted cross-entropy (modeling_flamingo.py:288-298)
# ----------------------------------------------------------------------------------------------------
class _ShiftedCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        lib = ffi.lib()
        logits = logits.contiguous()
        labels = labels.contiguous().to(torch.int64)
        b, L, V = logits.shape
        rows = torch.empty(b * (L - 1), dtype=torch.float32, device=logits.device)
        lse = torch.empty_like(rows)
        ffi.check(lib.ff_shifted_ce_fwd(ffi.dtype_code(logits.dtype), b, L, V, logits.data_ptr(), labels.data_ptr(), ignore_index,
                                        rows.data_ptr(), lse.data_ptr(), ffi.stream_handle(logits.device)), "ff_shifted_ce_fwd")
        ctx.ignore_index = ignore_index
        ctx.save_for_backward(logits, labels, lse)
        return rows

    @staticmethod
    def backward(ctx, grad_rows):
        lib = ffi.lib()
        logits, labels, lse = ctx.saved_tensors
        b, L, V = logits.shape
        dlogits = torch.empty_like(logits)
        g = grad_rows.contiguous().float()
        ffi.check(lib.ff_shifted_ce_bwd(ffi.dtype_code(logits.dtype), b, L, V, logits.data_ptr(), labels.data_ptr(), ctx.ignore_index,
                                        lse.data_ptr(), g.data_ptr(), dlogits.data_ptr(), ffi.stream_handle(logits.device)), "ff_shifted_ce_bwd")
        return dlogits, None, None


def shifted_cross_entropy(logits: torch.Tensor, labels: torch.Tensor, reduction: str = "mean", ignore_index: int = -100) -> torch.Tensor:
    """F.cross_entropy(logits[..., :-1, :].reshape(-1, V), labels[..., 1:].reshape(-1), reduction=reduction) in two fused passes
    (fp32 math).  logits (b, L, V) float32 / bfloat16 on the GPU, labels (b, L) integer."""
    ffi.require_cuda(logits, labels)
    rows = _ShiftedCEFn.apply(logits, labels, ignore_index)
    if reduction == "none":
        return rows
    if reduction == "sum":
        return rows.sum()
    if reduction == "mean":
        count = (labels[..., 1:] != ignore_index).sum()
        return rows.sum() / count            # an all-ignored batch gives 0 / 0 = NaN, like F.cross_entropy(reduction="mean")
    raise ValueError(f"unknown reduction {reduction}")


# ----------------------------------------------------------------------------------------------------
# primitive wrappers (parity tests / micro-benchmarks): thin, allocation + one C call each
# ----------------------------------------------------------------------------------------------------
def gemm(A, B, *, a_layout=0, b_layout=0, scale=1.0, act=None, act_bwd=None, aux_in=None, residual=None, gate=None,
         want_aux_out=False, split_k=0, tile=0, stages=0):
    lib = ffi.lib()
    ffi.require_cuda(A, B)
    M, K = (A.shape if a_layout == 0 else A.shape[::-1])
    N = B.shape[0] if b_layout == 0 else B.shape[1]
    C_ = torch.empty((M, N), dtype=A.dtype, device=A.device)
    aux_out = torch.empty_like(C_) if want_aux_out else None
    d = ffi.GemmDesc(ffi.dtype_code(A.dtype), M, N, K, a_layout, b_layout, ffi.rowmap(A.stride(0)), ffi.rowmap(B.stride(0)),
                     ffi.rowmap(N), float(scale), ffi.ACTS.get(act, ffi.ACT_NONE), ffi.ACTS.get(act_bwd, ffi.ACT_NONE), split_k, tile, stages)
    ws = _empty_bytes(lib.ff_gemm_workspace_bytes(d), A.device)
    ffi.check(lib.ff_gemm(d, A.data_ptr(), B.data_ptr(), C_.data_ptr(), ffi.ptr(aux_out), ffi.ptr(aux_in), ffi.ptr(residual), ffi.ptr(gate),
                          ws.data_ptr(), ws.numel(), ffi.stream_handle(A.device)), "ff_gemm")
    return (C_, aux_out) if want_aux_out else C_


def layernorm_fwd(x, gamma, beta, add=None, add_rows_per_seg=0, add_div=0, eps=1e-5):
    lib = ffi.lib()
    rows, cols = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    d = ffi.LnDesc(ffi.dtype_code(x.dtype), rows, cols, ffi.rowmap(cols), ffi.rowmap(cols), ffi.rowmap(cols), add_rows_per_seg, add_div, eps, 0)
    ffi.check(lib.ff_layernorm_fwd(d, x.data_ptr(), ffi.ptr(add), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                   rstd.data_ptr(), ffi.stream_handle(x.device)), "ff_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, add=None, add_rows_per_seg=0, add_div=0, dx_residual=None, eps=1e-5):
    lib = ffi.lib()
    rows, cols = x.shape
    dx = torch.empty_like(x)
    dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
    d = ffi.LnDesc(ffi.dtype_code(x.dtype), rows, cols, ffi.rowmap(cols), ffi.rowmap(cols), ffi.rowmap(cols), add_rows_per_seg, add_div, eps, 1)
    ws = _empty_bytes(lib.ff_layernorm_bwd_workspace_bytes(d), x.device)
    ffi.check(lib.ff_layernorm_bwd(d, dy.data_ptr(), x.data_ptr(), ffi.ptr(add), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                   dx.data_ptr(), ffi.ptr(dx_residual), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(),
                                   ffi.stream_handle(x.device)), "ff_layernorm_bwd")
    return dx, dg, db


def rows_reduce(x, rows_per_batch, rows_per_group):
    lib = ffi.lib()
    rows, cols = x.shape
    out = torch.empty((rows_per_batch // rows_per_group, cols), dtype=x.dtype, device=x.device)
    d = ffi.ReduceDesc(ffi.dtype_code(x.dtype), rows, cols, ffi.rowmap(cols), rows_per_batch, rows_per_group)
    ws = _empty_bytes(lib.ff_rows_reduce_workspace_bytes(d), x.device)
    ffi.check(lib.ff_rows_reduce(d, x.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), ffi.stream_handle(x.device)), "ff_rows_reduce")
    return out


def gate_grad(a, b, alpha):
    lib = ffi.lib()
    rows, cols = a.shape
    out = torch.empty_like(alpha)
    ws = _empty_bytes(lib.ff_gate_grad_workspace_bytes(rows, cols), a.device)
    ffi.check(lib.ff_gate_grad(ffi.dtype_code(a.dtype), rows, cols, a.data_ptr(), b.data_ptr(), alpha.data_ptr(), out.data_ptr(),
                               ws.data_ptr(), ws.numel(), ffi.stream_handle(a.device)), "ff_gate_grad")
    return out


def _bnhd_strides(t):  # (b, n, h, d) tensor -> (sb, sr, sh)
    return ffi.Strides(t.stride(0), t.stride(1), t.stride(2))


def _attn_desc(q, k, mode, n_visual, tt):
    b, nq, h, dh = q.shape
    d = ffi.AttnDesc(ffi.dtype_code(q.dtype), b, h, dh, nq, k.shape[1], mode, n_visual, 0 if tt is None else tt.shape[1], 0)
    return d


def attention_fwd(q, k, v, tt=None, n_visual=0):
    """q (b, nq, h, d), k/v (b, nkv, h, d) -> o (b, nq, h, d), lse (b, h, nq).  tt=None: dense."""
    lib = ffi.lib()
    mode = ffi.ATTN_DENSE if tt is None else ffi.ATTN_MEDIA
    d = _attn_desc(q, k, mode, n_visual, tt)
    o = torch.empty_like(q)
    lse = torch.empty((q.shape[0], q.shape[2], q.shape[1]), dtype=torch.float32, device=q.device)
    d.q, d.k, d.v, d.o = _bnhd_strides(q), _bnhd_strides(k), _bnhd_strides(v), _bnhd_strides(o)
    ffi.check(lib.ff_attention_fwd(d, q.data_ptr(), k.data_ptr(), v.data_ptr(), ffi.ptr(tt), o.data_ptr(), lse.data_ptr(),
                                   ffi.stream_handle(q.device)), "ff_attention_fwd")
    return o, lse


def attention_bwd(q, k, v, o, do, lse, tt=None, n_visual=0):
    lib = ffi.lib()
    mode = ffi.ATTN_DENSE if tt is None else ffi.ATTN_MEDIA
    d = _attn_desc(q, k, mode, n_visual, tt)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    d.q, d.k, d.v, d.o = _bnhd_strides(q), _bnhd_strides(k), _bnhd_strides(v), _bnhd_strides(o)
    d.dq, d.dk, d.dv, d.dout = _bnhd_strides(dq), _bnhd_strides(dk), _bnhd_strides(dv), _bnhd_strides(do)
    ws = _empty_bytes(lib.ff_attention_bwd_workspace_bytes(d), q.device)
    ffi.check(lib.ff_attention_bwd(d, q.data_ptr(), k.data_ptr(), v.data_ptr(), ffi.ptr(tt), o.data_ptr(), do.data_ptr(), lse.data_ptr(),
                                   dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ws.data_ptr(), ws.numel(), ffi.stream_handle(q.device)),
              "ff_attention_bwd")
    return dq, dk, dv
