"""TESTS ONLY: a host stand-in for libflamingo_fusion.so's training entry points, so that the PRODUCT's Python layer - the autograd
Functions of flamingo_mini_amd.functional, the deferred weight-gradient queue, the gradient buckets and the data-parallel reducers that
hang off them - can run on CPU tensors (2-rank gloo) instead of only its oracle-backed imitation (tests/oracle_backend.py).

Every function below has the C signature of include/flamingo_fusion.h, takes raw HOST pointers (CPU tensors' data_ptr), computes with
the numpy oracle in float64 and writes float32 results back through the pointers.  The opaque `saved` / `stash` buffers only serve as
keys for the Python-side caches.  `install()` points ffi.lib() at it and lets CPU tensors through ffi.require_cuda; nothing in the
product package knows about this file (the product has no CPU path: without install() CPU tensors raise)."""
import ctypes as C

import numpy as np

from oracle import flamingo_oracle as O
from oracle_backend import XA_KEYS, rs_keys

_ACT = {0: "gelu", 1: "sqrelu", 2: "relu"}


def _view(ptr, shape, dtype=np.float32):
    ptr = ptr.value if hasattr(ptr, "value") else ptr
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.frombuffer((C.c_byte * n).from_address(int(ptr)), dtype=dtype).reshape(shape)


def _ptrs(arr, n):
    return [arr[i] for i in range(n)]        # c_void_p array elements come back as int / None


class HostLib:
    """Float32 only (dtype code 0)."""

    def __init__(self):
        self.cache = {}          # saved pointer -> whatever the backward needs
        self.pending = {}        # stash pointer -> parameter gradients of a block whose weight gradients were deferred
        self.calls = []          # names of the entry points in call order (the tests look at the grouping)
        self.err = b""

    # ---- bookkeeping -------------------------------------------------------------------------------------------------
    def ff_version(self):
        return 2

    def ff_last_error(self):
        return self.err

    def _f32(self, d):
        assert d.dtype == 0, "the host stand-in computes float32 problems only"

    # ---- text_time ----------------------------------------------------------------------------------------------------
    def ff_text_time(self, b, n, ml, elem_bytes, out, stream):
        src = _view(ml, (b, n), {8: np.int64, 4: np.int32, 1: np.uint8}[elem_bytes])
        _view(out, (b, n), np.int32)[...] = np.cumsum(src.astype(np.int64), axis=1).astype(np.int32)
        return 0

    # ---- resampler ----------------------------------------------------------------------------------------------------
    def _rs_shapes(self, d):
        inner, ffi_ = d.heads * d.dim_head, d.ff_mult * d.dim
        D = d.dim
        layer = [(D,), (D,), (D,), (D,), (inner, D), (inner, D), (inner, D), (D, inner), (D,), (D,), (ffi_, D), (D, ffi_)]
        return [(d.num_latents, D), (d.num_time_embeds, 1, D), (D,), (D,)] + layer * d.depth

    def ff_resampler_saved_bytes(self, d):
        return 64

    def ff_resampler_scratch_bytes(self, d):
        return 64

    def ff_resampler_fwd(self, d, x, params, out, saved, saved_n, scratch, scratch_n, stream):
        self._f32(d)
        self.calls.append("ff_resampler_fwd")
        shapes = self._rs_shapes(d)
        p = {k: _view(q, s).astype(np.float64) for k, q, s in zip(rs_keys(d.depth), _ptrs(params, len(shapes)), shapes)}
        xf = _view(x, (d.batch, d.n_frames, d.n_tokens, d.dim)).astype(np.float64)
        y, cache = O.resampler_fwd(xf, p, heads=d.heads, dim_head=d.dim_head, act=_ACT[d.act])
        _view(out, y.shape)[...] = y
        self.cache[int(saved)] = (cache, p)
        return 0

    def ff_resampler_bwd(self, d, x, params, dout, saved, saved_n, grads, dx, scratch, scratch_n, stream):
        self.calls.append("ff_resampler_bwd")
        cache, p = self.cache.pop(int(saved))
        dy = _view(dout, (d.batch, d.num_latents, d.dim)).astype(np.float64)
        dxf, g = O.resampler_bwd(dy, cache, p, heads=d.heads, dim_head=d.dim_head, act=_ACT[d.act])
        shapes = self._rs_shapes(d)
        for k, q, s in zip(rs_keys(d.depth), _ptrs(grads, len(shapes)), shapes):
            _view(q, s)[...] = np.asarray(g[k]).reshape(s)
        if dx:
            _view(dx, dxf.shape)[...] = dxf
        return 0

    # ---- resampler, one layer per call (ff_resampler_layer_* + prologue / epilogue) -------------------------------------
    _LAYER_KEYS = ("0.norm_media.weight", "0.norm_media.bias", "0.norm_latents.weight", "0.norm_latents.bias", "0.to_q.weight", "0.to_k.weight",
                   "0.to_v.weight", "0.to_out.weight", "1.0.weight", "1.0.bias", "1.1.weight", "1.3.weight")

    def ff_resampler_prologue_saved_bytes(self, d):
        return 64

    def ff_resampler_layer_saved_bytes(self, d):
        return 64

    def ff_resampler_layer_scratch_bytes(self, d):
        return 64

    def ff_resampler_epilogue_saved_bytes(self, d):
        return 64

    def _layer_params(self, d, params):
        shapes = self._rs_shapes(d)[4:16]
        return {"layers.0." + k: _view(q, s).astype(np.float64) for k, q, s in zip(self._LAYER_KEYS, _ptrs(params, 12), shapes)}

    def _xf(self, d, x, tpe):
        xf = _view(x, (d.batch, d.n_frames, d.n_tokens, d.dim)).astype(np.float64)
        t = _view(tpe, (d.num_time_embeds, 1, d.dim)).astype(np.float64)
        return (xf + t[:d.n_frames]).reshape(d.batch, d.n_frames * d.n_tokens, d.dim)            # perceiver_resampler.py:166,172

    def ff_resampler_prologue_fwd(self, d, x, tpe, pro, pro_n, stream):
        self._f32(d)
        self.calls.append("ff_resampler_prologue_fwd")
        return 0

    def ff_resampler_layer_fwd(self, d, x, tpe, pro, pro_n, x_in, is_latents, params, x_out, saved, saved_n, scratch, scratch_n, stream):
        self._f32(d)
        self.calls.append("ff_resampler_layer_fwd")
        p = self._layer_params(d, params)
        feats = self._xf(d, x, tpe)
        if is_latents:
            lat = _view(x_in, (d.num_latents, d.dim)).astype(np.float64)
            xin = np.broadcast_to(lat, (d.batch,) + lat.shape).copy()                             # :179
        else:
            xin = _view(x_in, (d.batch, d.num_latents, d.dim)).astype(np.float64)
        a_out, a_c = O.perceiver_attention_fwd(feats, xin, p, "layers.0.0.", d.heads, d.dim_head)
        xm = xin + a_out                                                                           # :182
        f_out, f_c = O.feedforward_fwd(xm, p, "layers.0.1.", _ACT[d.act])
        _view(x_out, xm.shape)[...] = xm + f_out                                                   # :183
        self.cache[int(saved)] = (a_c, f_c, p)
        return 0

    def ff_resampler_layer_bwd(self, d, x, tpe, pro, pro_n, x_in, is_latents, params, dx_out, saved, saved_n, grads, dx_in, dx_f, accumulate,
                               scratch, scratch_n, stream):
        self.calls.append("ff_resampler_layer_bwd")
        a_c, f_c, p = self.cache.pop(int(saved))
        g = {}
        dx = _view(dx_out, (d.batch, d.num_latents, d.dim)).astype(np.float64)
        dx = dx + O.feedforward_bwd(dx, f_c, p, "layers.0.1.", _ACT[d.act], g)
        dfeat, dlat = O.perceiver_attention_bwd(dx, a_c, p, "layers.0.0.", d.heads, d.dim_head, g)
        _view(dx_in, dx.shape)[...] = dx + dlat
        out = _view(dx_f, (d.batch, d.n_frames * d.n_tokens, d.dim))
        out[...] = (out.astype(np.float64) if accumulate else 0.0) + dfeat
        shapes = self._rs_shapes(d)[4:16]
        for k, q, s in zip(self._LAYER_KEYS, _ptrs(grads, 12), shapes):
            _view(q, s)[...] = np.asarray(g["layers.0." + k]).reshape(s)
        return 0

    def ff_resampler_prologue_bwd(self, d, dx0, dx_f, d_latents, d_tpe, scratch, scratch_n, stream):
        self.calls.append("ff_resampler_prologue_bwd")
        _view(d_latents, (d.num_latents, d.dim))[...] = _view(dx0, (d.batch, d.num_latents, d.dim)).astype(np.float64).sum(axis=0)
        dxf = _view(dx_f, (d.batch, d.n_frames, d.n_tokens, d.dim)).astype(np.float64)
        gt = np.zeros((d.num_time_embeds, 1, d.dim))
        gt[:d.n_frames] = dxf.sum(axis=(0, 2))[:, None, :]
        _view(d_tpe, gt.shape)[...] = gt
        return 0

    def ff_resampler_epilogue_fwd(self, d, x_last, gamma, beta, out, epi, epi_n, stream):
        self._f32(d)
        self.calls.append("ff_resampler_epilogue_fwd")
        x = _view(x_last, (d.batch, d.num_latents, d.dim)).astype(np.float64)
        g, b = _view(gamma, (d.dim,)).astype(np.float64), _view(beta, (d.dim,)).astype(np.float64)
        y, c = O.layernorm_fwd(x, g, b)
        _view(out, y.shape)[...] = y
        self.cache[int(epi)] = (c, g)
        return 0

    def ff_resampler_epilogue_bwd(self, d, dout, x_last, gamma, epi, epi_n, dx_last, dgamma, dbeta, scratch, scratch_n, stream):
        self.calls.append("ff_resampler_epilogue_bwd")
        c, g = self.cache.pop(int(epi))
        dy = _view(dout, (d.batch, d.num_latents, d.dim)).astype(np.float64)
        dx, dg, db = O.layernorm_bwd(dy, c, g)
        _view(dx_last, dx.shape)[...] = dx
        _view(dgamma, (d.dim,))[...] = dg
        _view(dbeta, (d.dim,))[...] = db
        return 0

    # ---- K / V projection of all layers -------------------------------------------------------------------------------
    def ff_kv_project_workspace_bytes(self, d, with_dvf):
        return 64

    def ff_kv_project_fwd(self, d, vf, weights, kv_out, ws, ws_n, stream):
        self._f32(d)
        self.calls.append(f"ff_kv_project_fwd[{d.n_layers}]")
        rows = _view(vf, (d.rows, d.dim_visual)).astype(np.float64)
        for w, o in zip(_ptrs(weights, d.n_layers), _ptrs(kv_out, d.n_layers)):
            _view(o, (d.rows, d.kv_dim))[...] = rows @ _view(w, (d.kv_dim, d.dim_visual)).astype(np.float64).T
        return 0

    def ff_kv_project_bwd(self, d, vf, weights, dkv, dweights, dvf, ws, ws_n, stream):
        self.calls.append(f"ff_kv_project_bwd[{d.n_layers}]")
        rows = _view(vf, (d.rows, d.dim_visual)).astype(np.float64)
        acc = np.zeros_like(rows)
        for w, g, dw in zip(_ptrs(weights, d.n_layers), _ptrs(dkv, d.n_layers), _ptrs(dweights, d.n_layers)):
            g64 = _view(g, (d.rows, d.kv_dim)).astype(np.float64)
            _view(dw, (d.kv_dim, d.dim_visual))[...] = g64.T @ rows
            acc += g64 @ _view(w, (d.kv_dim, d.dim_visual)).astype(np.float64)
        if dvf:
            _view(dvf, acc.shape)[...] = acc
        return 0

    # ---- gated cross-attention block on externally projected K / V ------------------------------------------------------
    def _xa_shapes(self, d):
        inner, ffi_ = d.heads * d.dim_head, d.ff_mult * d.dim
        return [(1,), (1,), (d.dim,), (d.dim,), (inner, d.dim), (2 * inner, d.dim_visual), (d.dim, inner), (d.dim,), (d.dim,), (ffi_, d.dim),
                (d.dim, ffi_)]

    def ff_xattn_saved_bytes(self, d):
        # the library's layout starts with K / V as (b, n_kv, 2, heads, dim_head) when the block projects them itself: functional._kv_views
        # hands views of that region to the caller (output_kv=True), so the stand-in keeps the same contract
        own_kv = d.cached_k.sr == 0
        return 64 + (d.batch * d.n_media * d.n_visual * 2 * d.heads * d.dim_head * 4 if own_kv else 0)

    def ff_xattn_kv_offset(self, d):
        return 0

    def ff_xattn_scratch_bytes(self, d):
        return 64

    def ff_xattn_wgrad_stash_bytes(self, d):
        return 64

    def ff_xattn_wgrad_workspace_bytes(self, d):
        return 64

    @staticmethod
    def _strided(ptr, d, st):
        """(b, heads, n_kv, dim_head) float32 view of externally held K or V: element strides (batch, row, head) from the descriptor."""
        n_kv = d.n_media * d.n_visual
        span = (d.batch - 1) * st.sb + (n_kv - 1) * st.sr + (d.heads - 1) * st.sh + d.dim_head
        base = _view(ptr, (span,))
        return np.lib.stride_tricks.as_strided(base, shape=(d.batch, d.heads, n_kv, d.dim_head),
                                               strides=(st.sb * 4, st.sh * 4, st.sr * 4, 4), writeable=False)

    def ff_xattn_block_fwd(self, d, y, vf, tt, params, ck, cv, out, saved, saved_n, scratch, scratch_n, stream):
        """Three layouts, as in the library: vf given (the block projects K / V itself and keeps them at the start of `saved`), K / V
        projected outside during training (ck = the layer's (b, n_kv, 2 inner) tensor: full sequence), cached decode (ck / cv with arbitrary
        (b, h, n, d) strides, the LAST n_tokens rows of text_time)."""
        self._f32(d)
        inner = d.heads * d.dim_head
        n_kv = d.n_media * d.n_visual
        shapes = self._xa_shapes(d)
        p = {k: (None if q is None else _view(q, s).astype(np.float64)) for k, q, s in zip(XA_KEYS, _ptrs(params, len(shapes)), shapes)}
        ttv = _view(tt, (d.batch, d.tt_stride), np.int32).astype(np.int64)
        ml_full = np.diff(ttv, axis=1, prepend=0)
        yv = _view(y, (d.batch, d.n_tokens, d.dim)).astype(np.float64)
        kw = dict(heads=d.heads, dim_head=d.dim_head, act=_ACT[d.act], n_visual=d.n_visual)
        if vf is not None:                                              # per-layer projection (hoist_kv = False)
            self.calls.append("ff_xattn_block_fwd[own kv]")
            assert d.tt_offset == 0 and d.tt_stride == d.n_tokens
            vfv = _view(vf, (d.batch, d.n_media, d.n_visual, d.dim_visual)).astype(np.float64)
            o, (k, v), cache = O.gated_xattn_block_fwd(yv, vfv, ml_full, p, **kw)
            kvbuf = _view(saved, (d.batch, n_kv, 2, d.heads, d.dim_head))
            kvbuf[:, :, 0] = k.transpose(0, 2, 1, 3)
            kvbuf[:, :, 1] = v.transpose(0, 2, 1, 3)
            self.cache[int(saved)] = (cache, p, n_kv)
        elif d.tt_offset == 0 and d.tt_stride == d.n_tokens and int(cv) - int(ck) == inner * 4 and d.cached_k.sr == 2 * inner:
            self.calls.append("ff_xattn_block_fwd")                    # training layout: K / V of this layer projected outside the block
            p["attn.to_kv.weight"] = np.eye(2 * inner)                  # the oracle projects K / V itself: identity on the projected tensor
            kv = _view(ck, (d.batch, d.n_media, d.n_visual, 2 * inner)).astype(np.float64)
            o, _, cache = O.gated_xattn_block_fwd(yv, kv, ml_full, p, **kw)
            self.cache[int(saved)] = (cache, p, n_kv)
        else:                                                           # cached decode: inference only
            self.calls.append("ff_xattn_block_fwd[cached]")
            k = self._strided(ck, d, d.cached_k).astype(np.float64)
            v = self._strided(cv, d, d.cached_v).astype(np.float64)
            assert d.tt_offset + d.n_tokens == d.tt_stride, "cached decode reads the last n_tokens entries of text_time"
            o, _, _ = O.gated_xattn_block_fwd(yv, None, ml_full, p, previous_kv=(k, v), **kw)
        _view(out, o.shape)[...] = o
        return 0

    def ff_xattn_block_bwd(self, d, y, vf, tt, params, dy_out, saved, saved_n, grads, dy, dvf, scratch, scratch_n, stream):
        self.calls.append("ff_xattn_block_bwd")
        cache, p, n_kv = self.cache.pop(int(saved))
        g_out = _view(dy_out, (d.batch, d.n_tokens, d.dim)).astype(np.float64)
        dyv, dvfv, g = O.gated_xattn_block_bwd(g_out, cache, p, heads=d.heads, dim_head=d.dim_head, act=_ACT[d.act])
        _view(dy, dyv.shape)[...] = dyv
        if dvf:
            _view(dvf, dvfv.shape)[...] = dvfv
        shapes = self._xa_shapes(d)
        for k, q, s in zip(XA_KEYS, _ptrs(grads, len(shapes)), shapes):
            _view(q, s)[...] = np.asarray(g[k]).reshape(s)
        return 0

    def _xa_backward(self, d, dy_out, saved):
        cache, p, n_kv = self.cache.pop(int(saved))
        g_out = _view(dy_out, (d.batch, d.n_tokens, d.dim)).astype(np.float64)
        dy, dkv4, g = O.gated_xattn_block_bwd(g_out, cache, p, heads=d.heads, dim_head=d.dim_head, act=_ACT[d.act])
        return dy, dkv4.reshape(d.batch, n_kv, -1), g

    def _write_grads(self, d, grads, g):
        shapes = self._xa_shapes(d)
        for i, (k, q, s) in enumerate(zip(XA_KEYS, _ptrs(grads, len(shapes)), shapes)):
            if i != 5:                                                 # d to_kv.weight belongs to ff_kv_project_bwd
                assert q, f"null gradient pointer for {k}"
                _view(q, s)[...] = np.asarray(g[k]).reshape(s)

    def ff_xattn_block_bwd_kv(self, d, y, k, v, tt, params, dy_out, saved, saved_n, grads, dy, dkv, scratch, scratch_n, stream):
        self.calls.append("ff_xattn_block_bwd_kv")
        dyv, dkvv, g = self._xa_backward(d, dy_out, saved)
        _view(dy, dyv.shape)[...] = dyv
        _view(dkv, dkvv.shape)[...] = dkvv
        self._write_grads(d, grads, g)
        return 0

    def ff_xattn_block_bwd_kv_data(self, d, y, k, v, tt, params, dy_out, saved, saved_n, grads, dy, dkv, stash, stash_n, scratch,
                                   scratch_n, stream):
        """Data gradients now; EVERY parameter gradient only with ff_xattn_wgrad_grouped (the strictest reading of the contract)."""
        self.calls.append("ff_xattn_block_bwd_kv_data")
        for t in (y, dy_out, dy):
            assert int(t) % 16 == 0, "the deferred entry point requires 16-byte aligned rows"
        dyv, dkvv, g = self._xa_backward(d, dy_out, saved)
        _view(dy, dyv.shape)[...] = dyv
        _view(dkv, dkvv.shape)[...] = dkvv
        shapes = self._xa_shapes(d)
        for i, (q, s) in enumerate(zip(_ptrs(grads, len(shapes)), shapes)):
            if i != 5:
                _view(q, s)[...] = np.nan                               # whoever reads a gradient before the grouped call sees it
        self.pending[int(stash)] = g
        return 0

    def ff_xattn_wgrad_grouped(self, d, n, dy_out, saved, saved_n, stash, stash_n, params, grads, ws, ws_n, stream):
        self.calls.append(f"ff_xattn_wgrad_grouped[{n}]")
        assert 1 <= n <= 12
        for i in range(n):
            g = self.pending.pop(int(stash[i]))
            sub = (C.c_void_p * 11)(*[grads[i * 11 + j] for j in range(11)])
            self._write_grads(d, sub, g)
        return 0


_saved = {}


def install() -> HostLib:
    from flamingo_mini_amd import ffi
    host = HostLib()
    if not _saved:
        _saved.update(lib=ffi.lib, require_cuda=ffi.require_cuda, stream_handle=ffi.stream_handle)
    ffi.lib = lambda: host
    ffi.require_cuda = lambda *tensors: None
    ffi.stream_handle = lambda device: None
    return host


def uninstall() -> None:
    from flamingo_mini_amd import ffi
    for k, v in _saved.items():
        setattr(ffi, k, v)
    _saved.clear()
