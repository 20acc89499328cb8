"""ctypes binding of libflamingo_fusion.so (C ABI declared in include/flamingo_fusion.h).

The library is the product path: if it is missing or fails to load, every call raises — there is no
PyTorch/CPU fallback.  Tensors cross the boundary as raw device pointers + the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# FLAMINGO_FUSION_LIB: "debug" = the development build next to the shipped library (build.py --debug: -DFF_DEBUG, the only build whose
# kernels' A/B switches read the environment), or a path; default = the shipped library
_which = os.environ.get("FLAMINGO_FUSION_LIB", "")
LIB_PATH = os.path.join(PKG_DIR, "libflamingo_fusion_debug.so") if _which == "debug" else (_which or os.path.join(PKG_DIR, "libflamingo_fusion.so"))
ABI_VERSION = 4

FF_OK = 0
DTYPE_F32, DTYPE_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_SQRELU, ACT_RELU = -1, 0, 1, 2
ACTS = {"gelu": ACT_GELU, "sqrelu": ACT_SQRELU, "relu": ACT_RELU}
ATTN_DENSE, ATTN_MEDIA = 0, 1
RESAMPLER_GLOBAL_PARAMS, RESAMPLER_LAYER_PARAMS, XATTN_PARAMS = 4, 12, 11
WGRAD_GROUP_MAX = 12           # capacity of one ff_xattn_wgrad_grouped call (FF_WGRAD_GROUP_MAX); functional._WgradQueue.group = how many it batches


class FusionLibraryError(RuntimeError):
    pass


class RowMap(C.Structure):
    _fields_ = [("ld", C.c_longlong), ("seg_stride", C.c_longlong), ("rows_per_seg", C.c_int), ("reserved", C.c_int)]


def rowmap(ld: int, seg_stride: int = 0, rows_per_seg: int = 0) -> RowMap:
    return RowMap(ld, seg_stride, rows_per_seg, 0)


class GemmDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("a_layout", C.c_int), ("b_layout", C.c_int),
                ("a_map", RowMap), ("b_map", RowMap), ("c_map", RowMap), ("scale", C.c_float), ("act", C.c_int), ("act_bwd", C.c_int),
                ("split_k", C.c_int), ("tile", C.c_int), ("stages", C.c_int)]


class GemmProfileRecord(C.Structure):
    _fields_ = [("dtype", C.c_int), ("tile", C.c_int), ("a_layout", C.c_int), ("b_layout", C.c_int), ("M", C.c_int), ("N", C.c_int),
                ("K", C.c_int), ("nz", C.c_int), ("split_k", C.c_int), ("ms", C.c_float)]


class LnDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("x_map", RowMap), ("y_map", RowMap), ("dx_map", RowMap),
                ("add_rows_per_seg", C.c_int), ("add_div", C.c_int), ("eps", C.c_float), ("stats_given", C.c_int)]


class ReduceDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("rows", C.c_int), ("cols", C.c_int), ("x_map", RowMap), ("rows_per_batch", C.c_int),
                ("rows_per_group", C.c_int)]


class Strides(C.Structure):
    _fields_ = [("sb", C.c_longlong), ("sr", C.c_longlong), ("sh", C.c_longlong)]


class AttnDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("batch", C.c_int), ("heads", C.c_int), ("dim_head", C.c_int), ("n_q", C.c_int), ("n_kv", C.c_int),
                ("mode", C.c_int), ("n_visual", C.c_int), ("tt_stride", C.c_int), ("tt_offset", C.c_int),
                ("q", Strides), ("k", Strides), ("v", Strides), ("o", Strides),
                ("dq", Strides), ("dk", Strides), ("dv", Strides), ("dout", Strides)]


class ResamplerDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("batch", C.c_int), ("n_frames", C.c_int), ("n_tokens", C.c_int), ("dim", C.c_int),
                ("depth", C.c_int), ("heads", C.c_int), ("dim_head", C.c_int), ("num_latents", C.c_int), ("num_time_embeds", C.c_int),
                ("ff_mult", C.c_int), ("act", C.c_int)]


class AdamWDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("n_tensors", C.c_int), ("step", C.c_int), ("lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float), ("grad_scale", C.c_float), ("step_dev", C.c_void_p)]


class KvProjDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dtype", "n_layers", "rows", "dim_visual", "kv_dim")]


class XattnDesc(C.Structure):
    _fields_ = [("dtype", C.c_int), ("batch", C.c_int), ("n_tokens", C.c_int), ("dim", C.c_int), ("dim_visual", C.c_int),
                ("n_media", C.c_int), ("n_visual", C.c_int), ("heads", C.c_int), ("dim_head", C.c_int), ("ff_mult", C.c_int),
                ("act", C.c_int), ("tt_stride", C.c_int), ("tt_offset", C.c_int), ("cached_k", Strides), ("cached_v", Strides),
                ("sync", C.c_void_p)]


_P, _SZ, _I = C.c_void_p, C.c_size_t, C.c_int
_SIGNATURES = {
    # name: (restype, argtypes) — mirrors include/flamingo_fusion.h one to one
    "ff_version": (_I, []),
    "ff_arch": (C.c_char_p, []),
    "ff_last_error": (C.c_char_p, []),
    "ff_gemm_workspace_bytes": (_SZ, [C.POINTER(GemmDesc)]),
    "ff_gemm": (_I, [C.POINTER(GemmDesc), _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "ff_gemm_profile_enable": (_I, [_I]),
    "ff_gemm_profile_read": (_I, [C.POINTER(GemmProfileRecord), _I]),
    "ff_gemm_plan": (_I, [C.POINTER(GemmDesc), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "ff_layernorm_fwd": (_I, [C.POINTER(LnDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "ff_layernorm_bwd_workspace_bytes": (_SZ, [C.POINTER(LnDesc)]),
    "ff_layernorm_bwd": (_I, [C.POINTER(LnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "ff_rows_reduce_workspace_bytes": (_SZ, [C.POINTER(ReduceDesc)]),
    "ff_rows_reduce": (_I, [C.POINTER(ReduceDesc), _P, _P, _P, _SZ, _P]),
    "ff_gate_grad_workspace_bytes": (_SZ, [_I, _I]),
    "ff_gate_grad": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "ff_text_time": (_I, [_I, _I, _P, _I, _P, _P]),
    "ff_attention_fwd": (_I, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P, _P]),
    "ff_attention_bwd_workspace_bytes": (_SZ, [C.POINTER(AttnDesc)]),
    "ff_attention_bwd": (_I, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "ff_resampler_saved_bytes": (_SZ, [C.POINTER(ResamplerDesc)]),
    "ff_resampler_scratch_bytes": (_SZ, [C.POINTER(ResamplerDesc)]),
    "ff_resampler_fwd": (_I, [C.POINTER(ResamplerDesc), _P, _P, _P, _P, _SZ, _P, _SZ, _P]),
    "ff_resampler_bwd": (_I, [C.POINTER(ResamplerDesc), _P, _P, _P, _P, _SZ, _P, _P, _P, _SZ, _P]),
    "ff_resampler_prologue_saved_bytes": (_SZ, [C.POINTER(ResamplerDesc)]),
    "ff_resampler_layer_saved_bytes": (_SZ, [C.POINTER(ResamplerDesc)]),
    "ff_resampler_layer_scratch_bytes": (_SZ, [C.POINTER(ResamplerDesc)]),
    "ff_resampler_epilogue_saved_bytes": (_SZ, [C.POINTER(ResamplerDesc)]),
    "ff_resampler_prologue_fwd": (_I, [C.POINTER(ResamplerDesc), _P, _P, _P, _SZ, _P]),
    "ff_resampler_layer_fwd": (_I, [C.POINTER(ResamplerDesc), _P, _P, _P, _SZ, _P, _I, _P, _P, _P, _SZ, _P, _SZ, _P]),
    "ff_resampler_epilogue_fwd": (_I, [C.POINTER(ResamplerDesc), _P, _P, _P, _P, _P, _SZ, _P]),
    "ff_resampler_epilogue_bwd": (_I, [C.POINTER(ResamplerDesc), _P, _P, _P, _P, _SZ, _P, _P, _P, _P, _SZ, _P]),
    "ff_resampler_layer_bwd": (_I, [C.POINTER(ResamplerDesc), _P, _P, _P, _SZ, _P, _I, _P, _P, _P, _SZ, _P, _P, _P, _I, _P, _SZ, _P]),
    "ff_resampler_prologue_bwd": (_I, [C.POINTER(ResamplerDesc), _P, _P, _P, _P, _P, _SZ, _P]),
    "ff_xattn_sync_bytes": (_SZ, []),
    "ff_xattn_sync_status": (_I, [_P, _P]),
    "ff_xattn_saved_bytes": (_SZ, [C.POINTER(XattnDesc)]),
    "ff_xattn_scratch_bytes": (_SZ, [C.POINTER(XattnDesc)]),
    "ff_xattn_kv_offset": (_SZ, [C.POINTER(XattnDesc)]),
    "ff_xattn_block_fwd": (_I, [C.POINTER(XattnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _SZ, _P]),
    "ff_quick_gelu_fwd": (_I, [_I, C.c_longlong, _P, _P, _P]),
    "ff_quick_gelu_bwd": (_I, [_I, C.c_longlong, _P, _P, _P, _P]),
    "ff_shifted_ce_fwd": (_I, [_I, _I, _I, _I, _P, _P, C.c_longlong, _P, _P, _P]),
    "ff_shifted_ce_bwd": (_I, [_I, _I, _I, _I, _P, _P, C.c_longlong, _P, _P, _P, _P]),
    "ff_adamw_step": (_I, [C.POINTER(AdamWDesc), _P, _P, _P, _P, _P, _P]),
    "ff_adamw_step_mixed": (_I, [C.POINTER(AdamWDesc), _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "ff_xattn_block_bwd_kv": (_I, [C.POINTER(XattnDesc), _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _P, _P, _P, _SZ, _P]),
    "ff_kv_project_workspace_bytes": (_SZ, [C.POINTER(KvProjDesc), _I]),
    "ff_kv_project_fwd": (_I, [C.POINTER(KvProjDesc), _P, _P, _P, _P, _SZ, _P]),
    "ff_kv_project_bwd": (_I, [C.POINTER(KvProjDesc), _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "ff_xattn_wgrad_stash_bytes": (_SZ, [C.POINTER(XattnDesc)]),
    "ff_xattn_wgrad_workspace_bytes": (_SZ, [C.POINTER(XattnDesc)]),
    "ff_xattn_block_bwd_kv_data": (_I, [C.POINTER(XattnDesc), _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _P, _P, _P, _SZ, _P, _SZ, _P]),
    "ff_xattn_wgrad_grouped": (_I, [C.POINTER(XattnDesc), _I, _P, _P, _SZ, _P, _SZ, _P, _P, _P, _SZ, _P]),
    "ff_xattn_block_bwd": (_I, [C.POINTER(XattnDesc), _P, _P, _P, _P, _P, _P, _SZ, _P, _P, _P, _P, _SZ, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the library once.  Raises FusionLibraryError (never falls back) if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FusionLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m flamingo_mini_amd.build` (needs hipcc, gfx950). "
            "There is no CPU/PyTorch fallback for the fusion path.")
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise FusionLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise FusionLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if handle.ff_version() != ABI_VERSION:
        raise FusionLibraryError(f"ABI mismatch: library {handle.ff_version()} vs binding {ABI_VERSION}; rebuild")
    _lib = handle
    return handle


_TRACE = os.environ.get("FF_TRACE", "0") == "1"      # debugging aid: name every library call on stderr and synchronise after it


def check(rc: int, what: str) -> None:
    if rc != FF_OK:
        msg = lib().ff_last_error().decode(errors="replace")
        raise FusionLibraryError(f"{what} failed (code {rc}): {msg}")
    if _TRACE:
        import sys
        print(f"[ff] {what} enqueued", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        print(f"[ff] {what} done", file=sys.stderr, flush=True)


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return DTYPE_F32
    if dt == torch.bfloat16:
        return DTYPE_BF16
    raise FusionLibraryError(f"the fusion path computes in float32 or bfloat16, got {dt}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def ptr_array(ts: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def stream_handle(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise FusionLibraryError(
                "the Flamingo fusion path runs only on an MI355X (HIP) device; got a CPU tensor. "
                "There is deliberately no CPU fallback in the product path.")
