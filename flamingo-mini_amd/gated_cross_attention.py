"""Gated cross-attention drop-in (reference: flamingo_mini/gated_cross_attention.py).

`GatedCrossAttentionBlock(y, visual_features, media_locations, previous_kv=None, output_kv=False) -> (y, kv)` and
`ModifiedLMBlock(lm_block, **kw)` with `.condition()` / `.forward()` / `.kv_output` keep the reference's names
and state_dict keys (`alpha_attn`, `alpha_ffw`, `attn.{norm,to_q,to_kv,to_out}`, `ffw.{0,1,3}`).  The whole block
(LayerNorm, projections, masked softmax attention, tanh gates, feed-forward, residuals) is one call into
libflamingo_fusion forward and one backward.
"""
from __future__ import annotations

import inspect
from typing import Optional, Tuple

import torch
from torch import nn

from . import functional as F
from .utils import FeedForward, feedforward_params


class MaskedCrossAttention(nn.Module):
    """Parameter container (reference :15-40); executed inside ff_xattn_block_fwd."""

    def __init__(self, *, dim, dim_visual, dim_head=64, heads=8, n_visual=64):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.n_visual = n_visual
        inner = dim_head * heads
        self.norm = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim_visual, inner * 2, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def forward(self, *args, **kwargs):
        raise RuntimeError("MaskedCrossAttention is executed inside the fused block kernel; call GatedCrossAttentionBlock")


class GatedCrossAttentionBlock(nn.Module):
    def __init__(self, *, dim, dim_visual, dim_head=64, heads=8, ff_mult=4, act='gelu', n_visual=64):
        super().__init__()
        self.attn = MaskedCrossAttention(dim=dim, dim_visual=dim_visual, dim_head=dim_head, heads=heads, n_visual=n_visual)
        self.alpha_attn = nn.Parameter(torch.tensor([0.]))   # gates start closed (reference :155,158)
        self.ffw = FeedForward(dim, mult=ff_mult, act=act)
        self.alpha_ffw = nn.Parameter(torch.tensor([0.]))
        self.cfg = (heads, dim_head, ff_mult, act)
        self.n_visual = n_visual
        # launch structure of this block's backward (not part of the state_dict): weight gradients deferred into launches grouped with the
        # neighbouring layers' (functional._WgradQueue), `wgrad_group` blocks per launch (None: the library default, 12)
        self.defer_wgrad = True
        self.wgrad_group: Optional[int] = None

    def fused_params(self):
        a = self.attn
        return [self.alpha_attn, self.alpha_ffw, a.norm.weight, a.norm.bias, a.to_q.weight, a.to_kv.weight, a.to_out.weight] + \
            feedforward_params(self.ffw)

    def forward(self, y: torch.Tensor, visual_features: Optional[torch.Tensor], media_locations: torch.Tensor,
                previous_kv: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, output_kv: bool = False,
                text_time: Optional[torch.Tensor] = None, hoisted_kv: Optional[torch.Tensor] = None):
        """y (b, L, dim); visual_features (b, N, n_visual, dim_visual); media_locations (b, L_total) 0/1.
        `text_time` (int32 cumsum of media_locations) may be passed to share it between layers; `hoisted_kv` is this layer's
        slice of functional.kv_project (keys / values of all layers projected up front)."""
        if previous_kv is None and hoisted_kv is None:
            assert visual_features is not None and visual_features.ndim == 4
        if text_time is None:
            text_time = F.text_time(media_locations)
        if previous_kv is None:
            assert text_time.shape == y.shape[:2]
        shape_before = y.shape
        params, defer = self.fused_params(), self.defer_wgrad
        cdt = F.autocast_compute_dtype(y)
        if cdt is not None:
            # torch.autocast over fp32 parameters (the reference's --fp16 / --bf16 recipe): the kernels get casts of the parameters in their own
            # dtype and the output goes back in the dtype it came in.  The casts' backward READS the gradients the moment this block's backward
            # returns them, so the weight gradients cannot be deferred into a later grouped launch.
            in_dtype, defer = y.dtype, False
            params = F.autocast_params(params, cdt)
            y = y.to(cdt)
            visual_features = None if visual_features is None else visual_features.to(cdt)
            hoisted_kv = None if hoisted_kv is None else hoisted_kv.to(cdt)
        extra = {} if hoisted_kv is None else {"hoisted_kv": hoisted_kv, "wgrad": (defer, self.wgrad_group)}
        out, kv = F.xattn_block(y, visual_features, text_time, params, self.cfg, self.n_visual,
                                previous_kv=previous_kv, output_kv=bool(output_kv), **extra)
        assert out.shape == shape_before
        if cdt is not None and out.dtype != in_dtype:
            out = out.to(in_dtype)
        return out, kv


class ModifiedLMBlock(nn.Module):
    """The LM-layer interleave hook: gated cross-attention block, then the wrapped (frozen) LM block.

    Visual input arrives through `condition()` because the HF layer loop only passes hidden states.  Unlike the
    reference (:231-252) the forward accepts whatever positional / keyword arguments the HF version in use gives its
    blocks (transformers >= 5 calls GPT-2 blocks positionally) and forwards them untouched.
    """

    def __init__(self, lm_block, **kwargs):
        super().__init__()
        self.xattn_block = GatedCrossAttentionBlock(**kwargs)
        self.lm_block = lm_block
        self.visual_features = None
        self.media_locations = None
        self.xattn_layer_past = None
        self.text_time = None
        self.hoisted_kv = None
        self.kv_output = None
        self.autograd_cut = None     # graphs.AutogradCuts.cut when this layer starts a backward segment (FlamingoBaseModel.install_autograd_cuts)

    def condition(self, visual_features: torch.Tensor, media_locations: torch.Tensor, xattn_layer_past=None,
                  text_time: Optional[torch.Tensor] = None, hoisted_kv: Optional[torch.Tensor] = None) -> None:
        self.visual_features = visual_features
        self.media_locations = media_locations
        self.xattn_layer_past = xattn_layer_past
        self.text_time = text_time
        self.hoisted_kv = hoisted_kv

    def _use_cache_flag(self, args, kwargs) -> bool:
        if "use_cache" in kwargs:
            return bool(kwargs["use_cache"])
        if args:  # an HF version that passes it positionally: resolve through the wrapped block's own signature
            try:
                bound = inspect.signature(self.lm_block.forward).bind_partial(None, *args, **kwargs)
                return bool(bound.arguments.get("use_cache", False))
            except TypeError:
                pass
        return False

    def forward(self, hidden_states, *args, **kwargs):
        use_cache = self._use_cache_flag(args, kwargs)
        if self.autograd_cut is not None:
            hidden_states = self.autograd_cut(hidden_states)
        hidden_states, kv = self.xattn_block(
            y=hidden_states,
            visual_features=self.visual_features,
            media_locations=self.media_locations,
            previous_kv=self.xattn_layer_past,
            output_kv=use_cache,
            text_time=self.text_time,
            hoisted_kv=self.hoisted_kv,
        )
        self.kv_output = kv
        return self.lm_block(hidden_states, *args, **kwargs)
