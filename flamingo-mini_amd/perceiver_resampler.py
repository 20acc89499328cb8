"""PerceiverResampler drop-in (reference: flamingo_mini/perceiver_resampler.py:99-188).

Same constructor, same parameter names (`latents`, `time_pos_emb`, `layers.{i}.0.{norm_media,norm_latents,to_q,
to_k,to_v,to_out}`, `layers.{i}.1.{0,1,3}`, `norm`), same call `resampler(x_f) -> (b, num_latents, dim)`.
The modules below only OWN parameters; forward + backward of the whole stack is one call each into
libflamingo_fusion (ff_resampler_fwd / ff_resampler_bwd).
"""
from __future__ import annotations

import torch
from torch import nn

from . import functional as F
from .utils import FeedForward, feedforward_params


class PerceiverAttentionLayer(nn.Module):
    """Parameter container of one latent<-(media ++ latents) attention layer (reference :9-30)."""

    def __init__(self, *, dim, dim_head=64, heads=8):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        inner = dim_head * heads
        self.norm_media = nn.LayerNorm(dim)
        self.norm_latents = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(dim, inner, bias=False)
        self.to_v = nn.Linear(dim, inner, bias=False)
        self.to_out = nn.Linear(inner, dim, bias=False)

    def fused_params(self):
        return [self.norm_media.weight, self.norm_media.bias, self.norm_latents.weight, self.norm_latents.bias,
                self.to_q.weight, self.to_k.weight, self.to_v.weight, self.to_out.weight]

    def forward(self, features, latents):
        raise RuntimeError("PerceiverAttentionLayer is executed inside the fused resampler kernels; call PerceiverResampler")


class PerceiverResampler(nn.Module):
    def __init__(self, *, dim, depth, dim_head=64, heads=8, num_latents=64, num_time_embeds=4, ff_mult=4, act='gelu'):
        super().__init__()
        self.dim = dim
        self.n_queries = num_latents
        self.depth, self.heads, self.dim_head = depth, heads, dim_head
        self.num_time_embeds, self.ff_mult, self.act = num_time_embeds, ff_mult, act
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.time_pos_emb = nn.Parameter(torch.randn(num_time_embeds, 1, dim))
        self.layers = nn.ModuleList(
            nn.ModuleList([PerceiverAttentionLayer(dim=dim, dim_head=dim_head, heads=heads), FeedForward(dim=dim, mult=ff_mult, act=act)])
            for _ in range(depth))
        self.norm = nn.LayerNorm(dim)
        # One library call per layer (ff_resampler_layer_* + prologue / epilogue) instead of the stack-level call: every layer's gradients are
        # final - one data-parallel bucket - when its own backward is done.  Data-parallel reducers switch it on for THEIR model; a single GPU
        # keeps the stack-level call (weight gradients of four layers per launch).  `autograd_cut`: graphs.AutogradCuts.cut between the layers.
        self.layerwise = False
        self.autograd_cut = None

    def fused_params(self):
        """Flat parameter list in the order of include/flamingo_fusion.h (ff_resampler_fwd)."""
        ps = [self.latents, self.time_pos_emb, self.norm.weight, self.norm.bias]
        for attn, ffw in self.layers:
            ps += attn.fused_params() + feedforward_params(ffw)
        return ps

    def forward(self, x_f: torch.Tensor) -> torch.Tensor:
        """x_f: (b, v, d) or (b, T, v, d) CLIP features -> (b, num_latents, d)."""
        if x_f.ndim == 3:
            x_f = x_f.unsqueeze(1)
        assert x_f.ndim == 4
        assert x_f.shape[3] == self.dim
        if x_f.shape[1] > self.num_time_embeds:
            raise RuntimeError(f"{x_f.shape[1]} frames but only {self.num_time_embeds} time embeddings")
        cfg = (self.depth, self.heads, self.dim_head, self.n_queries, self.num_time_embeds, self.ff_mult, self.act)
        params = self.fused_params()
        cdt = F.autocast_compute_dtype(x_f)
        if cdt is not None:                      # torch.autocast over fp32 parameters: see functional.autocast_compute_dtype
            params = F.autocast_params(params, cdt)
        if x_f.dtype != params[0].dtype:
            x_f = x_f.to(params[0].dtype)
        if self.layerwise:
            out = F.resampler_layerwise(x_f, params, cfg, cut=self.autograd_cut)
        else:
            out = F.resampler(x_f, params, cfg)
        assert out.shape == (x_f.shape[0], self.n_queries, self.dim)
        return out
