"""CPU oracle for the Flamingo fusion path (PerceiverResampler + GatedCrossAttentionBlock).

TEST INFRASTRUCTURE ONLY.  This file is a plain-numpy restatement of the reference's
algorithm, forward AND backward, used as the checker for the HIP kernels.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
the product package (`flamingo-mini_amd/`) never does and fails loudly without its HIP
library.

Pinning: the reference ships no tests, so this oracle is pinned
against outputs of the reference itself, generated in the build container by
`tests/golden/make_golden.py` (imports /root/reference, runs its modules + autograd in
fp64/fp32) and committed as `tests/golden/*.npz`.  `tests/test_oracle_golden.py` checks
every function here against those vectors.

Every function cites the reference lines it restates (paths relative to the reference
repository root).  Parameters are passed as dicts keyed by the reference's own
state_dict names so checkpoints/fixtures interchange.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np

try:  # exact erf for nn.GELU() (flamingo_mini/utils.py:37 -> torch erf GELU)
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover - scipy is present in the image
    _erf = np.vectorize(math.erf)

LN_EPS = 1e-5  # torch.nn.LayerNorm default, used at perceiver_resampler.py:24-25,141; gated_cross_attention.py:36; utils.py:46

Params = Dict[str, np.ndarray]


# ----------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------
def layernorm_fwd(x: np.ndarray, g: np.ndarray, b: np.ndarray):
    """nn.LayerNorm over the last axis (biased variance, eps inside the sqrt)."""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + x.dtype.type(LN_EPS))
    xhat = xc * rstd
    return xhat * g + b, (xhat, rstd)


def layernorm_bwd(dy: np.ndarray, cache, g: np.ndarray):
    xhat, rstd = cache
    dyh = dy * g
    m1 = dyh.mean(axis=-1, keepdims=True)
    m2 = (dyh * xhat).mean(axis=-1, keepdims=True)
    dx = rstd * (dyh - m1 - xhat * m2)
    red = tuple(range(dy.ndim - 1))
    return dx, (dy * xhat).sum(axis=red), dy.sum(axis=red)


def act_fwd(h: np.ndarray, act: str) -> np.ndarray:
    """utils.py:26-30: gelu = nn.GELU() (exact erf), sqrelu = relu(x)**2, relu."""
    if act == "gelu":
        return (0.5 * h * (1.0 + _erf(h / math.sqrt(2.0)))).astype(h.dtype)
    if act == "sqrelu":
        r = np.maximum(h, 0)
        return r * r
    if act == "relu":
        return np.maximum(h, 0)
    raise AssertionError(f"act. can only be one of gelu/sqrelu/relu, got {act}")


def act_bwd(da: np.ndarray, h: np.ndarray, act: str) -> np.ndarray:
    if act == "gelu":
        cdf = 0.5 * (1.0 + _erf(h / math.sqrt(2.0)))
        pdf = np.exp(-0.5 * h * h) / math.sqrt(2.0 * math.pi)
        return (da * (cdf + h * pdf)).astype(h.dtype)
    if act == "sqrelu":
        return da * 2.0 * np.maximum(h, 0)
    if act == "relu":
        return da * (h > 0)
    raise AssertionError(act)


def linear_fwd(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """nn.Linear(bias=False): weight layout (out, in)."""
    return x @ w.T


def linear_bwd(dy: np.ndarray, x: np.ndarray, w: np.ndarray):
    dx = dy @ w
    dw = dy.reshape(-1, dy.shape[-1]).T @ x.reshape(-1, x.shape[-1])
    return dx, dw


def _split_heads(t: np.ndarray, h: int) -> np.ndarray:  # 'b n (h d) -> b h n d'
    b, n, hd = t.shape
    return t.reshape(b, n, h, hd // h).transpose(0, 2, 1, 3)


def _merge_heads(t: np.ndarray) -> np.ndarray:  # 'b h n d -> b n (h d)'
    b, h, n, d = t.shape
    return t.transpose(0, 2, 1, 3).reshape(b, n, h * d)


def _softmax_lastdim(s: np.ndarray) -> np.ndarray:
    s = s - s.max(axis=-1, keepdims=True)
    e = np.exp(s)
    return e / e.sum(axis=-1, keepdims=True)


# ----------------------------------------------------------------------------------------
# FeedForward  (flamingo_mini/utils.py:22-50)
# ----------------------------------------------------------------------------------------
def feedforward_fwd(x: np.ndarray, p: Params, prefix: str, act: str):
    """Sequential(LayerNorm, Linear(dim,4dim,no bias), act, Linear(4dim,dim,no bias)); keys {prefix}0/1/3."""
    xn, ln_c = layernorm_fwd(x, p[prefix + "0.weight"], p[prefix + "0.bias"])
    h = linear_fwd(xn, p[prefix + "1.weight"])
    a = act_fwd(h, act)
    y = linear_fwd(a, p[prefix + "3.weight"])
    return y, (ln_c, xn, h, a)


def feedforward_bwd(dy: np.ndarray, cache, p: Params, prefix: str, act: str, grads: Params):
    ln_c, xn, h, a = cache
    da, grads[prefix + "3.weight"] = linear_bwd(dy, a, p[prefix + "3.weight"])
    dh = act_bwd(da, h, act)
    dxn, grads[prefix + "1.weight"] = linear_bwd(dh, xn, p[prefix + "1.weight"])
    dx, grads[prefix + "0.weight"], grads[prefix + "0.bias"] = layernorm_bwd(dxn, ln_c, p[prefix + "0.weight"])
    return dx


# ----------------------------------------------------------------------------------------
# PerceiverAttentionLayer  (flamingo_mini/perceiver_resampler.py:9-96)
# ----------------------------------------------------------------------------------------
def perceiver_attention_fwd(features: np.ndarray, latents: np.ndarray, p: Params, prefix: str,
                            heads: int, dim_head: int):
    assert features.ndim == 3 and latents.ndim == 3                      # :42-43
    assert features.shape[0] == latents.shape[0]                          # :44
    assert features.shape[2] == latents.shape[2]                          # :45
    scale = features.dtype.type(dim_head ** -0.5)                         # :18
    xm, lnm_c = layernorm_fwd(features, p[prefix + "norm_media.weight"], p[prefix + "norm_media.bias"])      # :52
    ln, lnl_c = layernorm_fwd(latents, p[prefix + "norm_latents.weight"], p[prefix + "norm_latents.bias"])  # :53
    q = _split_heads(linear_fwd(ln, p[prefix + "to_q.weight"]), heads)   # :57-58
    kv_in = np.concatenate((xm, ln), axis=-2)                             # :65  (normalised latents attend to themselves)
    k = _split_heads(linear_fwd(kv_in, p[prefix + "to_k.weight"]), heads)  # :69,75
    v = _split_heads(linear_fwd(kv_in, p[prefix + "to_v.weight"]), heads)  # :70,75
    q = q * scale                                                         # :79  (scaled BEFORE the dot product)
    sim = q @ k.transpose(0, 1, 3, 2)                                     # :85
    alphas = _softmax_lastdim(sim)                                        # :88-89 (amax shift is a no-op on the result)
    o = _merge_heads(alphas @ v)                                          # :92,95
    out = linear_fwd(o, p[prefix + "to_out.weight"])                      # :96
    return out, (lnm_c, lnl_c, ln, kv_in, q, k, v, alphas, o, features.shape[1])


def perceiver_attention_bwd(dout: np.ndarray, cache, p: Params, prefix: str, heads: int, dim_head: int,
                            grads: Params):
    lnm_c, lnl_c, ln, kv_in, q, k, v, alphas, o, n_feat = cache
    scale = dout.dtype.type(dim_head ** -0.5)
    do, grads[prefix + "to_out.weight"] = linear_bwd(dout, o, p[prefix + "to_out.weight"])
    do = _split_heads(do, heads)
    dalphas = do @ v.transpose(0, 1, 3, 2)
    dv = alphas.transpose(0, 1, 3, 2) @ do
    dsim = alphas * (dalphas - (dalphas * alphas).sum(axis=-1, keepdims=True))
    dq = (dsim @ k) * scale
    dk = dsim.transpose(0, 1, 3, 2) @ q          # q is the scaled query
    dq, dk, dv = _merge_heads(dq), _merge_heads(dk), _merge_heads(dv)
    dln, grads[prefix + "to_q.weight"] = linear_bwd(dq, ln, p[prefix + "to_q.weight"])
    dkv_k, grads[prefix + "to_k.weight"] = linear_bwd(dk, kv_in, p[prefix + "to_k.weight"])
    dkv_v, grads[prefix + "to_v.weight"] = linear_bwd(dv, kv_in, p[prefix + "to_v.weight"])
    dkv = dkv_k + dkv_v
    dxm = dkv[:, :n_feat]
    dln = dln + dkv[:, n_feat:]
    dfeat, grads[prefix + "norm_media.weight"], grads[prefix + "norm_media.bias"] = \
        layernorm_bwd(dxm, lnm_c, p[prefix + "norm_media.weight"])
    dlat, grads[prefix + "norm_latents.weight"], grads[prefix + "norm_latents.bias"] = \
        layernorm_bwd(dln, lnl_c, p[prefix + "norm_latents.weight"])
    return dfeat, dlat


# ----------------------------------------------------------------------------------------
# PerceiverResampler  (flamingo_mini/perceiver_resampler.py:99-188)
# ----------------------------------------------------------------------------------------
def resampler_depth(p: Params) -> int:
    d = 0
    while f"layers.{d}.0.to_q.weight" in p:
        d += 1
    return d


def resampler_fwd(x_f: np.ndarray, p: Params, heads: int = 8, dim_head: int = 64, act: str = "gelu"):
    """x_f: (b, v, d) or (b, T, v, d) -> (b, num_latents, d).  Params keyed as PerceiverResampler.state_dict()."""
    if x_f.ndim == 3:                                                     # :150-152
        x_f = x_f[:, None]
    assert x_f.ndim == 4                                                  # :154
    b, T, n, d = x_f.shape
    assert d == p["latents"].shape[1]                                     # :161
    tpe = p["time_pos_emb"]
    if T > tpe.shape[0]:
        raise ValueError("more frames than resampler_num_time_embeds (broadcast error at perceiver_resampler.py:166)")
    x_f = x_f + tpe[:T]                                                   # :166 (always added, also for T=1)
    x_f = x_f.reshape(b, T * n, d)                                        # :172
    x = np.broadcast_to(p["latents"], (b,) + p["latents"].shape).copy()  # :179
    caches = []
    for i in range(resampler_depth(p)):                                   # :181-183
        a_out, a_c = perceiver_attention_fwd(x_f, x, p, f"layers.{i}.0.", heads, dim_head)
        x = x + a_out
        f_out, f_c = feedforward_fwd(x, p, f"layers.{i}.1.", act)
        x = x + f_out
        caches.append((a_c, f_c))
    assert x.shape == (b, p["latents"].shape[0], d)                      # :185
    y, ln_c = layernorm_fwd(x, p["norm.weight"], p["norm.bias"])         # :187
    return y, (caches, ln_c, (b, T, n, d))


def resampler_bwd(dy: np.ndarray, cache, p: Params, heads: int = 8, dim_head: int = 64, act: str = "gelu"):
    """Returns (d x_f with the input's 4-D shape, grads keyed like the state_dict)."""
    caches, ln_c, (b, T, n, d) = cache
    grads: Params = {}
    dx, grads["norm.weight"], grads["norm.bias"] = layernorm_bwd(dy, ln_c, p["norm.weight"])
    dxf = np.zeros((b, T * n, d), dtype=dy.dtype)
    for i in reversed(range(len(caches))):
        a_c, f_c = caches[i]
        dx = dx + feedforward_bwd(dx, f_c, p, f"layers.{i}.1.", act, grads)
        dfeat, dlat = perceiver_attention_bwd(dx, a_c, p, f"layers.{i}.0.", heads, dim_head, grads)
        dx = dx + dlat
        dxf = dxf + dfeat
    grads["latents"] = dx.sum(axis=0)
    dxf = dxf.reshape(b, T, n, d)
    gt = np.zeros_like(p["time_pos_emb"])
    gt[:T] = dxf.sum(axis=(0, 2))[:, None, :]
    grads["time_pos_emb"] = gt
    return dxf, grads


# ----------------------------------------------------------------------------------------
# MaskedCrossAttention + GatedCrossAttentionBlock  (flamingo_mini/gated_cross_attention.py:15-184)
# ----------------------------------------------------------------------------------------
def text_time_of(media_locations: np.ndarray) -> np.ndarray:
    """gated_cross_attention.py:97 — cumulative count of media tags up to and including each token."""
    return np.cumsum(media_locations.astype(np.int64), axis=-1)


def attention_masks(text_time: np.ndarray, n_media: int, n_visual: int):
    """(:106-121) equality mask (token sees ONLY the latents of image #text_time) and the no-media row mask."""
    media_time = np.repeat(np.arange(n_media) + 1, n_visual)              # :106,111
    allow = text_time[:, None, :, None] == media_time[None, None, None, :]
    no_media = (text_time == 0)[:, None, :, None]                         # :119-120
    return allow, no_media


def masked_cross_attention_fwd(y: np.ndarray, media_locations: np.ndarray, visual_features: Optional[np.ndarray],
                               p: Params, prefix: str, heads: int, dim_head: int, n_visual: int,
                               previous_kv: Optional[Tuple[np.ndarray, np.ndarray]] = None):
    """Returns (conditioned_tokens, (k, v), cache).  Keys {prefix}norm/to_q/to_kv/to_out."""
    n_token = y.shape[1]
    scale = y.dtype.type(dim_head ** -0.5)
    yn, ln_c = layernorm_fwd(y, p[prefix + "norm.weight"], p[prefix + "norm.bias"])     # :74
    q = _split_heads(linear_fwd(yn, p[prefix + "to_q.weight"]) * scale, heads)          # :77-78,87
    if previous_kv is None:
        vf = visual_features.reshape(visual_features.shape[0], -1, visual_features.shape[-1])  # :84
        kv = linear_fwd(vf, p[prefix + "to_kv.weight"])                                  # :86
        inner = kv.shape[-1] // 2
        k = _split_heads(kv[..., :inner], heads)                                         # chunk(2): K first, V second
        v = _split_heads(kv[..., inner:], heads)
    else:
        vf = None
        k, v = previous_kv                                                               # :90
    n_media = k.shape[2] // n_visual                                                     # :69 / :91
    sim = q @ k.transpose(0, 1, 3, 2)                                                    # :95
    tt = text_time_of(media_locations)                                                   # :97
    if previous_kv is not None:
        tt = tt[:, -n_token:]                                                            # :103
        assert tt.shape == y.shape[:2]                                                   # :104
    allow, no_media = attention_masks(tt, n_media, n_visual)
    sim = np.where(allow, sim, -np.finfo(sim.dtype).max)                                 # :112
    alphas = _softmax_lastdim(sim)                                                       # :114-115 (all-masked row -> uniform)
    alphas = np.where(no_media, sim.dtype.type(0), alphas)                               # :121
    o = _merge_heads(alphas @ v)                                                         # :123-124
    out = linear_fwd(o, p[prefix + "to_out.weight"])                                     # :126
    return out, (k, v), (ln_c, yn, vf, q, k, v, alphas, allow, no_media, o)


def masked_cross_attention_bwd(dout: np.ndarray, cache, p: Params, prefix: str, heads: int, dim_head: int,
                               grads: Params):
    """Returns (dy, d visual_features flattened (b, N*q, dv))."""
    ln_c, yn, vf, q, k, v, alphas, allow, no_media, o = cache
    scale = dout.dtype.type(dim_head ** -0.5)
    do, grads[prefix + "to_out.weight"] = linear_bwd(dout, o, p[prefix + "to_out.weight"])
    do = _split_heads(do, heads)
    dalphas = do @ v.transpose(0, 1, 3, 2)
    dalphas = np.where(no_media, dout.dtype.type(0), dalphas)          # masked_fill(rows, 0) blocks the gradient
    dv = alphas.transpose(0, 1, 3, 2) @ do
    dsim = alphas * (dalphas - (dalphas * alphas).sum(axis=-1, keepdims=True))
    dsim = np.where(allow, dsim, dout.dtype.type(0))                   # masked_fill(~mask, const) blocks the gradient
    dq = _merge_heads(dsim @ k) * scale
    dk = _merge_heads(dsim.transpose(0, 1, 3, 2) @ q)
    dv = _merge_heads(dv)
    dyn, grads[prefix + "to_q.weight"] = linear_bwd(dq, yn, p[prefix + "to_q.weight"])
    dkv = np.concatenate((dk, dv), axis=-1)
    dvf, grads[prefix + "to_kv.weight"] = linear_bwd(dkv, vf, p[prefix + "to_kv.weight"])
    dy, grads[prefix + "norm.weight"], grads[prefix + "norm.bias"] = layernorm_bwd(dyn, ln_c, p[prefix + "norm.weight"])
    return dy, dvf


def gated_xattn_block_fwd(y: np.ndarray, visual_features: Optional[np.ndarray], media_locations: np.ndarray,
                          p: Params, heads: int = 8, dim_head: int = 64, act: str = "gelu", n_visual: int = 64,
                          previous_kv=None):
    """GatedCrossAttentionBlock.forward (:160-184).  Params keyed as the block's state_dict
    (alpha_attn, alpha_ffw, attn.*, ffw.*).  Returns (y_out, (k, v), cache)."""
    if previous_kv is None:
        assert visual_features.ndim == 4                                                 # :175
    attn_out, kv, a_c = masked_cross_attention_fwd(y, media_locations, visual_features, p, "attn.", heads,
                                                   dim_head, n_visual, previous_kv)      # :179
    ta = np.tanh(p["alpha_attn"]).astype(y.dtype)
    y1 = y + ta * attn_out                                                               # :180
    ffw_out, f_c = feedforward_fwd(y1, p, "ffw.", act)
    tf = np.tanh(p["alpha_ffw"]).astype(y.dtype)
    y2 = y1 + tf * ffw_out                                                               # :182
    vshape = None if visual_features is None else visual_features.shape
    return y2, kv, (a_c, f_c, attn_out, ffw_out, ta, tf, vshape)


def gated_xattn_block_bwd(dy2: np.ndarray, cache, p: Params, heads: int = 8, dim_head: int = 64,
                          act: str = "gelu"):
    """Returns (dy, d visual_features (b,N,q,dv), grads keyed like the block's state_dict)."""
    a_c, f_c, attn_out, ffw_out, ta, tf, vshape = cache
    grads: Params = {}
    grads["alpha_ffw"] = np.array([(dy2 * ffw_out).sum() * (1.0 - tf[0] ** 2)], dtype=dy2.dtype)
    dy1 = dy2 + feedforward_bwd(dy2 * tf, f_c, p, "ffw.", act, grads)
    grads["alpha_attn"] = np.array([(dy1 * attn_out).sum() * (1.0 - ta[0] ** 2)], dtype=dy2.dtype)
    dy_attn, dvf = masked_cross_attention_bwd(dy1 * ta, a_c, p, "attn.", heads, dim_head, grads)
    return dy1 + dy_attn, dvf.reshape(vshape), grads


# ----------------------------------------------------------------------------------------
# FLOP model — used by bench.py and DESIGN.md
# ----------------------------------------------------------------------------------------
def resampler_flops_fwd(b, T, v, dv, depth=6, q=64, h=8, dh=64, ff_mult=4):
    inner = h * dh
    f = T * v + q
    per = 2 * q * dv * inner + 4 * f * dv * inner + 4 * h * q * f * dh + 2 * q * inner * dv + 4 * q * dv * ff_mult * dv
    return b * depth * per


def xattn_block_flops_fwd(b, L, d, dv, N=1, q=64, h=8, dh=64, ff_mult=4):
    inner = h * dh
    per = 2 * L * d * inner + 4 * N * q * dv * inner + 4 * h * L * N * q * dh + 2 * L * inner * d + 4 * L * d * ff_mult * d
    return b * per


# ----------------------------------------------------------------------------------------
# AdamW (the reference trains with HF Trainer `--optim adamw_torch`, training/train.sh:10-13): torch.optim.AdamW's rule
# ----------------------------------------------------------------------------------------
def adamw_step(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
    """One decoupled-weight-decay Adam step on numpy arrays; returns new (p, m, v).  `step` is 1-based."""
    p = p * (1.0 - lr * weight_decay)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    p = p - (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + eps)
    return p, m, v
