"""Closed-form deterministic tensor generator shared by make_golden.py (build container, reference
present) and the parity tests (GPU box, reference absent), so large weight sets need not be stored."""
from __future__ import annotations

import zlib

import numpy as np


def det(shape, tag: str, scale: float = 1.0, offset: float = 0.0) -> np.ndarray:
    """float32 tensor of uniform(-1,1)*scale + offset values from an integer hash of (tag, element index).

    Pure uint64 arithmetic (wraps mod 2**64) -> bit-identical on every platform; each value is a 24-bit
    fraction, exactly representable in float32."""
    n = int(np.prod(shape))
    seed = np.uint64(zlib.crc32(tag.encode()) * 0x9E3779B97F4A7C15 % (1 << 64))
    with np.errstate(over="ignore"):
        z = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + seed
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)          # [0, 1)
    return ((u * 2.0 - 1.0) * scale + offset).astype(np.float32).reshape(shape)


def resampler_params(dim, depth, heads, dim_head, num_latents, num_time_embeds, ff_mult, tag="rs"):
    """Deterministic PerceiverResampler.state_dict() (reference key names, perceiver_resampler.py:128-141)."""
    inner = heads * dim_head
    p = {
        "latents": det((num_latents, dim), tag + "lat", 1.0),
        "time_pos_emb": det((num_time_embeds, 1, dim), tag + "tpe", 1.0),
        "norm.weight": det((dim,), tag + "nw", 0.2, 1.0),
        "norm.bias": det((dim,), tag + "nb", 0.1),
    }
    for i in range(depth):
        a, f = f"layers.{i}.0.", f"layers.{i}.1."
        for ln in ("norm_media", "norm_latents"):
            p[a + ln + ".weight"] = det((dim,), tag + a + ln + "w", 0.2, 1.0)
            p[a + ln + ".bias"] = det((dim,), tag + a + ln + "b", 0.1)
        p[a + "to_q.weight"] = det((inner, dim), tag + a + "q", (3.0 / dim) ** 0.5)
        p[a + "to_k.weight"] = det((inner, dim), tag + a + "k", (3.0 / dim) ** 0.5)
        p[a + "to_v.weight"] = det((inner, dim), tag + a + "v", (3.0 / dim) ** 0.5)
        p[a + "to_out.weight"] = det((dim, inner), tag + a + "o", (3.0 / inner) ** 0.5)
        p[f + "0.weight"] = det((dim,), tag + f + "lw", 0.2, 1.0)
        p[f + "0.bias"] = det((dim,), tag + f + "lb", 0.1)
        p[f + "1.weight"] = det((ff_mult * dim, dim), tag + f + "1", (3.0 / dim) ** 0.5)
        p[f + "3.weight"] = det((dim, ff_mult * dim), tag + f + "3", (3.0 / (ff_mult * dim)) ** 0.5)
    return p


def xattn_params(dim, dim_visual, heads, dim_head, ff_mult, alpha_attn=0.5, alpha_ffw=-0.4, tag="xa"):
    """Deterministic GatedCrossAttentionBlock.state_dict() (gated_cross_attention.py:36-40,154-158)."""
    inner = heads * dim_head
    return {
        "alpha_attn": np.array([alpha_attn], np.float32),
        "alpha_ffw": np.array([alpha_ffw], np.float32),
        "attn.norm.weight": det((dim,), tag + "nw", 0.2, 1.0),
        "attn.norm.bias": det((dim,), tag + "nb", 0.1),
        "attn.to_q.weight": det((inner, dim), tag + "q", (3.0 / dim) ** 0.5),
        "attn.to_kv.weight": det((2 * inner, dim_visual), tag + "kv", (3.0 / dim_visual) ** 0.5),
        "attn.to_out.weight": det((dim, inner), tag + "o", (3.0 / inner) ** 0.5),
        "ffw.0.weight": det((dim,), tag + "lw", 0.2, 1.0),
        "ffw.0.bias": det((dim,), tag + "lb", 0.1),
        "ffw.1.weight": det((ff_mult * dim, dim), tag + "1", (3.0 / dim) ** 0.5),
        "ffw.3.weight": det((dim, ff_mult * dim), tag + "3", (3.0 / (ff_mult * dim)) ** 0.5),
    }


def bf16_round(a: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32: values a bf16 kernel and a float64 reference both hold exactly."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def det_state(name: str, shape, tag: str = "h64") -> np.ndarray:
    """A whole model's state_dict by NAME (the drop-in keeps the reference's parameter names, so both sides can call this): closed-form,
    bf16-representable values with magnitudes that keep activations O(1).  Used for the full-model fixture whose weights are not stored."""
    shape = tuple(int(s) for s in shape)
    leaf = name.split(".")[-1]
    if "alpha_attn" in name:
        return np.array([0.5 - 0.125 * (zlib.crc32(name.encode()) % 3)], np.float32)
    if "alpha_ffw" in name:
        return np.array([-0.375 + 0.25 * (zlib.crc32(name.encode()) % 3)], np.float32)
    if len(shape) == 0:
        return bf16_round(det((1,), tag + name, 0.5)).reshape(())
    if len(shape) == 1:
        norm_like = any(k in name for k in ("norm", "ln_", "layer_norm", "layrnorm")) or name.endswith("ffw.0.weight") or ".1.0." in name
        if leaf == "weight" and norm_like:
            return bf16_round(det(shape, tag + name, 0.2, 1.0))
        return bf16_round(det(shape, tag + name, 0.1 if leaf != "class_embedding" else 0.5))
    if leaf in ("latents", "time_pos_emb"):
        return bf16_round(det(shape, tag + name, 0.5))
    if "embed" in name or name.endswith(("wte.weight", "wpe.weight")):          # token / position embeddings (the token embedding is also the tied lm_head)
        return bf16_round(det(shape, tag + name, 0.1))
    fan = int(np.prod(shape[1:])) if len(shape) > 2 else max(shape)
    return bf16_round(det(shape, tag + name, (3.0 / fan) ** 0.5))
