"""Stock-PyTorch CPU restatement of the Flamingo fusion path (PerceiverResampler + GatedCrossAttentionBlock), backward by torch autograd.

TEST / BASELINE INFRASTRUCTURE ONLY - the same standing as oracle/flamingo_oracle.py: imported by `tests/` and by `bench.py`'s
`cpu_baseline` leg, never by the product package (which has no CPU path and fails loudly without its HIP library).

Why it exists next to the numpy oracle: SURVEY.md 8(d4) asks for the CPU figure that is timed beside the HIP path to be the
all-core torch CPU backend running the same ops in the same order as the reference (ATen `mm` / `bmm` / `softmax` / `layer_norm`,
autograd backward), not a single-threaded-in-places numpy port.  The reference itself cannot travel to the GPU box, so this is
a restatement, written against the reference lines cited below (paths relative to the reference repository), with parameters in
dicts keyed by the reference's state_dict names.  Pinning: tests/test_oracle_golden.py holds it to the same reference-generated
golden vectors (tests/golden/*.npz) as the numpy oracle, forward and gradients, in float64.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as TF

Params = Dict[str, torch.Tensor]
LN_EPS = 1e-5


def _act(h: torch.Tensor, act: str) -> torch.Tensor:
    """utils.py:26-30: gelu = nn.GELU() (exact erf), sqrelu = relu(x)**2, relu."""
    if act == "gelu":
        return TF.gelu(h)
    if act == "sqrelu":
        return torch.relu(h) ** 2
    if act == "relu":
        return torch.relu(h)
    raise AssertionError(f"act. can only be one of gelu/sqrelu/relu, got {act}")


def _ln(x, w, b):
    return TF.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


def _ffw(x, p: Params, prefix: str, act: str):
    """utils.py:45-50: LayerNorm -> Linear(dim, 4 dim, no bias) -> act -> Linear(4 dim, dim, no bias)."""
    h = _ln(x, p[prefix + "0.weight"], p[prefix + "0.bias"]) @ p[prefix + "1.weight"].t()
    return _act(h, act) @ p[prefix + "3.weight"].t()


def _heads(t, h):      # 'b n (h d) -> b h n d'
    b, n, _ = t.shape
    return t.reshape(b, n, h, -1).permute(0, 2, 1, 3)


def resampler(x_f: torch.Tensor, p: Params, heads: int = 8, dim_head: int = 64, act: str = "gelu") -> torch.Tensor:
    """perceiver_resampler.py:143-188 (PerceiverAttentionLayer :32-96): x_f (b, [T,] v, d) -> (b, q, d)."""
    if x_f.ndim == 3:
        x_f = x_f[:, None]                                                  # :150-152
    b, T, v, d = x_f.shape
    x_f = x_f + p["time_pos_emb"][:T]                                       # :166 (always added, also for T = 1)
    x_f = x_f.reshape(b, T * v, d)                                          # :172 frames flattened into the key axis
    x = p["latents"].unsqueeze(0).expand(b, -1, -1)                         # :179
    depth = 1 + max(int(k.split(".")[1]) for k in p if k.startswith("layers."))
    scale = dim_head ** -0.5
    for i in range(depth):
        a = f"layers.{i}.0."
        feats = _ln(x_f, p[a + "norm_media.weight"], p[a + "norm_media.bias"])          # :52
        lat = _ln(x, p[a + "norm_latents.weight"], p[a + "norm_latents.bias"])          # :53
        q = _heads(lat @ p[a + "to_q.weight"].t(), heads) * scale                       # :57-58, :79
        kv_in = torch.cat([feats, lat], dim=1)                                          # :65 latents attend to themselves too
        k = _heads(kv_in @ p[a + "to_k.weight"].t(), heads)                             # :69
        vv = _heads(kv_in @ p[a + "to_v.weight"].t(), heads)                            # :70
        sim = q @ k.transpose(-1, -2)                                                   # :85
        sim = sim - sim.amax(dim=-1, keepdim=True).detach()                             # :88
        o = sim.softmax(dim=-1) @ vv                                                    # :89-92
        o = o.permute(0, 2, 1, 3).reshape(b, -1, heads * dim_head)                      # :95
        x = x + o @ p[a + "to_out.weight"].t()                                          # :96, :182
        x = x + _ffw(x, p, f"layers.{i}.1.", act)                                       # :183
    return _ln(x, p["norm.weight"], p["norm.bias"])                                     # :187


def gated_xattn_block(y: torch.Tensor, vf: Optional[torch.Tensor], media_locations: torch.Tensor, p: Params, heads: int = 8, dim_head: int = 64,
                      act: str = "gelu", n_visual: int = 64, previous_kv: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """gated_cross_attention.py:160-184 (MaskedCrossAttention :42-131).  y (b, L, d), vf (b, N, q, dv), media_locations (b, L_total) 0/1.
    Returns (y_out, (k, v))."""
    b, L, _ = y.shape
    scale = dim_head ** -0.5
    q = _heads(_ln(y, p["attn.norm.weight"], p["attn.norm.bias"]) @ p["attn.to_q.weight"].t(), heads) * scale      # :74-78
    if previous_kv is None:
        flat = vf.reshape(b, -1, vf.shape[-1])                                                                       # :84
        k, v = (flat @ p["attn.to_kv.weight"].t()).chunk(2, dim=-1)                                                  # :86
        k, v = _heads(k, heads), _heads(v, heads)                                                                    # :87
    else:
        k, v = previous_kv                                                                                           # :90-92
    n_media = k.shape[2] // n_visual
    sim = q @ k.transpose(-1, -2)                                                                                    # :95
    text_time = media_locations.to(torch.int64).cumsum(dim=-1)[:, -L:]                                               # :97, :102-104
    media_time = torch.arange(1, n_media + 1, device=y.device).repeat_interleave(n_visual)                           # :106
    mask = text_time[:, None, :, None] == media_time[None, None, None, :]                                            # :111 equality, not >=
    sim = sim.masked_fill(~mask, -torch.finfo(sim.dtype).max)                                                        # :112
    sim = sim - sim.amax(dim=-1, keepdim=True).detach()                                                              # :114
    attn = sim.softmax(dim=-1)                                                                                       # :115
    attn = attn.masked_fill((text_time == 0)[:, None, :, None], 0.0)                                                 # :119-121 tokens before any image
    o = (attn @ v).permute(0, 2, 1, 3).reshape(b, L, heads * dim_head)                                               # :123-124
    y = y + torch.tanh(p["alpha_attn"]) * (o @ p["attn.to_out.weight"].t())                                          # :126, :180
    y = y + torch.tanh(p["alpha_ffw"]) * _ffw(y, p, "ffw.", act)                                                     # :182
    return y, (k, v)


# ---------------------------------------------------------------------------------------------------------------------------
# bench.py's cpu_baseline leg: let the drop-in FlamingoModel run end to end on the host cores by pointing the three entry points
# of flamingo_mini_amd.functional at the functions above (the product has no CPU path of its own).  Undo with uninstall().
# ---------------------------------------------------------------------------------------------------------------------------
_saved = {}
RS_LAYER_KEYS = ["0.norm_media.weight", "0.norm_media.bias", "0.norm_latents.weight", "0.norm_latents.bias", "0.to_q.weight", "0.to_k.weight",
                 "0.to_v.weight", "0.to_out.weight", "1.0.weight", "1.0.bias", "1.1.weight", "1.3.weight"]
XA_KEYS = ["alpha_attn", "alpha_ffw", "attn.norm.weight", "attn.norm.bias", "attn.to_q.weight", "attn.to_kv.weight", "attn.to_out.weight",
           "ffw.0.weight", "ffw.0.bias", "ffw.1.weight", "ffw.3.weight"]


def install() -> None:
    from flamingo_mini_amd import functional as F
    if _saved:
        return
    _saved.update(resampler=F.resampler, xattn_block=F.xattn_block, text_time=F.text_time, kv_project=F.kv_project)

    def rs(x_f, params: Sequence[torch.Tensor], cfg):
        depth, heads, dim_head, _, _, _, act = cfg
        keys = ["latents", "time_pos_emb", "norm.weight", "norm.bias"] + [f"layers.{i}.{k}" for i in range(depth) for k in RS_LAYER_KEYS]
        return resampler(x_f, dict(zip(keys, params)), heads=heads, dim_head=dim_head, act=act)

    def xa(y, vf, tt, params, cfg, n_visual, previous_kv=None, output_kv=False, hoisted_kv=None, wgrad=None):
        heads, dim_head, _, act = cfg
        ml = torch.diff(tt.to(torch.int64), dim=1, prepend=torch.zeros_like(tt[:, :1], dtype=torch.int64))
        out, kv = gated_xattn_block(y, vf, ml, dict(zip(XA_KEYS, params)), heads=heads, dim_head=dim_head, act=act, n_visual=n_visual,
                                    previous_kv=previous_kv)
        return out, ((kv[0].detach(), kv[1].detach()) if output_kv else None)

    F.resampler, F.xattn_block = rs, xa
    F.text_time = lambda ml: ml.to(torch.int64).cumsum(-1).to(torch.int32)
    F.kv_project = None       # the model is run with hoist_kv = False on this path


def uninstall() -> None:
    from flamingo_mini_amd import functional as F
    for k, v in _saved.items():
        setattr(F, k, v)
    _saved.clear()
