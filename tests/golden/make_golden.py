"""Generate golden vectors from the REFERENCE implementation (build container only).

Runs /root/reference's own `PerceiverResampler` and `GatedCrossAttentionBlock` (forward via the modules,
backward via torch autograd) in float64 on deterministic inputs and writes `tests/golden/*.npz`.
The reference never travels to the GPU box: only these data files do.

    python tests/golden/make_golden.py            # needs /root/reference

Import recipe: `einops_exts` is absent -> 6-line in-memory shim; the three hot-path
files are loaded by path under a synthetic `flamingo_mini` package so `__init__.py` (which pulls in
transformers-dependent modules) is bypassed.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from detgen import det, resampler_params, xattn_params  # noqa: E402

REF = os.environ.get("FLAMINGO_REFERENCE", "/root/reference")


def load_reference():
    from einops import rearrange, repeat
    shim = types.ModuleType("einops_exts")
    shim.rearrange_many = lambda ts, pattern, **kw: tuple(rearrange(t, pattern, **kw) for t in ts)
    shim.repeat_many = lambda ts, pattern, **kw: tuple(repeat(t, pattern, **kw) for t in ts)
    sys.modules["einops_exts"] = shim
    pkg = types.ModuleType("flamingo_mini")
    pkg.__path__ = [os.path.join(REF, "flamingo_mini")]
    sys.modules["flamingo_mini"] = pkg
    mods = {}
    for name in ("utils", "perceiver_resampler", "gated_cross_attention"):
        spec = importlib.util.spec_from_file_location(f"flamingo_mini.{name}", os.path.join(REF, "flamingo_mini", name + ".py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"flamingo_mini.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


def t64(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def load_sd(module, params):
    sd = {k: t64(v) for k, v in params.items()}
    missing, unexpected = module.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def grads_of(module):
    return {k: p.grad.detach().numpy().copy() for k, p in module.named_parameters()}


class GateProbe:
    """Natural scale of the two scalar gate gradients of a GatedCrossAttentionBlock, d alpha = (1 - tanh^2 alpha) * sum(d branch_sum .* branch)
    (gated_cross_attention.py:180,182): || d branch_sum .* branch ||_2 * (1 - tanh^2 alpha) - the noise floor of such a sum of rounded
    products.  Stored next to the gradients as `gs.<parameter name>` so that the parity tests can hold a scalar gate gradient to
    tol * (scale + |reference|) (tests/util.py: gate_grad_ok) instead of an absolute bound.  Reads the reference module's own tensors:
    forward hooks keep the branch outputs (.attn, .ffw) and the two residual sums (the input of .ffw, the block's output) with their gradients."""

    def __init__(self, named_blocks):
        self.items = []
        for prefix, blk in named_blocks:
            rec = {"prefix": prefix, "blk": blk}
            blk.attn.register_forward_hook(lambda mod, inp, out, rec=rec: rec.__setitem__("attn", out[0].detach()))
            blk.ffw.register_forward_pre_hook(lambda mod, inp, rec=rec: self._keep(rec, "y1", inp[0]))      # y1 = y + tanh(alpha_attn) * attn_out
            blk.ffw.register_forward_hook(lambda mod, inp, out, rec=rec: rec.__setitem__("ffw", out.detach()))
            blk.register_forward_hook(lambda mod, inp, out, rec=rec: self._keep(rec, "y2", out[0]))          # y2 = y1 + tanh(alpha_ffw) * ffw_out
            self.items.append(rec)

    @staticmethod
    def _keep(rec, key, t):
        if t.requires_grad:
            t.retain_grad()
            rec[key] = t

    def scales(self):
        out = {}
        for rec in self.items:
            for branch, total, pname in (("attn", "y1", "alpha_attn"), ("ffw", "y2", "alpha_ffw")):
                b, s = rec.get(branch), rec.get(total)
                if b is None or s is None or s.grad is None:
                    continue
                th = float(torch.tanh(getattr(rec["blk"], pname).detach()))
                out["gs." + rec["prefix"] + pname] = np.array(float((s.grad * b).norm()) * (1.0 - th * th))
        return out


def resampler_case(mods, name, *, dim, depth, heads, dim_head, q, nte, ff_mult, act, xshape, store_params):
    R = mods["perceiver_resampler"].PerceiverResampler
    m = R(dim=dim, depth=depth, dim_head=dim_head, heads=heads, num_latents=q, num_time_embeds=nte,
          ff_mult=ff_mult, act=act).double()
    params = resampler_params(dim, depth, heads, dim_head, q, nte, ff_mult, tag=name)
    load_sd(m, params)
    x = t64(det(xshape, name + "x", 1.0)).requires_grad_(True)
    y = m(x)
    dy = t64(det(tuple(y.shape), name + "dy", 1.0))
    y.backward(dy)
    out = {"y": y.detach().numpy(), "dx": x.grad.numpy()}
    out.update({"g." + k: v for k, v in grads_of(m).items()})
    meta = dict(dim=dim, depth=depth, heads=heads, dim_head=dim_head, q=q, nte=nte, ff_mult=ff_mult)
    if store_params:
        out.update({"p." + k: v.astype(np.float64) for k, v in params.items()})
        out["x"] = x.detach().numpy()
        out["dy"] = dy.numpy()
    else:  # large case: regenerate inputs with detgen, store float32 results + a digest of the inputs
        out = {k: v.astype(np.float32) for k, v in out.items()}
        out["digest"] = np.array([float(sum(np.abs(v.astype(np.float64)).sum() for v in params.values())),
                                  float(np.abs(x.detach().numpy()).sum())])
    out["meta"] = np.array([meta[k] for k in ("dim", "depth", "heads", "dim_head", "q", "nte", "ff_mult")])
    out["xshape"] = np.array(xshape)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "y", tuple(y.shape), "|y|", float(y.abs().mean()))


def xattn_case(mods, name, *, dim, dv, heads, dim_head, n_visual, ff_mult, act, b, L, N, ml, store_params):
    G = mods["gated_cross_attention"].GatedCrossAttentionBlock
    m = G(dim=dim, dim_visual=dv, dim_head=dim_head, heads=heads, ff_mult=ff_mult, act=act, n_visual=n_visual).double()
    params = xattn_params(dim, dv, heads, dim_head, ff_mult, tag=name)
    load_sd(m, params)
    y = t64(det((b, L, dim), name + "y", 1.0)).requires_grad_(True)
    vf = t64(det((b, N, n_visual, dv), name + "vf", 1.0)).requires_grad_(True)
    mlt = torch.from_numpy(np.asarray(ml, dtype=np.int64))
    probe = GateProbe([("", m)])
    out_y, kv = m(y, vf, mlt, previous_kv=None, output_kv=True)
    dy = t64(det((b, L, dim), name + "dy", 1.0))
    out_y.backward(dy)
    out = {"y_out": out_y.detach().numpy(), "k": kv[0].detach().numpy(), "v": kv[1].detach().numpy(),
           "dy_in": y.grad.numpy(), "dvf": vf.grad.numpy()}
    out.update({"g." + k: v for k, v in grads_of(m).items()})
    out.update(probe.scales())
    # cached-decode path (gated_cross_attention.py:88-92,102-104): last token only, K/V reused
    with torch.no_grad():
        y_last = y[:, -1:].detach()
        out_c, _ = m(y_last, torch.zeros(b, 1, n_visual, dv, dtype=torch.float64), mlt,
                     previous_kv=(kv[0].detach(), kv[1].detach()), output_kv=False)
    out["y_out_cached_last"] = out_c.numpy()
    out["ml"] = np.asarray(ml, dtype=np.int64)
    meta = dict(dim=dim, dv=dv, heads=heads, dim_head=dim_head, n_visual=n_visual, ff_mult=ff_mult, b=b, L=L, N=N)
    if store_params:
        out.update({"p." + k: v.astype(np.float64) for k, v in params.items()})
        out["y"] = y.detach().numpy()
        out["vf"] = vf.detach().numpy()
        out["dy"] = dy.numpy()
    else:
        keep = {"ml"}
        out = {k: (v if k in keep else v.astype(np.float32)) for k, v in out.items()}
        out["digest"] = np.array([float(sum(np.abs(v.astype(np.float64)).sum() for v in params.values())),
                                  float(y.detach().abs().sum()), float(vf.detach().abs().sum())])
    out["meta"] = np.array([meta[k] for k in ("dim", "dv", "heads", "dim_head", "n_visual", "ff_mult", "b", "L", "N")])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "y_out", tuple(out_y.shape), "|delta|", float((out_y - y).abs().mean()))


def main():
    torch.manual_seed(0)
    torch.set_grad_enabled(True)
    mods = load_reference()
    toy_rs = dict(dim=64, depth=2, heads=2, dim_head=16, q=8, nte=4, ff_mult=4, store_params=True)
    resampler_case(mods, "rs_toy_gelu_T3", act="gelu", xshape=(2, 3, 10, 64), **toy_rs)
    resampler_case(mods, "rs_toy_sqrelu_3d", act="sqrelu", xshape=(2, 10, 64), **toy_rs)
    resampler_case(mods, "rs_toy_relu_T4", act="relu", xshape=(1, 4, 7, 64), **toy_rs)
    # real head geometry (8 heads x 64, 64 latents), ViT-B/32-like 50 tokens, two frames
    resampler_case(mods, "rs_geom_v50_T2", act="gelu", xshape=(2, 2, 50, 128), dim=128, depth=2, heads=8,
                   dim_head=64, q=64, nte=4, ff_mult=4, store_params=False)
    # ViT-L/14-like 257 tokens (ragged: 257+64 = 321 keys), single frame given 3-D
    resampler_case(mods, "rs_geom_v257", act="gelu", xshape=(1, 257, 128), dim=128, depth=1, heads=8,
                   dim_head=64, q=64, nte=4, ff_mult=4, store_params=False)

    # media_locations rows: (0) tags at 0 and 3 -> t=1,1,1,2,2,..; (1) leading no-media tokens and a third tag
    # with only N=2 images -> t=3 > N (fully masked row -> uniform softmax quirk); (2) no tags at all (t=0).
    ml_toy = [[1, 0, 0, 1, 0, 0, 0, 0],
              [0, 0, 1, 0, 0, 1, 0, 1],
              [0, 0, 0, 0, 0, 0, 0, 0]]
    toy_xa = dict(dim=32, dv=64, heads=2, dim_head=16, n_visual=8, ff_mult=4, b=3, L=8, N=2, ml=ml_toy, store_params=True)
    for act in ("gelu", "sqrelu", "relu"):
        xattn_case(mods, f"xa_toy_{act}", act=act, **toy_xa)
    L = 40
    ml_geom = np.zeros((2, L), dtype=np.int64)
    ml_geom[0, [0, 17]] = 1           # two images, boundary inside a 16-token tile
    ml_geom[1, [3, 20, 33]] = 1       # leading t=0 tokens, and a 3rd tag with only 2 images (quirk rows)
    xattn_case(mods, "xa_geom_L40_N2", act="gelu", dim=192, dv=128, heads=8, dim_head=64, n_visual=64, ff_mult=4,
               b=2, L=L, N=2, ml=ml_geom.tolist(), store_params=False)


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------------------------------------------
# tiny FULL model (reference FlamingoModel, OPT-backed) -> state_dict + inputs + logits / loss / grads
# ---------------------------------------------------------------------------------------------------
TINY = dict(
    lm_kw=dict(hidden_size=32, num_hidden_layers=3, num_attention_heads=2, ffn_dim=64, word_embed_proj_dim=32,
               do_layer_norm_before=True, vocab_size=96, max_position_embeddings=64, dropout=0.0),
    clip_kw=dict(hidden_size=48, num_hidden_layers=2, num_attention_heads=2, intermediate_size=96, patch_size=16, image_size=32),
    flamingo_kw=dict(lm="facebook/opt-tiny", clip_model_type="openai/clip-vit-tiny", dim=32, dim_visual=48, xattn_every=2,
                     xattn_dim_head=16, xattn_heads=2, xattn_ff_mult=2, xattn_act="sqrelu", resampler_depth=2,
                     resampler_dim_head=16, resampler_heads=2, resampler_num_latents=8, resampler_num_time_embeds=4,
                     resampler_ff_mult=2, resampler_act="gelu"),
)


def full_model_case():
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel, OPTConfig, OPTForCausalLM
    # the reference calls from_pretrained (needs the hub): build the same classes with tiny random-init configs instead
    CLIPVisionModel.from_pretrained = classmethod(lambda cls, name, **kw: CLIPVisionModel(CLIPVisionConfig(**TINY["clip_kw"])))
    OPTForCausalLM.from_pretrained = classmethod(lambda cls, name, **kw: OPTForCausalLM(OPTConfig(**TINY["lm_kw"])))
    shim = sys.modules["einops_exts"]
    for name in list(sys.modules):
        if name.startswith("flamingo_mini"):
            del sys.modules[name]
    sys.modules["einops_exts"] = shim
    sys.path.insert(0, REF)
    import flamingo_mini as ref   # the real package this time (modeling_flamingo imports transformers)
    torch.manual_seed(7)
    cfg = ref.FlamingoConfig(**TINY["flamingo_kw"])
    model = ref.FlamingoModel(cfg).double()
    with torch.no_grad():
        for i, hook in enumerate(model.flamingo.get_modified_layers()):
            hook.xattn_block.alpha_attn.fill_(0.5 - 0.2 * i)
            hook.xattn_block.alpha_ffw.fill_(-0.3 + 0.25 * i)
    model.train()
    probe = GateProbe([(n + ".", m) for n, m in model.named_modules() if type(m).__name__ == "GatedCrossAttentionBlock"])
    b, L, N = 2, 10, 2
    px = t64(det((b, N, 3, 32, 32), "full-px"))
    ids = torch.from_numpy((np.abs(det((b, L), "full-ids")) * 96).astype(np.int64) % 96)
    ml = torch.zeros(b, L, dtype=torch.long); ml[0, [0, 5]] = 1; ml[1, [2, 3, 7]] = 1   # row 1: 3 tags, 2 images -> quirk rows
    am = torch.ones(b, L, dtype=torch.long)
    out = model(input_ids=ids, attention_mask=am, media_locations=ml, pixel_values=px, labels=ids)
    out.loss.backward()
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    grads = {k: p.grad.numpy() for k, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    gate_scales = probe.scales()
    # cached two-step decode at the forward level: step 1 full prompt with use_cache, step 2 one more token
    model.eval()
    with torch.no_grad():
        o1 = model(input_ids=ids[:, :-1], attention_mask=am[:, :-1], media_locations=ml[:, :-1], pixel_values=px, use_cache=True)
        o2 = model(input_ids=ids[:, -1:], attention_mask=am, media_locations=ml, past_key_values=o1.past_key_values, use_cache=True)
        full = model(input_ids=ids, attention_mask=am, media_locations=ml, pixel_values=px)
    assert torch.allclose(o2.logits[:, -1], full.logits[:, -1], atol=1e-8)
    save = {"sd." + k: v for k, v in sd.items()}
    save.update({"g." + k: v for k, v in grads.items()})
    save.update(px=px.numpy(), ids=ids.numpy(), ml=ml.numpy(), logits=out.logits.detach().numpy(), loss=np.array(out.loss.item()),
                eval_logits=full.logits.numpy(), step2_logits=o2.logits.numpy())
    save.update(gate_scales)
    np.savez_compressed(os.path.join(HERE, "full_opt_tiny.npz"), **save)
    print("full_opt_tiny: logits", tuple(out.logits.shape), "loss", out.loss.item(), "trainable grads", len(grads),
          "transformers", transformers.__version__)


if __name__ == "__main__" and os.environ.get("FLAMINGO_GOLDEN_FULL", "1") == "1":
    full_model_case()


# ---------------------------------------------------------------------------------------------------
# tiny FULL model, GPT-2-backed, config-A-like geometry (BASELINE configs[0]: gpt2 + ViT-B/32: 50 CLIP tokens per image,
# xattn_every=1, 1 image, seq_len 32, batch 2).  transformers >= 5 calls GPT-2 blocks positionally, which the reference's
# ModifiedLMBlock.forward(hidden_states, use_cache=False, **kwargs) (gated_cross_attention.py:231-252) cannot take, so ONLY that
# method is replaced by an argument-tolerant one that runs the reference's own xattn_block + lm_block in the same order.
# Also: 6-D (video) and 4-D pixel inputs and the visual_features= hand-off (modeling_flamingo.py:153-167, 189, 212-215).
# ---------------------------------------------------------------------------------------------------
TINY_GPT2 = dict(
    lm_kw=dict(n_embd=64, n_layer=3, n_head=2, vocab_size=96, n_positions=64, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0),
    clip_kw=dict(hidden_size=48, num_hidden_layers=2, num_attention_heads=2, intermediate_size=96, patch_size=16, image_size=112),
    flamingo_kw=dict(lm="gpt2-tiny", clip_model_type="openai/clip-vit-tiny", dim=64, dim_visual=48, xattn_every=1,
                     xattn_dim_head=32, xattn_heads=2, xattn_ff_mult=2, xattn_act="gelu", resampler_depth=2,
                     resampler_dim_head=32, resampler_heads=2, resampler_num_latents=8, resampler_num_time_embeds=4,
                     resampler_ff_mult=2, resampler_act="sqrelu"),
)


def full_model_case_gpt2():
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel, GPT2Config, GPT2LMHeadModel
    CLIPVisionModel.from_pretrained = classmethod(lambda cls, name, **kw: CLIPVisionModel(CLIPVisionConfig(**TINY_GPT2["clip_kw"])))
    GPT2LMHeadModel.from_pretrained = classmethod(lambda cls, name, **kw: GPT2LMHeadModel(GPT2Config(**TINY_GPT2["lm_kw"])))
    shim = sys.modules["einops_exts"]
    for name in list(sys.modules):
        if name.startswith("flamingo_mini"):
            del sys.modules[name]
    sys.modules["einops_exts"] = shim
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import flamingo_mini as ref
    from flamingo_mini import gated_cross_attention as gca

    def tolerant_forward(self, hidden_states, *args, use_cache=False, **kwargs):
        hidden_states, kv = self.xattn_block(y=hidden_states, visual_features=self.visual_features, media_locations=self.media_locations,
                                             previous_kv=self.xattn_layer_past, output_kv=use_cache)
        self.kv_output = kv
        return self.lm_block(hidden_states, *args, use_cache=use_cache, **kwargs)

    gca.ModifiedLMBlock.forward = tolerant_forward
    torch.manual_seed(11)
    cfg = ref.FlamingoConfig(**TINY_GPT2["flamingo_kw"])
    model = ref.FlamingoModel(cfg)
    assert type(model.flamingo).__name__ == "FlamingoGPT2"
    with torch.no_grad():      # every value float32-representable, so the state_dict can be stored in float32 without loss
        for i, hook in enumerate(model.flamingo.get_modified_layers()):
            hook.xattn_block.alpha_attn.fill_(0.5 - 0.25 * i)
            hook.xattn_block.alpha_ffw.fill_(-0.375 + 0.25 * i)
    model = model.float().double()
    model.train()
    probe = GateProbe([(n + ".", m) for n, m in model.named_modules() if type(m).__name__ == "GatedCrossAttentionBlock"])
    b, L, N = 2, 32, 1
    px = t64(det((b, N, 3, 112, 112), "gpt2-px"))
    ids = torch.from_numpy((np.abs(det((b, L), "gpt2-ids")) * 96).astype(np.int64) % 96)
    ml = torch.zeros(b, L, dtype=torch.long); ml[0, 0] = 1; ml[1, [3, 17]] = 1     # row 1: leading t=0 tokens + a 2nd tag with 1 image (uniform rows)
    am = torch.ones(b, L, dtype=torch.long)
    out = model(input_ids=ids, attention_mask=am, media_locations=ml, pixel_values=px, labels=ids)
    out.loss.backward()
    sd = {k: v.detach().numpy().astype(np.float32) for k, v in model.state_dict().items()}
    for k, v in model.state_dict().items():
        assert np.array_equal(sd[k].astype(np.float64), v.detach().numpy()), k
    grads = {k: p.grad.numpy() for k, p in model.named_parameters() if p.requires_grad and p.grad is not None}
    gate_scales = probe.scales()
    model.eval()
    with torch.no_grad():
        o1 = model(input_ids=ids[:, :-1], attention_mask=am[:, :-1], media_locations=ml[:, :-1], pixel_values=px, use_cache=True)
        o2 = model(input_ids=ids[:, -1:], attention_mask=am, media_locations=ml, past_key_values=o1.past_key_values, use_cache=True)
        full = model(input_ids=ids, attention_mask=am, media_locations=ml, pixel_values=px)
        assert torch.allclose(o2.logits[:, -1], full.logits[:, -1], atol=1e-8)
        # visual_features= instead of pixel_values (modeling_flamingo.py:189,212-215): same logits
        vf = model.flamingo.encode_resample_visuals(px)
        via_vf = model(input_ids=ids, attention_mask=am, media_locations=ml, visual_features=vf)
        assert torch.allclose(via_vf.logits, full.logits, atol=1e-10)
        # 6-D video pixels (b N T c h w), T = 2 frames flattened into the resampler's key axis (:153-167)
        px6 = t64(det((b, 1, 2, 3, 112, 112), "gpt2-px6"))
        vid = model(input_ids=ids, attention_mask=am, media_locations=ml, pixel_values=px6)
        # 4-D pixels (N c h w) = ONE sequence with N = 2 images (:153-156).  forward() itself asserts pixel_values.size(0) == batch
        # (:244), so the 4-D form is only reachable through encode_resample_visuals; its output then feeds visual_features=.
        px4 = t64(det((2, 3, 112, 112), "gpt2-px4"))
        ml4 = torch.zeros(1, L, dtype=torch.long); ml4[0, [1, 9]] = 1
        vf4 = model.flamingo.encode_resample_visuals(px4)
        assert tuple(vf4.shape) == (1, 2, 8, 48)
        four = model(input_ids=ids[:1], attention_mask=am[:1], media_locations=ml4, visual_features=vf4)
    save = {"sd." + k: v for k, v in sd.items()}
    save.update({"g." + k: v for k, v in grads.items()})
    save.update(ids=ids.numpy(), ml=ml.numpy(), ml4=ml4.numpy(), logits=out.logits.detach().numpy(), loss=np.array(out.loss.item()),
                eval_logits=full.logits.numpy(), step2_logits=o2.logits.numpy(), vf=vf.numpy(), video_logits=vid.logits.numpy(),
                vf4=vf4.numpy(), four_d_logits=four.logits.numpy())
    save.update(gate_scales)
    np.savez_compressed(os.path.join(HERE, "full_gpt2_tiny.npz"), **save)
    n_rs = sum(p.numel() for p in model.flamingo.resampler.parameters())
    print("full_gpt2_tiny: logits", tuple(out.logits.shape), "loss", out.loss.item(), "trainable grads", len(grads),
          "resampler params", n_rs, "transformers", transformers.__version__)


# ---------------------------------------------------------------------------------------------------
# Full model at the FUSED bf16 kernels' geometry: 64-wide heads in the gated blocks and the resampler, LM width 256,
# 64 latents, one image + 32 tokens per sequence - small enough to store, large enough that the drop-in takes the resident fused kernels,
# the hoisted K / V projection, the deferred grouped weight gradients and the fused loss / optimizer in bfloat16.  TWO training steps of the
# reference (torch.optim.AdamW in float64): logits, loss and every trainable gradient of both.  No weights are stored: every parameter is
# detgen.det_state(name, shape) - closed-form, bf16-representable - on both sides; so are the pixels.
# ---------------------------------------------------------------------------------------------------
H64 = dict(
    lm_kw=dict(n_embd=256, n_layer=2, n_head=4, n_inner=256, vocab_size=96, n_positions=64, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0),
    clip_kw=dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64, patch_size=16, image_size=64),
    flamingo_kw=dict(lm="gpt2-h64", clip_model_type="openai/clip-vit-h64", dim=256, dim_visual=64, xattn_every=1,
                     xattn_dim_head=64, xattn_heads=2, xattn_ff_mult=1, xattn_act="gelu", resampler_depth=1,
                     resampler_dim_head=64, resampler_heads=2, resampler_num_latents=64, resampler_num_time_embeds=4,
                     resampler_ff_mult=2, resampler_act="gelu"),
    # (eps at the scale of the gradients: with the default 1e-8 the first AdamW steps are sign updates, which no finite-precision gradient can
    # reproduce element by element near zero; 1e-3 makes the update a smooth function of the gradient, so parameters can be compared)
    adamw=dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-3, weight_decay=1e-2),
)


def full_model_case_h64():
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel, GPT2Config, GPT2LMHeadModel
    from detgen import bf16_round, det_state
    CLIPVisionModel.from_pretrained = classmethod(lambda cls, name, **kw: CLIPVisionModel(CLIPVisionConfig(**H64["clip_kw"])))
    GPT2LMHeadModel.from_pretrained = classmethod(lambda cls, name, **kw: GPT2LMHeadModel(GPT2Config(**H64["lm_kw"])))
    import flamingo_mini as ref        # (full_model_case_gpt2 ran before: the package is imported and ModifiedLMBlock.forward is the tolerant one)
    cfg = ref.FlamingoConfig(**H64["flamingo_kw"])
    model = ref.FlamingoModel(cfg).double()
    assert type(model.flamingo).__name__ == "FlamingoGPT2"
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if v.dtype.is_floating_point and "lm_head" not in k:        # (lm_head.weight is the token embedding: tied)
                v.copy_(t64(det_state(k, tuple(v.shape))))
        assert model.flamingo.lm_head.weight.data_ptr() == model.flamingo.lm.wte.weight.data_ptr()
    model.train()
    probe = GateProbe([(n + ".", m) for n, m in model.named_modules() if type(m).__name__ == "GatedCrossAttentionBlock"])
    b, L = 4, 32
    px = t64(bf16_round(det((b, 1, 3, 64, 64), "h64-px")))
    ids = torch.from_numpy((np.abs(det((b, L), "h64-ids")) * 96).astype(np.int64) % 96)
    ml = torch.zeros(b, L, dtype=torch.long); ml[:, 0] = 1; ml[1, 0] = 0; ml[1, 5] = 1; ml[3, 20] = 1   # row 1: leading t = 0 tokens; row 3: a 2nd tag with 1 image (uniform rows)
    am = torch.ones(b, L, dtype=torch.long)
    opt = torch.optim.AdamW([p for p in model.parameters_trainable()], **H64["adamw"])
    save = dict(ids=ids.numpy(), ml=ml.numpy())
    for step in (1, 2):
        opt.zero_grad(set_to_none=True)
        out = model(input_ids=ids, attention_mask=am, media_locations=ml, pixel_values=px, labels=ids)
        out.loss.backward()
        grads = {k: p.grad.numpy() for k, p in model.named_parameters() if p.requires_grad and p.grad is not None}
        save[f"logits{step}"] = out.logits.detach().numpy().astype(np.float32)
        save[f"loss{step}"] = np.array(out.loss.item())
        gdt = np.float32 if step == 1 else np.float16          # step 2 is compared in bfloat16 only (weights have left the bf16 grid by then)
        save.update({f"g{step}." + k: v.astype(gdt) for k, v in grads.items()})
        save.update({f"gs{step}." + k[3:]: v for k, v in probe.scales().items()})
        opt.step()
        print(f"full_gpt2_h64 step {step}: loss", out.loss.item(), "trainable grads", len(grads))
    np.savez_compressed(os.path.join(HERE, "full_gpt2_h64.npz"), **save)
    print("full_gpt2_h64: logits", tuple(out.logits.shape), "transformers", transformers.__version__)


def param_count_pins():
    """The two known-answer parameter counts of the reference (examples/model_stats.ipynb:1605), re-derived from the reference classes."""
    mods = load_reference()
    rs = mods["perceiver_resampler"].PerceiverResampler(dim=1024, depth=6)
    xa = mods["gated_cross_attention"].GatedCrossAttentionBlock(dim=1280, dim_visual=1024)
    n_rs, n_xa = sum(p.numel() for p in rs.parameters()), sum(p.numel() for p in xa.parameters())
    assert (n_rs, n_xa) == (63023104, 15471618), (n_rs, n_xa)
    print("param counts: resampler(dim 1024, depth 6) =", n_rs, " xattn block(1280, 1024) =", n_xa)


if __name__ == "__main__" and os.environ.get("FLAMINGO_GOLDEN_FULL", "1") == "1":
    full_model_case_gpt2()
    full_model_case_h64()
    param_count_pins()
