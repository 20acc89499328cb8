"""The op substitutions inside the frozen backbones (flamingo_mini_amd/backbones.py: ViT patch convolution evaluated as a matmul, CLIP's
QuickGELU as one pass of the fusion library, HF's NewGELUActivation -> torch's fused tanh GELU) must leave the model's function alone:
the substituted model (FlamingoConfig(backbone_op_substitutions=True)) against the UNTOUCHED Hugging Face modules (the default) on the same weights and inputs - logits and
loss, fp32 and bf16 - eagerly and when the training step is replayed from a captured HIP graph."""
import os

import pytest
import torch

from detgen import det
from util import rel

pytestmark = pytest.mark.gpu


def _build(stock: bool, dtype):
    from flamingo_mini_amd import FlamingoConfig, FlamingoModel
    # flamingo-tiny's architectures (BASELINE configs[0]) with fewer layers and no dropout inside the LM (dropout draws differ between
    # an eager step and a replayed one by construction; everything else is deterministic)
    cfg = FlamingoConfig(lm="gpt2", clip_model_type="openai/clip-vit-base-patch32", dim=768, dim_visual=768, random_init_backbones=True,
                         backbone_op_substitutions=not stock,
                         backbone_overrides={"lm": dict(n_layer=4, resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0),
                                             "clip": dict(num_hidden_layers=4)})
    torch.manual_seed(7)
    model = FlamingoModel(cfg)
    with torch.no_grad():
        for hook in model.flamingo.get_modified_layers():
            hook.xattn_block.alpha_attn.fill_(0.5)
            hook.xattn_block.alpha_ffw.fill_(0.5)
    return model.to(device="cuda", dtype=dtype).train()


def _names(model):
    return {type(m).__name__ for m in model.modules()}


def _batch(dtype):
    b, L = 4, 16
    px = torch.from_numpy(det((b, 1, 3, 224, 224), "bb-px")).to(device="cuda", dtype=dtype)
    ids = (torch.from_numpy(det((b, L), "bb-ids")).abs() * 50000).long().cuda() % 50257
    ml = torch.zeros((b, L), dtype=torch.long, device="cuda"); ml[:, 0] = 1
    return dict(pixel_values=px, input_ids=ids, media_locations=ml, attention_mask=torch.ones_like(ids), labels=ids)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_substituted_backbones_equal_stock_hf_modules(dtype):
    stock, tweaked = _build(True, dtype), _build(False, dtype)
    assert "_QuickGELU" in _names(tweaked) and "_PatchConvAsMatmul" in _names(tweaked)
    assert "_QuickGELU" not in _names(stock) and "_PatchConvAsMatmul" not in _names(stock) and "NewGELUActivation" in _names(stock)
    tweaked.load_state_dict(stock.state_dict(), strict=True)          # same parameter names: the substitutions own no parameters
    batch = _batch(dtype)
    with torch.no_grad():
        out_s, out_t = stock(**batch), tweaked(**batch)
    # fp32: the tanh-GELU spellings and the unfolded convolution differ by rounding only; bf16: every op rounds its output to bf16, the
    # substitutions round ONCE where the stock expression rounds after each of its 3-8 elementwise kernels
    tol = 1e-5 if dtype == torch.float32 else 2e-2      # measured 1.4e-6 / 1.14e-2
    err = rel(out_t.logits, out_s.logits)
    print(f"[backbones {dtype}] logits rel-L2 substituted vs stock: {err:.3e}; loss {float(out_t.loss):.6f} vs {float(out_s.loss):.6f}", flush=True)
    assert err < tol
    assert abs(float(out_t.loss) - float(out_s.loss)) < (2e-5 if dtype == torch.float32 else 2e-2) * max(1.0, abs(float(out_s.loss)))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_graph_replay_of_substituted_model_follows_eager_stock_model(dtype):
    """The same training steps: eager launches on the stock model, a captured HIP graph replayed on the substituted one (what bench.py's
    default line does): the losses must agree step by step."""
    from flamingo_mini_amd import FusedAdamW, GraphedTrainStep
    tweaked = _build(False, dtype)
    batch = _batch(dtype)
    start = {k: v.detach().clone() for k, v in tweaked.state_dict().items()}
    opt_t = FusedAdamW(list(tweaked.parameters_trainable()), lr=1e-4, capturable=True)
    step = GraphedTrainStep(tweaked, opt_t, batch, warmup=1)          # one eager step inside, then replays
    losses_t = [None] + [float(step()) for _ in range(3)]
    stock = _build(True, dtype)
    stock.load_state_dict(start, strict=True)
    params_s = list(stock.parameters_trainable())
    opt_s = FusedAdamW(params_s, lr=1e-4)
    losses_s = []
    for _ in range(4):
        for p in params_s:
            p.grad = None
        loss = stock(**batch).loss
        loss.backward()
        opt_s.step()
        losses_s.append(float(loss))
    print(f"[backbones {dtype}] loss per step, stock eager {losses_s} vs substituted graph replay {losses_t}", flush=True)
    ltol = 1e-4 if dtype == torch.float32 else 3e-2
    for a, c in zip(losses_s[1:], losses_t[1:]):
        assert abs(a - c) < ltol * max(1.0, abs(a)), (losses_s, losses_t)
