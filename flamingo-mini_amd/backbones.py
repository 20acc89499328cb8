"""Frozen backbones (stock HF / PyTorch-ROCm, NOT part of the accelerated path).

`from_pretrained` needs hub files; this container / the GPU box have no network, so benchmarks and tests build the
same ARCHITECTURES with random weights from the tables below (public model cards: openai/clip-vit-*, gpt2*, facebook/opt-*).
"""
from __future__ import annotations

import os

import torch

CLIP_VISION = {
    # name: hidden, layers, heads, intermediate, patch, image
    "openai/clip-vit-base-patch32": (768, 12, 12, 3072, 32, 224),
    "openai/clip-vit-base-patch16": (768, 12, 12, 3072, 16, 224),
    "openai/clip-vit-large-patch14": (1024, 24, 16, 4096, 14, 224),
}
GPT2 = {  # n_embd, n_layer, n_head
    "gpt2": (768, 12, 12), "gpt2-medium": (1024, 24, 16), "gpt2-large": (1280, 36, 20), "gpt2-xl": (1600, 48, 25),
}
OPT = {  # hidden, layers, heads, ffn, word_embed_proj_dim, do_layer_norm_before
    "facebook/opt-125m": (768, 12, 12, 3072, 768, True),
    "facebook/opt-350m": (1024, 24, 16, 4096, 512, False),
    "facebook/opt-1.3b": (2048, 24, 32, 8192, 2048, True),
    "facebook/opt-2.7b": (2560, 32, 32, 10240, 2560, True),
    "facebook/opt-6.7b": (4096, 32, 32, 16384, 4096, True),
}


class _PatchConvAsMatmul(torch.nn.Conv2d):
    """Conv2d whose kernel equals its stride (ViT patch embedding) evaluated as one matmul over unfolded patches.
    Same parameters / state_dict keys / result; avoids MIOpen's naive bf16 convolution (4.9 ms per call at 32x3x224x224 on
    MI355X vs ~0.1 ms).  Still stock PyTorch ops - the backbone itself is untouched."""

    def forward(self, x):
        p = self.kernel_size[0]
        b, c, h, w = x.shape
        if self.kernel_size != self.stride or self.padding != (0, 0) or h % p or w % p:
            return super().forward(x)
        gh, gw = h // p, w // p
        patches = x.reshape(b, c, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5).reshape(b * gh * gw, c * p * p)
        y = torch.nn.functional.linear(patches, self.weight.reshape(self.out_channels, -1), self.bias)
        return y.reshape(b, gh, gw, self.out_channels).permute(0, 3, 1, 2)


class _QuickGELU(torch.nn.Module):
    """transformers' QuickGELUActivation (x * sigmoid(1.702 x), three elementwise kernels) as one pass of the fusion library on
    the GPU for float32 / bfloat16; the stock expression everywhere else."""

    def forward(self, x):
        if x.is_cuda and x.dtype in (torch.float32, torch.bfloat16):
            from . import functional as F
            return F.quick_gelu(x)
        return x * torch.sigmoid(1.702 * x)


def stock_only(config) -> bool:
    """The default: the frozen backbones stay exactly as Hugging Face builds them (the north star's "backbones left on stock
    PyTorch-ROCm").  `FlamingoConfig(backbone_op_substitutions=True)` opts into the three result-identical op substitutions of this file
    (tests/test_hip_backbones.py pins them against the untouched modules); bench.py reports both configurations in one line."""
    return not bool(getattr(config, "backbone_op_substitutions", False))


def _tune_vision_encoder(model, config):
    if stock_only(config):
        return model
    vm = getattr(model, "vision_model", model)      # transformers < 5 nests the tower under .vision_model
    emb = vm.embeddings.patch_embedding
    if type(emb) is torch.nn.Conv2d:
        emb.__class__ = _PatchConvAsMatmul
    for layer in vm.encoder.layers:
        if type(layer.mlp.activation_fn).__name__ == "QuickGELUActivation":
            layer.mlp.activation_fn = _QuickGELU()
    return model


def _tune_gpt2(model, config):
    """HF's NewGELUActivation spells the tanh GELU as ~8 elementwise kernels; torch's fused gelu(approximate='tanh') is the
    same function in one kernel (forward and backward)."""
    if stock_only(config):
        return model
    for block in model.transformer.h:
        if type(block.mlp.act).__name__ == "NewGELUActivation":
            block.mlp.act = torch.nn.GELU(approximate="tanh")
    return model


def load_stock_gemm_tuning(path: str = None) -> bool:
    """Point PyTorch's TunableOp at a pre-tuned hipBLASLt / rocBLAS solution file for the GEMM shapes of the *stock* backbones
    (tools/tune_stock_gemms.py writes it; tuning itself stays off, so nothing is measured or written at run time).  hipBLASLt's
    default heuristic picks e.g. a 52 us kernel for GPT-2-large's 1024x1280x5120 MLP projection where a 37 us one exists.
    Returns False (and changes nothing) when the file is missing or was made for another PyTorch / ROCm / GPU."""
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950_configB.csv")
    if not (torch.cuda.is_available() and os.path.exists(path)):
        return False
    tun = torch.cuda.tunable
    tun.enable(True)
    tun.tuning_enable(False)
    ok = bool(tun.read_file(path))
    if not ok:
        tun.enable(False)
        return False
    import tempfile
    # TunableOp dumps its table to `get_filename()` at interpreter exit: keep that out of the working directory
    tun.set_filename(os.path.join(tempfile.gettempdir(), f"ff_tunableop_{os.getpid()}.csv"))
    return True


def want_random_init(config) -> bool:
    return bool(getattr(config, "random_init_backbones", False))


def _tiny_override(config, key):
    """config.backbone_overrides = {'lm': {...}, 'clip': {...}}: keyword overrides of the random-init backbone configs - tests shrink a
    backbone with it, debugging sessions pass e.g. {'lm': {'attn_implementation': 'eager', 'resid_pdrop': 0.0}} (no environment variable)"""
    return dict(getattr(config, "backbone_overrides", None) or {}).get(key, {})


def load_vision_encoder(config):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    if not want_random_init(config):
        return _tune_vision_encoder(CLIPVisionModel.from_pretrained(config.clip_model_type), config)
    kw = {}
    if config.clip_model_type in CLIP_VISION:
        hidden, layers, heads, inter, patch, image = CLIP_VISION[config.clip_model_type]
        kw = dict(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=inter,
                  patch_size=patch, image_size=image)
    elif not _tiny_override(config, "clip"):
        raise ValueError(f"no built-in architecture for {config.clip_model_type}; known: {sorted(CLIP_VISION)}")
    kw.update(_tiny_override(config, "clip"))
    return _tune_vision_encoder(CLIPVisionModel(CLIPVisionConfig(**kw)), config)


def load_language_model(config):
    """Returns the *ForCausalLM model (embedding + lm_head tied as in the released checkpoints)."""
    name = config.lm
    if not want_random_init(config):
        if name.startswith("gpt2"):
            from transformers import GPT2LMHeadModel
            return _tune_gpt2(GPT2LMHeadModel.from_pretrained(name), config)
        from transformers import OPTForCausalLM
        return OPTForCausalLM.from_pretrained(name)
    if name.startswith("gpt2"):
        from transformers import GPT2Config, GPT2LMHeadModel
        kw = {}
        if name in GPT2:
            n_embd, n_layer, n_head = GPT2[name]
            kw = dict(n_embd=n_embd, n_layer=n_layer, n_head=n_head, vocab_size=50257, n_positions=1024)
        elif not _tiny_override(config, "lm"):
            raise ValueError(f"no built-in architecture for {name}; known: {sorted(GPT2)}")
        kw.update(_tiny_override(config, "lm"))
        return _tune_gpt2(GPT2LMHeadModel(GPT2Config(**kw)), config)
    from transformers import OPTConfig, OPTForCausalLM
    kw = {}
    if name in OPT:
        hidden, layers, heads, ffn, proj, ln_before = OPT[name]
        kw = dict(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, ffn_dim=ffn, word_embed_proj_dim=proj,
                  do_layer_norm_before=ln_before, vocab_size=50272, max_position_embeddings=2048)
    elif not _tiny_override(config, "lm"):
        raise ValueError(f"no built-in architecture for {name}; known: {sorted(OPT)}")
    kw.update(_tiny_override(config, "lm"))
    return OPTForCausalLM(OPTConfig(**kw))
