"""FlamingoConfig — same 18 fields and defaults as the reference (flamingo_mini/configuration_flamingo.py:6-25), so
configs and checkpoints interchange.  The reference defaults are not self-consistent (lm='gpt2' has hidden size 768
while dim defaults to 1024): always pass dim / dim_visual explicitly.

Extra, optional key understood by this implementation only (stored like any other HF config kwarg):
    random_init_backbones (bool): build CLIP / LM from built-in architecture tables with random weights when the
    HF hub files are not available (benchmarks, tests).  Default False = `from_pretrained` like the reference.
    backbone_overrides (dict): {'lm': {...}, 'clip': {...}} keyword overrides of those random-init backbone configs.
    backbone_op_substitutions (bool): default False = the frozen CLIP / LM stay exactly as Hugging Face builds them (stock
    PyTorch-ROCm); True swaps three ops inside them for result-identical faster forms (backbones.py).
"""
from __future__ import annotations

from transformers.configuration_utils import PretrainedConfig


class FlamingoConfig(PretrainedConfig):
    model_type = "flamingo"

    def __init__(
        self,
        lm: str = "gpt2",                                       # 'gpt2*' or 'facebook/opt-*'
        clip_model_type: str = "openai/clip-vit-base-patch32",
        dim: int = 1024,                                        # LM hidden size
        dim_visual: int = 768,                                  # vision encoder hidden size
        xattn_every: int = 1,                                   # gated xattn block in front of every n-th LM layer
        xattn_dim_head: int = 64,
        xattn_heads: int = 8,
        xattn_ff_mult: int = 4,
        xattn_act: str = "gelu",                                # 'gelu' | 'sqrelu' | 'relu'
        resampler_depth: int = 6,
        resampler_dim_head: int = 64,
        resampler_heads: int = 8,
        resampler_num_latents: int = 64,
        resampler_num_time_embeds: int = 4,
        resampler_ff_mult: int = 4,
        resampler_act: str = "gelu",
        freeze_language_model: bool = True,
        freeze_vision_model: bool = True,
        **kwargs,
    ):
        kwargs.setdefault("tie_word_embeddings", True)         # lm_head shares the LM's token embedding (GPT-2 / OPT): see _tied_weights_keys
        super().__init__(**kwargs)
        own = dict(locals())
        for name in ("self", "kwargs", "__class__"):
            own.pop(name, None)
        for name, value in own.items():
            setattr(self, name, value)
