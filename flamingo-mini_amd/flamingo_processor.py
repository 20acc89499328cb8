"""FlamingoProcessor drop-in (reference: flamingo_mini/flamingo_processor.py): CLIP image preprocessing + GPT-2 / OPT
tokenizer with the extra <EOC> token and the media-location finder.  CPU-side; nothing here is accelerated.
Tokenizer files come from the HF hub (or its local cache) exactly like the reference; without them only the
image side and `media_locations_from_ids` are usable."""
from __future__ import annotations

from typing import List, Tuple

import torch

from .configuration_flamingo import FlamingoConfig

# id of "<" without / with a preceding blank (reference :53-57) — lets media locations be derived without a tokenizer
KNOWN_LEQ_IDS = {"gpt2": (27, 1279), "facebook/opt": (51552, 28696)}


class FlamingoProcessor:
    def __init__(self, config: FlamingoConfig, use_fast: bool = True, eoc_token: str = '<EOC>'):
        from transformers import CLIPImageProcessor
        self.config = config
        self.eoc_token = eoc_token
        try:
            self.vision_processor = CLIPImageProcessor.from_pretrained(config.clip_model_type)
        except Exception:   # offline: the class defaults ARE the CLIP ViT-B/32, L/14 preprocessing (224 px, CLIP mean/std)
            self.vision_processor = CLIPImageProcessor()
        if config.lm.startswith('gpt2'):
            from transformers import GPT2Tokenizer, GPT2TokenizerFast
            self.tokenizer = (GPT2TokenizerFast if use_fast else GPT2Tokenizer).from_pretrained('gpt2')
        elif config.lm.startswith('facebook/opt'):
            from transformers import AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained('facebook/opt-30b', use_fast=use_fast)
        else:
            raise ValueError(f"unsupported language model {config.lm}")
        self.tokenizer.add_bos_token = True
        self.tokenizer.pad_token = self.tokenizer.eos_token
        self.tokenizer.add_tokens(self.eoc_token)
        self.leq_ids = [self.tokenizer.encode("<")[-1], self.tokenizer.encode(" <")[-1]]

    # ---- text ----
    def encode_text(self, text, device=None, max_length=None, length=None, return_tensors='pt',
                    return_attention_mask=True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        kw = dict(return_tensors=return_tensors)
        if length is not None:
            kw.update(return_attention_mask=return_attention_mask, padding='max_length', truncation=True, max_length=length)
        elif max_length is None:
            kw.update(padding=True)
        else:
            kw.update(return_attention_mask=return_attention_mask, padding=True, truncation=True, max_length=max_length)
        enc = self.tokenizer(text, **kw)
        media = self.get_media_locations(enc.input_ids)
        return enc.input_ids.to(device), media.to(device), enc.attention_mask.to(device)

    def prepare_caption(self, caption: str) -> str:
        # BOS is added by the tokenizer, EOS is not
        return "<image>" + caption + self.eoc_token + self.tokenizer.eos_token

    def prepare_captions(self, captions: List[str]) -> List[str]:
        return [self.prepare_caption(c) for c in captions]

    def _remove_tags(self, text: str) -> str:
        for tag in ('<image>', self.tokenizer.eos_token, self.eoc_token, self.tokenizer.pad_token):
            text = text.replace(tag, '')
        return text.strip()

    def remove_tags(self, text):
        return self._remove_tags(text) if isinstance(text, str) else [self._remove_tags(t) for t in text]

    def get_media_locations(self, input_ids: torch.Tensor) -> torch.Tensor:
        """1 where a token is the '<' that opens an <image> tag (either spelling), else 0."""
        return self.media_locations_from_ids(input_ids, self.leq_ids)

    @staticmethod
    def media_locations_from_ids(input_ids: torch.Tensor, leq_ids) -> torch.Tensor:
        hit = torch.zeros_like(input_ids)
        for tok in leq_ids:
            hit = hit + (input_ids == tok).to(input_ids.dtype)
        return hit

    # ---- images ----
    def preprocess_images(self, images):
        return self.vision_processor(images=images, return_tensors="pt", padding=True)

    def __call__(self, images=None, text=None, device=None):
        result = {}
        if images is not None:
            result['pixel_values'] = self.vision_processor(images=images, return_tensors='pt', padding=True)['pixel_values'].to(device)
        if text is not None:
            ids, media, mask = self.encode_text(text, device=device)
            result.update(input_ids=ids, media_locations=media, attention_mask=mask)
        return result
