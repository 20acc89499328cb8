"""FlamingoProcessor — text / image pre-processing with the same public surface as the reference class
(flamingo_mini/flamingo_processor.py): `tokenizer`, `vision_processor`, `leq_ids`, `encode_text`, `prepare_caption(s)`,
`remove_tags`, `get_media_locations`, `preprocess_images`, `__call__`.

Nothing here is on the accelerated path: it runs on the host, once per batch.  The only arithmetic is the
media-location finder, which marks every token that is the "<" opening an `<image>` tag; the two spellings of that
token (with / without a leading blank) are looked up from the tokenizer, and the well-known ids are kept in
KNOWN_LEQ_IDS so media locations can be derived for pre-tokenised data without tokenizer files.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple, Union

import torch

from .configuration_flamingo import FlamingoConfig

KNOWN_LEQ_IDS = {"gpt2": (27, 1279), "facebook/opt": (51552, 28696)}   # ("<", " <")
_IMAGE_TAG = "<image>"


def _load_tokenizer(lm: str, use_fast: bool):
    """GPT-2 BPE for gpt2*, the OPT tokenizer (published with opt-30b) for facebook/opt-*."""
    if lm.startswith("gpt2"):
        import transformers
        cls = transformers.GPT2TokenizerFast if use_fast else transformers.GPT2Tokenizer
        return cls.from_pretrained("gpt2")
    if lm.startswith("facebook/opt"):
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained("facebook/opt-30b", use_fast=use_fast)
    raise ValueError(f"unsupported language model {lm}")


def _load_image_processor(name: str):
    from transformers import CLIPImageProcessor
    try:
        return CLIPImageProcessor.from_pretrained(name)
    except Exception:
        # no hub access: the class defaults are exactly CLIP's preprocessing (224 px, bicubic, CLIP mean / std)
        return CLIPImageProcessor()


def media_locations_from_ids(input_ids: torch.Tensor, leq_ids: Iterable[int]) -> torch.Tensor:
    """Same shape as `input_ids`; 1 where the id is one of `leq_ids`, else 0 (dtype of input_ids)."""
    marks = torch.zeros_like(input_ids)
    for token_id in leq_ids:
        marks += input_ids.eq(token_id).to(marks.dtype)
    return marks


class FlamingoProcessor:
    def __init__(self, config: FlamingoConfig, use_fast: bool = True, eoc_token: str = "<EOC>"):
        self.config = config
        self.eoc_token = eoc_token
        self.vision_processor = _load_image_processor(config.clip_model_type)
        tok = _load_tokenizer(config.lm, use_fast)
        tok.add_bos_token = True
        tok.pad_token = tok.eos_token            # pad with EOS, as the LM has no pad token
        tok.add_tokens(eoc_token)                # end-of-chunk marker: the extra embedding row of FlamingoGPT2 / FlamingoOPT
        self.tokenizer = tok
        self.leq_ids = [tok.encode(s)[-1] for s in ("<", " <")]

    # ------------------------------------------------------------------ text
    def _tokenizer_kwargs(self, max_length, length, return_attention_mask):
        if length is not None:           # fixed length: pad AND truncate to `length`
            return dict(padding="max_length", truncation=True, max_length=length, return_attention_mask=return_attention_mask)
        if max_length is not None:       # pad to the longest, cut at `max_length`
            return dict(padding=True, truncation=True, max_length=max_length, return_attention_mask=return_attention_mask)
        return dict(padding=True)

    def encode_text(self, text: Union[str, Sequence[str]], device=None, max_length=None, length=None, return_tensors="pt",
                    return_attention_mask=True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """-> (input_ids, media_locations, attention_mask), all (batch, tokens) on `device`."""
        batch = self.tokenizer(text, return_tensors=return_tensors, **self._tokenizer_kwargs(max_length, length, return_attention_mask))
        ids = batch.input_ids
        return ids.to(device), self.get_media_locations(ids).to(device), batch.attention_mask.to(device)

    def get_media_locations(self, input_ids: torch.Tensor) -> torch.Tensor:
        return media_locations_from_ids(input_ids, self.leq_ids)

    media_locations_from_ids = staticmethod(media_locations_from_ids)

    def prepare_caption(self, caption: str) -> str:
        """Training text of one image: tag, caption, end-of-chunk, EOS (BOS is added by the tokenizer itself)."""
        return f"{_IMAGE_TAG}{caption}{self.eoc_token}{self.tokenizer.eos_token}"

    def prepare_captions(self, captions: List[str]) -> List[str]:
        return list(map(self.prepare_caption, captions))

    def _remove_tags(self, text: str) -> str:
        for marker in (_IMAGE_TAG, self.tokenizer.eos_token, self.eoc_token, self.tokenizer.pad_token):
            text = text.replace(marker, "")
        return text.strip()

    def remove_tags(self, text: Union[str, List[str]]):
        if isinstance(text, str):
            return self._remove_tags(text)
        return [self._remove_tags(item) for item in text]

    # ------------------------------------------------------------------ images
    def preprocess_images(self, images):
        """PIL images (or tensors) -> BatchFeature with `pixel_values` (n, 3, 224, 224).  (The reference passes `padding=True` as well,
        flamingo_processor.py:125,140: image processors of transformers < 5 ignore the keyword, those of >= 5 reject it.)"""
        try:
            return self.vision_processor(images=images, return_tensors="pt", padding=True)
        except TypeError:
            return self.vision_processor(images=images, return_tensors="pt")

    def __call__(self, images=None, text=None, device=None) -> dict:
        out = {}
        if images is not None:
            out["pixel_values"] = self.preprocess_images(images)["pixel_values"].to(device)
        if text is not None:
            out["input_ids"], out["media_locations"], out["attention_mask"] = self.encode_text(text, device=device)
        return out
