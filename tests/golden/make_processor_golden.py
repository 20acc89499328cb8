"""FlamingoProcessor against the reference's own class (build container only, needs /root/reference).

    python tests/golden/make_processor_golden.py

There are no tokenizer files offline, so the script makes its own: a tiny byte-level BPE in GPT-2's format (tokenizers.ByteLevelBPETokenizer
trained on a few lines that contain the `<image>` tag with and without a leading blank, so that "<" and " <" are two different ids, as
in GPT-2: 27 / 1279), saved under tests/golden/tiny_gpt2_tokenizer/ (vocab.json + merges.txt: data, committed).  The REFERENCE
FlamingoProcessor (flamingo_mini/flamingo_processor.py, imported from /root/reference) is then built with
`GPT2TokenizerFast.from_pretrained('gpt2')` redirected to that directory and `CLIPImageProcessor.from_pretrained` to the class defaults (=
CLIP's published preprocessing), and everything its public surface returns for a fixed set of inputs is stored in
tests/golden/processor_gpt2_tiny.npz / .json: leq_ids, encode_text in its three padding modes, media locations, prepare_caption(s),
remove_tags, __call__ on text + a synthetic image.  tests/test_processor_golden.py replays the same calls through THIS repository's class.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
TOK_DIR = os.path.join(HERE, "tiny_gpt2_tokenizer")

CORPUS = [
    "<image>a photo of a cat on a mat<EOC>", "a dog <image>and a bird in the park", "the quick brown fox jumps over the lazy dog",
    "an image <image>of two people <image>walking", "<image>the cat sat<EOC><|endoftext|>", "a < b and b > a", "x <y> z <image> w",
] * 4
TEXTS = {
    "single": "<image>a photo of a cat",
    "batch": ["<image>a dog and a bird<EOC>", "the fox <image>jumps over <image>the dog", "no tag at all", "a < b <image>"],
    "caption": "a photo of a cat on a mat",
    "captions": ["two people walking", "the lazy dog"],
    "tagged": ["<image>the cat sat<EOC><|endoftext|>", "  <image> a bird <|endoftext|><|endoftext|>"],
}


def make_tokenizer():
    from tokenizers import ByteLevelBPETokenizer
    tok = ByteLevelBPETokenizer()
    tok.train_from_iterator(CORPUS, vocab_size=330, min_frequency=2, special_tokens=["<|endoftext|>"], show_progress=False)
    os.makedirs(TOK_DIR, exist_ok=True)
    tok.save_model(TOK_DIR)
    with open(os.path.join(TOK_DIR, "tokenizer_config.json"), "w") as f:
        json.dump({"model_max_length": 1024, "bos_token": "<|endoftext|>", "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>"}, f)


def load_reference_processor():
    pkg = types.ModuleType("flamingo_mini")
    pkg.__path__ = [os.path.join(REF, "flamingo_mini")]
    sys.modules["flamingo_mini"] = pkg
    cfg_mod = types.ModuleType("flamingo_mini.configuration_flamingo")         # the processor only reads two attributes of the config
    cfg_mod.FlamingoConfig = type("FlamingoConfig", (), {})
    sys.modules["flamingo_mini.configuration_flamingo"] = cfg_mod
    spec = importlib.util.spec_from_file_location("flamingo_mini.flamingo_processor", os.path.join(REF, "flamingo_mini", "flamingo_processor.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["flamingo_mini.flamingo_processor"] = m
    spec.loader.exec_module(m)
    return m.FlamingoProcessor


def synthetic_image():
    from PIL import Image
    yy, xx = np.mgrid[0:40, 0:56]
    rgb = np.stack([(xx * 4) % 256, (yy * 6) % 256, (xx * yy) % 256], -1).astype(np.uint8)
    return Image.fromarray(rgb, "RGB")


def exercise(proc):
    """Everything the public surface returns for the fixed inputs -> (arrays, strings)."""
    arrays, strings = {"leq_ids": np.array(proc.leq_ids)}, {}
    for name, kw in (("default", {}), ("max8", {"max_length": 8}), ("len12", {"length": 12})):
        for key in ("single", "batch"):
            ids, ml, am = proc.encode_text(TEXTS[key], **kw)
            arrays[f"{name}.{key}.ids"], arrays[f"{name}.{key}.ml"], arrays[f"{name}.{key}.am"] = ids.numpy(), ml.numpy(), am.numpy()
    strings["prepare_caption"] = proc.prepare_caption(TEXTS["caption"])
    strings["prepare_captions"] = proc.prepare_captions(TEXTS["captions"])
    strings["remove_tags_str"] = proc.remove_tags(TEXTS["tagged"][0])
    strings["remove_tags_list"] = proc.remove_tags(TEXTS["tagged"])
    out = proc(images=[synthetic_image(), synthetic_image().rotate(90, expand=True)], text=TEXTS["batch"])
    strings["call_keys"] = sorted(out)
    for k, v in out.items():
        arrays["call." + k] = v.numpy()
    arrays["preprocess_images"] = proc.preprocess_images([synthetic_image()])["pixel_values"].numpy()
    ids = proc.tokenizer(proc.prepare_caption(TEXTS["caption"]), return_tensors="pt").input_ids
    arrays["caption.ids"] = ids.numpy()
    strings["caption.decoded"] = proc.tokenizer.batch_decode(ids)[0]
    strings["eoc_id"] = int(proc.tokenizer.convert_tokens_to_ids(proc.eoc_token))
    strings["vocab_size_with_eoc"] = len(proc.tokenizer)
    return arrays, strings


def main():
    make_tokenizer()
    import transformers
    from transformers import CLIPImageProcessor, GPT2TokenizerFast
    orig = GPT2TokenizerFast.from_pretrained.__func__
    GPT2TokenizerFast.from_pretrained = classmethod(lambda cls, name, *a, **k: orig(cls, TOK_DIR))
    CLIPImageProcessor.from_pretrained = classmethod(lambda cls, name, *a, **k: CLIPImageProcessor())
    Ref = load_reference_processor()
    cfg = types.SimpleNamespace(lm="gpt2", clip_model_type="openai/clip-vit-large-patch14")
    ref = Ref(cfg)
    # transformers >= 5 rejects the `padding=True` the reference passes to the IMAGE processor (flamingo_processor.py:125,140; older versions
    # ignored it): the reference's calls are kept as they are and the keyword is dropped on the way in
    real = ref.vision_processor
    ref.vision_processor = lambda images=None, return_tensors=None, padding=None: real(images=images, return_tensors=return_tensors)
    arrays, strings = exercise(ref)
    assert arrays["leq_ids"][0] != arrays["leq_ids"][1], "the synthetic tokenizer must give '<' and ' <' different ids"
    assert arrays["default.batch.ml"].sum() == 5, arrays["default.batch.ml"]        # every "<" counts, tag or not (flamingo_processor.py:118-119): 1 + 2 + 0 + 2
    np.savez_compressed(os.path.join(HERE, "processor_gpt2_tiny.npz"), **arrays)
    with open(os.path.join(HERE, "processor_gpt2_tiny.json"), "w") as f:
        json.dump({"texts": TEXTS, "expected": strings}, f, indent=1)
    print("leq_ids", arrays["leq_ids"], "| media locations of the batch:", arrays["default.batch.ml"].sum(1), "|", strings["prepare_caption"])


if __name__ == "__main__":
    main()
