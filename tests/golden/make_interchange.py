"""Checkpoint interchange, build -> reference direction (build container only, needs /root/reference).

    python tests/golden/make_interchange.py

1. THIS repository's FlamingoModel (GPT-2-backed tiny model of full_gpt2_tiny.npz, float64, fused entry points on the numpy oracle) takes
   one optimizer step (torch AdamW, lr 1e-2) on the golden batch and exports `state_dict_trainable()`.
2. The REFERENCE FlamingoModel (same tiny architectures; only ModifiedLMBlock.forward made tolerant of positional arguments, as in
   make_golden.py) loads the original full state_dict strictly, then the exported trainable state through its own
   `flamingo.load_state_dict(..., strict=False)`: no unexpected key, and the keys it reports missing are exactly its non-trainable ones
   (`_keys_to_ignore_on_load_missing` contract, modeling_flamingo.py:125-130,376) - i.e. the two trainable key sets are identical.
3. The reference's eval logits with those weights are stored next to the exported state: tests/golden/interchange_gpt2_tiny.npz.
The parity test (tests/test_checkpoint_interchange.py) repeats step 1 and compares with this file on CPU and on the GPU.
"""
from __future__ import annotations

import os
import sys

# Bit-reproducible output: float64 BLAS reductions depend on how many threads split them, so every pool is pinned to ONE
# thread before numpy / torch load their BLAS, and torch is told to refuse nondeterministic algorithms.  With that, two runs in this
# container give byte-identical files (tests/test_oracle_golden.py::test_interchange_fixture_regenerates_bit_for_bit).
for _v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ[_v] = "1"

import numpy as np
import torch

torch.set_num_threads(1)
torch.use_deterministic_algorithms(True)

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("FLAMINGO_GOLDEN_OUT", HERE)        # where the fixture is written (the regeneration test writes to a scratch directory)
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

LR = 1e-2


def one_step_export():
    import oracle_backend
    from test_model_plumbing import build
    oracle_backend.install()
    try:
        model, z = build(torch.float64, "cpu", "gpt2")
        model.train()
        px = torch.from_numpy(z["px"]).double()
        ids, ml = torch.from_numpy(z["ids"]), torch.from_numpy(z["ml"])
        opt = torch.optim.AdamW(list(model.parameters_trainable()), lr=LR)
        model(input_ids=ids, attention_mask=torch.ones_like(ids), media_locations=ml, pixel_values=px, labels=ids).loss.backward()
        opt.step()
        return {k: v.detach().clone() for k, v in model.state_dict_trainable().items()}, z
    finally:
        oracle_backend.uninstall()


def main():
    exported, z = one_step_export()
    import make_golden as G
    G.load_reference()
    from transformers import CLIPVisionConfig, CLIPVisionModel, GPT2Config, GPT2LMHeadModel
    T = G.TINY_GPT2
    CLIPVisionModel.from_pretrained = classmethod(lambda cls, name, **kw: CLIPVisionModel(CLIPVisionConfig(**T["clip_kw"])))
    GPT2LMHeadModel.from_pretrained = classmethod(lambda cls, name, **kw: GPT2LMHeadModel(GPT2Config(**T["lm_kw"])))
    shim = sys.modules["einops_exts"]
    for name in list(sys.modules):
        if name.startswith("flamingo_mini") and not name.startswith("flamingo_mini_amd"):
            del sys.modules[name]
    sys.modules["einops_exts"] = shim
    if G.REF not in sys.path:
        sys.path.insert(0, G.REF)
    import flamingo_mini as ref
    from flamingo_mini import gated_cross_attention as gca

    def tolerant_forward(self, hidden_states, *args, use_cache=False, **kwargs):
        hidden_states, kv = self.xattn_block(y=hidden_states, visual_features=self.visual_features, media_locations=self.media_locations,
                                             previous_kv=self.xattn_layer_past, output_kv=use_cache)
        self.kv_output = kv
        return self.lm_block(hidden_states, *args, use_cache=use_cache, **kwargs)

    gca.ModifiedLMBlock.forward = tolerant_forward
    model = ref.FlamingoModel(ref.FlamingoConfig(**T["flamingo_kw"])).double()
    full = {k[3:]: torch.from_numpy(z[k]).double() for k in z["files"] if k.startswith("sd.")}
    missing, unexpected = model.load_state_dict(full, strict=False)
    assert not unexpected and all("lm_head" in k or "embed" in k for k in missing), (missing, unexpected)
    # the reference's own view of what is trainable
    ref_trainable = set(model.flamingo.state_dict_trainable())
    assert ref_trainable == set(exported), (ref_trainable ^ set(exported))
    missing, unexpected = model.flamingo.load_state_dict(exported, strict=False)
    assert not unexpected, unexpected
    assert not (set(missing) & ref_trainable), set(missing) & ref_trainable      # every trainable tensor of the reference was provided
    model.eval()
    px = torch.from_numpy(z["px"]).double()
    ids, ml = torch.from_numpy(z["ids"]), torch.from_numpy(z["ml"])
    with torch.no_grad():
        logits = model(input_ids=ids, attention_mask=torch.ones_like(ids), media_locations=ml, pixel_values=px).logits
    before = z["eval_logits"]
    assert float(np.abs(logits.numpy() - before).max()) > 1e-4, "the optimizer step must have changed the function"
    save = {"t." + k: v.numpy() for k, v in exported.items()}
    save.update(logits=logits.numpy(), lr=np.array(LR))
    np.savez_compressed(os.path.join(OUT, "interchange_gpt2_tiny.npz"), **save)
    print("interchange_gpt2_tiny:", len(exported), "trainable tensors exported by the build, loaded by the reference; logits", tuple(logits.shape),
          "max |delta logits| vs the unstepped weights", float(np.abs(logits.numpy() - before).max()))


if __name__ == "__main__":
    main()
