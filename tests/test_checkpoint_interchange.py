"""Checkpoint interchange, both directions (reference modeling_flamingo.py:125-130 `state_dict_trainable`, :376
`_keys_to_ignore_on_load_missing`).

reference -> build: every full-model test loads a REFERENCE state_dict by name (tests/test_model_plumbing.py::build).
build -> reference: tests/golden/make_interchange.py let this repository's model take one optimizer step, exported its
`state_dict_trainable()`, loaded it into the reference FlamingoModel through the reference's own load_state_dict (identical trainable key
sets, nothing unexpected) and stored the reference's logits with those weights (interchange_gpt2_tiny.npz).  Here: the same step is
repeated and must reproduce the exported tensors; a fresh model that loads the exported state must reproduce the REFERENCE's logits - on
the host (oracle-backed entry points) and on the GPU (HIP kernels); and save_pretrained -> from_pretrained round-trips."""
import os

import numpy as np
import pytest
import torch

from util import GOLDEN, rel


def _fixture():
    z = np.load(os.path.join(GOLDEN, "interchange_gpt2_tiny.npz"))
    return {k[2:]: z[k] for k in z.files if k.startswith("t.")}, z["logits"], float(z["lr"])


def _inputs(z, device, dtype):
    px = torch.from_numpy(z["px"]).to(device=device, dtype=dtype)
    ids, ml = torch.from_numpy(z["ids"]).to(device), torch.from_numpy(z["ml"]).to(device)
    return dict(input_ids=ids, attention_mask=torch.ones_like(ids), media_locations=ml, pixel_values=px)


def test_exported_trainable_state_is_what_the_reference_loaded_and_reproduces_its_logits():
    import oracle_backend
    from test_model_plumbing import build
    exported, ref_logits, lr = _fixture()
    oracle_backend.install()
    try:
        model, z = build(torch.float64, "cpu", "gpt2")
        batch = _inputs(z, "cpu", torch.float64)
        model.train()
        opt = torch.optim.AdamW(list(model.parameters_trainable()), lr=lr)
        model(labels=batch["input_ids"], **batch).loss.backward()
        opt.step()
        mine = model.state_dict_trainable()
        assert set(mine) == set(exported)                        # the key set the reference accepted as ITS trainable set
        for k, v in mine.items():
            assert rel(v, exported[k]) < 1e-9, k
        fresh, _ = build(torch.float64, "cpu", "gpt2")          # a model that never trained: import the checkpoint
        missing, unexpected = fresh.flamingo.load_state_dict({k: torch.from_numpy(v) for k, v in exported.items()}, strict=False)
        assert not unexpected and not (set(missing) & set(exported))
        fresh.eval()
        with torch.no_grad():
            assert rel(fresh(**batch).logits, ref_logits) < 1e-9
    finally:
        oracle_backend.uninstall()


def test_save_pretrained_from_pretrained_round_trip(tmp_path):
    import oracle_backend
    from flamingo_mini_amd import FlamingoModel
    from test_model_plumbing import build
    oracle_backend.install()
    try:
        model, z = build(torch.float64, "cpu", "gpt2")
        exported, ref_logits, _ = _fixture()
        model.flamingo.load_state_dict({k: torch.from_numpy(v) for k, v in exported.items()}, strict=False)
        model.save_pretrained(str(tmp_path))
        again = FlamingoModel.from_pretrained(str(tmp_path)).double()
        sd_a, sd_b = model.state_dict(), again.state_dict()
        assert set(sd_a) == set(sd_b)
        for k in sd_a:
            assert torch.equal(sd_a[k], sd_b[k].to(sd_a[k].dtype)), k
        assert again.config.to_dict()["xattn_every"] == model.config.xattn_every and again.config.resampler_act == model.config.resampler_act
        again.eval()
        with torch.no_grad():
            assert rel(again(**_inputs(z, "cpu", torch.float64)).logits, ref_logits) < 1e-9
    finally:
        oracle_backend.uninstall()


@pytest.mark.gpu
def test_exported_state_reproduces_the_reference_logits_on_hip():
    from test_model_plumbing import build
    exported, ref_logits, _ = _fixture()
    model, z = build(torch.float32, "cuda", "gpt2")
    model.flamingo.load_state_dict({k: torch.from_numpy(v).float() for k, v in exported.items()}, strict=False)
    model.eval()
    with torch.no_grad():
        assert rel(model(**_inputs(z, "cuda", torch.float32)).logits, ref_logits) < 1e-4
