"""The PRODUCT's autograd Functions, deferred weight-gradient queue, gradient buckets and data-parallel reducers on CPU tensors: the
C library is replaced by tests/host_lib.py (same signatures, raw host pointers, numpy oracle inside), everything above it is the real
flamingo_mini_amd code.  Single process: gradients equal the oracle-backed model's.  Two gloo ranks: GradientAllReducer and ShardedAdamW
on the real deferred / hoisted gradient flow (the GPU suite covers the same code on one RCCL rank only)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(backend: str):
    """The tiny GPT-2-backed golden model in float32 on the host stand-in ("host") or on the oracle-backed entry points ("oracle")."""
    import host_lib
    import oracle_backend
    from test_model_plumbing import build
    if backend == "host":
        oracle_backend.uninstall()
        host = host_lib.install()
    else:
        host_lib.uninstall()
        oracle_backend.install()
        host = None
    model, z = build(torch.float32 if backend == "host" else torch.float64, "cpu", "gpt2")
    return model.train(), z, host


def _loss(model, z, rows, dtype):
    px = torch.from_numpy(z["px"])[rows].to(dtype)
    ids = torch.from_numpy(z["ids"])[rows]
    ml = torch.from_numpy(z["ml"])[rows]
    return model(input_ids=ids, attention_mask=torch.ones_like(ids), media_locations=ml, pixel_values=px, labels=ids).loss


def _grads(model):
    return {k: p.grad.detach().double().numpy().copy() for k, p in model.named_parameters() if p.requires_grad}


def _close(a, b, tol):
    return np.linalg.norm(a - b) <= tol * max(np.linalg.norm(b), 1e-30) + 1e-12


@pytest.fixture
def clean_patches():
    yield
    import host_lib
    import oracle_backend
    host_lib.uninstall()
    oracle_backend.uninstall()


def test_product_autograd_on_the_host_library_matches_the_oracle_model(clean_patches):
    ref, z, _ = _model("oracle")
    ref.zero_grad(set_to_none=True)
    _loss(ref, z, [0, 1], torch.float64).backward()
    want = _grads(ref)

    model, z, host = _model("host")
    n_hooks = len(model.flamingo.get_modified_layers())
    for group in (0, 1):                                  # one projection call for all layers / one per layer (the data-parallel layout)
        model.flamingo.kv_project_group = group
        model.zero_grad(set_to_none=True)
        host.calls.clear()
        _loss(model, z, [0, 1], torch.float32).backward()
        got = _grads(model)
        assert set(got) == set(want)
        for k in want:
            assert np.isfinite(got[k]).all(), k             # (a deferred gradient that was never completed would still hold the NaN fill)
            assert _close(got[k], want[k], 2e-4), (k, group)
        # the real queue deferred every block and flushed them in groups of at most four at the end of backward
        assert host.calls.count("ff_xattn_block_bwd_kv_data") == n_hooks and "ff_xattn_block_bwd_kv" not in host.calls
        grouped = [int(c.split("[")[1][:-1]) for c in host.calls if c.startswith("ff_xattn_wgrad_grouped")]
        assert sum(grouped) == n_hooks and max(grouped) <= 4
        assert sum(c.startswith("ff_kv_project_bwd") for c in host.calls) == (1 if group == 0 else n_hooks)
    # gradient accumulation: a second backward onto existing .grad must not defer (autograd would add the unfilled tensors) and doubles them
    host.calls.clear()
    _loss(model, z, [0, 1], torch.float32).backward()
    assert "ff_xattn_block_bwd_kv_data" not in host.calls and host.calls.count("ff_xattn_block_bwd_kv") == n_hooks
    for k, v in _grads(model).items():
        assert _close(v, 2.0 * want[k], 2e-4), k


def test_per_layer_projection_and_cached_decoding_on_the_host_library(clean_patches):
    """hoist_kv = False (the block projects K / V itself and returns views of the library's saved buffer) and the cached decode call with
    strided K / V and the tail of text_time - through the real functional.py; generation both with the growing cache and with the
    fixed-shape decode session."""
    ref, z, _ = _model("oracle")
    ref.flamingo.hoist_kv = False
    ref.zero_grad(set_to_none=True)
    _loss(ref, z, [0, 1], torch.float64).backward()
    want = _grads(ref)
    ref.eval()
    px64 = torch.from_numpy(z["px"]).double()
    ids, ml = torch.from_numpy(z["ids"])[:, :4], torch.from_numpy(z["ml"])[:, :4]
    kw = dict(media_locations=ml, attention_mask=torch.ones_like(ids), max_length=9)
    want_tokens = ref.generate(ids, pixel_values=px64, **kw)

    model, z, host = _model("host")
    model.flamingo.hoist_kv = False
    model.zero_grad(set_to_none=True)
    _loss(model, z, [0, 1], torch.float32).backward()
    assert "ff_xattn_block_bwd" in host.calls and not any(c.startswith("ff_kv_project") for c in host.calls)
    for k, v in _grads(model).items():
        assert _close(v, want[k], 2e-4), k
    model.eval()
    px32 = torch.from_numpy(z["px"]).float()
    host.calls.clear()
    got = model.generate(ids, pixel_values=px32, static_decode=False, **kw)
    assert torch.equal(got, want_tokens)
    n_hooks = len(model.flamingo.get_modified_layers())
    assert host.calls.count("ff_xattn_block_fwd[cached]") == n_hooks * 4          # 5 new tokens: the prompt step + 4 cached steps
    assert torch.equal(model.generate(ids, pixel_values=px32, static_decode=True, **kw), want_tokens)
    model.flamingo.hoist_kv = True                                                 # K / V views of the hoisted projection feed the same cached path
    assert torch.equal(model.generate(ids, pixel_values=px32, static_decode=False, **kw), want_tokens)


def _worker(rank, world, port, out_dir, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from flamingo_mini_amd.data_parallel import GradientAllReducer, ShardedAdamW
    model, z, host = _model("host")
    out = {}
    if mode == "reduce":
        reducer = GradientAllReducer(model)
        assert model.flamingo.kv_project_group == 4
        model.flamingo.kv_project_group = 1
        model.zero_grad(set_to_none=True)
        _loss(model, z, [rank], torch.float32).backward()
        reducer.finish()
        out = {k: v for k, v in _grads(model).items()}
        # two micro-batches (the same sequence twice at half weight): the first under no_sync(), the second finds .grad in place, so the
        # blocks do not defer, autograd accumulates, and the reducer all-reduces the accumulated gradients after backward
        model.zero_grad(set_to_none=True)
        with reducer.no_sync():
            (_loss(model, z, [rank], torch.float32) / 2).backward()
        host.calls.clear()
        (_loss(model, z, [rank], torch.float32) / 2).backward()
        assert "ff_xattn_block_bwd_kv_data" not in host.calls
        reducer.finish()
        out.update({"acc." + k: v for k, v in _grads(model).items()})
        reducer.close()
    else:
        from test_data_parallel import HP, _torch_adamw
        opt = ShardedAdamW(model, update_fn=_torch_adamw, **HP)
        for _ in range(2):
            opt.zero_grad()
            _loss(model, z, [rank], torch.float32).backward()
            opt.finish_step()
        out = {k: p.detach().double().numpy().copy() for k, p in model.named_parameters() if p.requires_grad}
        opt.close()
    np.savez(os.path.join(out_dir, f"{mode}{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["reduce", "sharded"])
def test_two_gloo_ranks_on_the_product_gradient_flow(tmp_path, mode, clean_patches):
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / f"{mode}0.npz"), np.load(tmp_path / f"{mode}1.npz")
    ref, z, _ = _model("oracle")
    if mode == "reduce":
        ref.zero_grad(set_to_none=True)
        ((_loss(ref, z, [0], torch.float64) + _loss(ref, z, [1], torch.float64)) / 2).backward()
        want = _grads(ref)
    else:
        from test_data_parallel import HP
        opt = torch.optim.AdamW([p for p in ref.parameters() if p.requires_grad], **HP)
        for _ in range(2):
            ref.zero_grad(set_to_none=True)
            ((_loss(ref, z, [0], torch.float64) + _loss(ref, z, [1], torch.float64)) / 2).backward()
            opt.step()
        want = {k: p.detach().numpy() for k, p in ref.named_parameters() if p.requires_grad}
    for k in want:
        assert np.array_equal(r0[k], r1[k]), k                        # the ranks hold the same values after the exchange
        assert _close(r0[k], want[k], 5e-4 if mode == "sharded" else 2e-4), k
        if mode == "reduce":
            assert np.array_equal(r0["acc." + k], r1["acc." + k]) and _close(r0["acc." + k], want[k], 2e-4), k
