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
    from flamingo_mini_amd import functional as F
    group = F._wgrad_queue.group
    yield
    import host_lib
    import oracle_backend
    host_lib.uninstall()
    oracle_backend.uninstall()
    F._wgrad_queue.group = group


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


def _batch(z, rows, dtype):
    ids = torch.from_numpy(z["ids"])[rows]
    return dict(input_ids=ids, attention_mask=torch.ones_like(ids), media_locations=torch.from_numpy(z["ml"])[rows],
                pixel_values=torch.from_numpy(z["px"])[rows].to(dtype), labels=ids)


def test_layerwise_resampler_equals_the_stack_level_call_and_buckets_per_layer(clean_patches):
    """SURVEY 8-b2 as written: the resampler driven layer by layer (ff_resampler_prologue / _layer / _epilogue, one autograd node per layer)
    computes what the stack-level call computes - loss and every gradient, incl. d latents (the batch-sum of layer 0's input gradient),
    d time_pos_emb (from d x_f accumulated over the layers outside autograd) and d pixel-side x_f - and announces one gradient bucket per
    layer, in backward order, each when its own layer's backward has run (the stack-level call announces ONE bucket at the very end)."""
    from flamingo_mini_amd import functional as F
    ref, z, _ = _model("oracle")
    ref.zero_grad(set_to_none=True)
    ref_loss = _loss(ref, z, [0, 1], torch.float64)
    ref_loss.backward()
    want = _grads(ref)
    model, z, host = _model("host")
    rs = model.flamingo.resampler
    depth = rs.depth
    rs.layerwise = True
    names = {id(p): n for n, p in rs.named_parameters()}
    buckets = []
    cb = lambda flat, owners: buckets.append((len(host.calls), sorted(names[id(p)] for p, _, _ in owners if id(p) in names)))
    F.add_grad_ready_callback(cb)
    try:
        model.zero_grad(set_to_none=True)
        host.calls.clear()
        loss = _loss(model, z, [0, 1], torch.float32)
        loss.backward()
    finally:
        F.remove_grad_ready_callback(cb)
    assert abs(float(loss) - float(ref_loss)) < 1e-4
    got = _grads(model)
    assert set(got) == set(want)
    for k in want:
        assert np.isfinite(got[k]).all() and _close(got[k], want[k], 2e-4), k
    assert "ff_resampler_fwd" not in host.calls and "ff_resampler_bwd" not in host.calls
    assert host.calls.count("ff_resampler_layer_fwd") == depth and host.calls.count("ff_resampler_layer_bwd") == depth
    rs_buckets = [b for b in buckets if b[1]]
    # epilogue (norm), then the layers from the last to the first, then the prologue (latents, time_pos_emb)
    assert rs_buckets[0][1] == ["norm.bias", "norm.weight"]
    for i, (_, owners) in enumerate(rs_buckets[1:1 + depth]):
        layer = depth - 1 - i
        assert len(owners) == 12 and all(n.startswith(f"layers.{layer}.") for n in owners), (layer, owners)
    assert rs_buckets[1 + depth][1] == ["latents", "time_pos_emb"] and len(rs_buckets) == depth + 2
    # every layer's bucket was announced before the next layer's backward call was made
    bwd_calls = [i for i, c in enumerate(host.calls) if c == "ff_resampler_layer_bwd"]
    for (at, _), nxt in zip(rs_buckets[1:depth], bwd_calls[1:]):
        assert at <= nxt
    # with cut points between the layers every layer is a backward segment of its own: same gradients
    from flamingo_mini_amd.graphs import PiecewiseGraphedTrainStep
    step = PiecewiseGraphedTrainStep(model, None, _batch(z, [0, 1], torch.float32), capture=False, segment_layers=1)
    assert rs.autograd_cut is not None
    step()
    n_pairs = depth - 1 + 1 + (len(model.flamingo.get_modified_layers()) - 1)
    got = _grads(model)
    for k in want:
        assert _close(got[k], want[k], 2e-4), k
    step.close()
    assert rs.autograd_cut is None


def test_segmented_backward_equals_one_backward_on_the_product_gradient_flow(clean_patches):
    """graphs.PiecewiseGraphedTrainStep(capture=False): the visual features and the hidden state in front of every gated layer become cut
    points, backward runs as one autograd call per segment (each with its own flush of the deferred weight gradients) - the gradients,
    including the tied token embedding's, which two segments accumulate into, must be those of a single backward pass."""
    from flamingo_mini_amd.graphs import PiecewiseGraphedTrainStep
    ref, z, _ = _model("oracle")
    ref.zero_grad(set_to_none=True)
    ref_loss = _loss(ref, z, [0, 1], torch.float64)
    ref_loss.backward()
    want = _grads(ref)
    model, z, host = _model("host")
    n_hooks = len(model.flamingo.get_modified_layers())
    step = PiecewiseGraphedTrainStep(model, None, _batch(z, [0, 1], torch.float32), capture=False, segment_layers=1)
    host.calls.clear()
    loss = step()
    assert abs(float(loss) - float(ref_loss)) < 1e-4
    got = _grads(model)
    assert set(got) == set(want)
    for k in want:
        assert np.isfinite(got[k]).all() and _close(got[k], want[k], 2e-4), k
    # one segment per gated layer: every block's weight gradients were flushed by its own segment's end-of-pass callback
    grouped = [int(c.split("[")[1][:-1]) for c in host.calls if c.startswith("ff_xattn_wgrad_grouped")]
    assert grouped == [1] * n_hooks, grouped
    assert host.calls.index("ff_resampler_bwd") > max(i for i, c in enumerate(host.calls) if c.startswith("ff_xattn_wgrad_grouped"))
    # While the step is not running, the installed cut points are inert: an ordinary training-mode backward on the same model
    # reaches every layer and the resampler, and records no pairs.
    model.zero_grad(set_to_none=True)
    plain = _loss(model, z, [0, 1], torch.float32)
    plain.backward()
    assert step.cuts.pairs == []
    got = _grads(model)
    for k in want:
        assert np.isfinite(got[k]).all() and _close(got[k], want[k], 2e-4), k
    step.close()
    assert all(h.autograd_cut is None for h in model.flamingo.get_modified_layers())
    step.close()                                            # idempotent


def test_a_piecewise_step_that_fails_to_capture_leaves_model_and_reducer_as_it_found_them(clean_patches, monkeypatch):
    """bench.py falls back to another launch mode ON THE SAME model and reducer when the piecewise capture raises.  The failed
    constructor must have ended the reducer's recording mode (a reducer left recording never exchanges a bucket again) and removed the cut
    points (an ordinary backward would stop at the top segment), and must have restored the K / V projection group."""
    from flamingo_mini_amd import graphs
    model, z, _ = _model("host")

    class Recorder:
        def __init__(self):
            self.collecting = False
        def begin_collect(self):
            self.collecting = True
        def end_collect(self):
            self.collecting = False
            return []

    red = Recorder()
    kv_before = model.flamingo.kv_project_group

    def failing_capture(self, model_, optimizer, reducer, warmup):      # what a capture does up to the point where a segment's capture raises
        reducer.begin_collect()
        raise RuntimeError("capture refused")

    monkeypatch.setattr(graphs.PiecewiseGraphedTrainStep, "_capture", failing_capture)
    with pytest.raises(RuntimeError, match="capture refused"):
        graphs.PiecewiseGraphedTrainStep(model, None, _batch(z, [0, 1], torch.float32), capture=True, segment_layers=1, reducer=red)
    assert not red.collecting
    assert all(h.autograd_cut is None for h in model.flamingo.get_modified_layers())
    assert model.flamingo.kv_project_group == kv_before


def test_uninstalling_cuts_after_a_reducer_closed_does_not_resurrect_a_stale_kv_group(clean_patches):
    """ADVICE r04 (order hazard): cuts installed (K / V group 0 -> segment size), then a reducer-style change of the group, then the cuts
    removed: the group the cuts had saved is stale by then and must not be written back."""
    model, z, _ = _model("host")
    from flamingo_mini_amd.graphs import AutogradCuts
    assert model.flamingo.kv_project_group == 0
    model.install_autograd_cuts(AutogradCuts(), 2)
    assert model.flamingo.kv_project_group == 2
    model.flamingo.kv_project_group = 1                     # somebody else's setting, made after the cuts were installed
    model.install_autograd_cuts(None)
    assert model.flamingo.kv_project_group == 1


def test_reducers_restore_the_models_launch_structure_on_close(clean_patches):
    """A reducer with collectives switches ITS model to the bucket-friendly launch structure (4 layers per weight-gradient group and per
    K / V projection call) and close() puts back what was there - nothing process-wide is touched."""
    from flamingo_mini_amd import functional as F
    from flamingo_mini_amd.data_parallel import _bucket_launch_structure
    model, z, _ = _model("host")
    before = F._wgrad_queue.group
    model.set_launch_structure(wgrad_group=7)
    undo = _bucket_launch_structure(model)
    blocks = [h.xattn_block for h in model.flamingo.get_modified_layers()]
    assert model.flamingo.kv_project_group == 4 and all(b.wgrad_group == 4 for b in blocks) and F._wgrad_queue.group == before
    from flamingo_mini_amd.data_parallel import _restore_launch_structure
    # what the user changes AFTER the reducer was built survives close(): one block's group, and a key the reducer never touched
    blocks[0].wgrad_group = 2
    model.set_launch_structure(hoist_kv=False)
    _restore_launch_structure(undo)
    assert model.flamingo.kv_project_group == 0 and blocks[0].wgrad_group == 2 and all(b.wgrad_group == 7 for b in blocks[1:])
    assert model.flamingo.hoist_kv is False


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
        # close() takes back what the reducer set where it still stands (every block's weight-gradient group), and leaves alone what was
        # changed afterwards (the K / V projection group, set to 1 above)
        assert model.flamingo.kv_project_group == 1 and all(h.xattn_block.wgrad_group is None for h in model.flamingo.get_modified_layers())
    elif mode == "piecewise":
        # the segmented step (graphs.PiecewiseGraphedTrainStep, eager launches on CPU ranks): bucket exchanges are issued from inside each
        # segment's backward, the tied embedding - accumulated by two segments - is exchanged once, in finish()
        from flamingo_mini_amd.graphs import PiecewiseGraphedTrainStep
        reducer = GradientAllReducer(model)
        step = PiecewiseGraphedTrainStep(model, None, _batch(z, [rank], torch.float32), capture=False, segment_layers=1, reducer=reducer)
        step()
        out = {k: v for k, v in _grads(model).items()}
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


@pytest.mark.parametrize("mode", ["reduce", "sharded", "piecewise"])
def test_two_gloo_ranks_on_the_product_gradient_flow(tmp_path, mode, clean_patches):
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / f"{mode}0.npz"), np.load(tmp_path / f"{mode}1.npz")
    ref, z, _ = _model("oracle")
    if mode in ("reduce", "piecewise"):
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


# ---- the deferred weight-gradient queue under failure and module reuse ---------------------------------------
def _host_blocks(n, tag):
    import host_lib
    import oracle_backend
    from detgen import xattn_params
    from flamingo_mini_amd import GatedCrossAttentionBlock
    oracle_backend.uninstall()
    host = host_lib.install()
    dim, dv, heads, dh, nv, ffm = 32, 24, 2, 8, 4, 2
    blocks = []
    for i in range(n):
        m = GatedCrossAttentionBlock(dim=dim, dim_visual=dv, dim_head=dh, heads=heads, ff_mult=ffm, n_visual=nv)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in xattn_params(dim, dv, heads, dh, ffm, tag=f"{tag}{i}").items()})
        blocks.append(m)
    return host, blocks, (dim, dv, heads, dh, nv, ffm)


def _oracle_chain(blocks, order, y, vf, ml, g):
    """float64 oracle of h = block_order[-1](... block_order[0](y)): parameter gradients summed per block, d y, d vf."""
    from oracle import flamingo_oracle as O
    heads, dh = blocks[0].cfg[0], blocks[0].cfg[1]
    ps = [{k: v.detach().double().numpy() for k, v in m.state_dict().items()} for m in blocks]
    h, caches = y.astype(np.float64), []
    for i in order:
        h, _, c = O.gated_xattn_block_fwd(h, vf.astype(np.float64), ml, ps[i], heads=heads, dim_head=dh, n_visual=blocks[0].n_visual)
        caches.append(c)
    grads = [dict() for _ in blocks]
    d, dvf = g.astype(np.float64), 0.0
    for i, c in reversed(list(zip(order, caches))):
        d, dv_i, gi = O.gated_xattn_block_bwd(d, c, ps[i], heads=heads, dim_head=dh)
        dvf = dvf + dv_i
        for k, v in gi.items():
            grads[i][k] = grads[i].get(k, 0.0) + np.asarray(v)
    return grads, d, dvf


def _run_chain(blocks, order, y, vf, ml, g, hook_at=None):
    from flamingo_mini_amd import functional as F
    yt = torch.from_numpy(y).requires_grad_(True)
    vft = torch.from_numpy(vf).requires_grad_(True)
    mlt = torch.from_numpy(ml)
    kvs = F.kv_project(vft, [m.attn.to_kv.weight for m in blocks])
    h = yt
    for n, i in enumerate(order):
        h, _ = blocks[i](h, vft, mlt, hoisted_kv=kvs[i])
        if hook_at == n:
            def boom(grad):
                raise RuntimeError("simulated failure inside backward")
            h.register_hook(boom)
    (h * torch.from_numpy(g)).sum().backward()
    return yt.grad, vft.grad


def test_a_backward_pass_that_raises_leaves_no_stale_weight_gradient_work(clean_patches):
    """backward() raising after some blocks deferred their weight gradients: the engine never runs that pass's queue callback, so its
    entries (raw addresses of buffers that are freed afterwards) must neither run later nor be grouped with the next pass's entries,
    and the next pass must flush its own trailing group (a 'catch the error, skip the batch' loop)."""
    from detgen import det
    from flamingo_mini_amd import functional as F
    host, blocks, (dim, dv, heads, dh, nv, ffm) = _host_blocks(5, "qf")
    F._wgrad_queue.group = 4            # (the single-GPU default batches up to 12 blocks: here a full group and a trailing one are wanted)
    b, L = 2, 6
    y, vf, g = det((b, L, dim), "qf-y"), det((b, 1, nv, dv), "qf-vf"), det((b, L, dim), "qf-g")
    ml = np.zeros((b, L), np.int64); ml[:, 0] = 1
    order = list(range(5))
    with pytest.raises(RuntimeError, match="simulated failure"):
        _run_chain(blocks, order, y, vf, ml, g, hook_at=1)          # blocks 4, 3, 2 defer (3 < group size: all still pending), then the hook raises
    assert len(F._wgrad_queue.pending) == 3
    for m in blocks:
        m.zero_grad(set_to_none=True)
    host.calls.clear()
    dy, dvf = _run_chain(blocks, order, y, vf, ml, g)
    assert not F._wgrad_queue.pending and not F._wgrad_queue._passes
    grouped = [int(c.split("[")[1][:-1]) for c in host.calls if c.startswith("ff_xattn_wgrad_grouped")]
    assert sorted(grouped) == [1, 4]                                 # this pass's five blocks only: one full group and its own trailing one
    want, dy_w, dvf_w = _oracle_chain(blocks, order, y, vf, ml, g)
    assert _close(dy.double().numpy(), dy_w, 2e-4) and _close(dvf.double().numpy(), dvf_w, 2e-4)
    for m, w in zip(blocks, want):
        for k, p in m.named_parameters():
            got = p.grad.double().numpy()
            assert np.isfinite(got).all() and _close(got.reshape(-1), np.asarray(w[k]).reshape(-1), 2e-4), k


def test_a_block_used_twice_in_one_pass_sums_both_contributions(clean_patches):
    """Module reuse (or an activation-checkpoint recompute): the same parameters receive two gradient contributions in one backward pass.
    The second use must not defer - autograd adds the two tensors the moment the second is returned - and the first must be complete by
    then."""
    from detgen import det
    from flamingo_mini_amd import functional as F
    host, blocks, (dim, dv, heads, dh, nv, ffm) = _host_blocks(3, "qr")
    b, L = 2, 6
    y, vf, g = det((b, L, dim), "qr-y"), det((b, 1, nv, dv), "qr-vf"), det((b, L, dim), "qr-g")
    ml = np.zeros((b, L), np.int64); ml[:, 0] = 1
    order = [0, 1, 2, 1, 0]
    dy, dvf = _run_chain(blocks, order, y, vf, ml, g)
    assert not F._wgrad_queue.pending and not F._wgrad_queue._passes
    assert host.calls.count("ff_xattn_block_bwd_kv_data") == 3 and host.calls.count("ff_xattn_block_bwd_kv") == 2
    want, dy_w, dvf_w = _oracle_chain(blocks, order, y, vf, ml, g)
    assert _close(dy.double().numpy(), dy_w, 2e-4) and _close(dvf.double().numpy(), dvf_w, 2e-4)
    for m, w in zip(blocks, want):
        for k, p in m.named_parameters():
            got = p.grad.double().numpy()
            assert np.isfinite(got).all() and _close(got.reshape(-1), np.asarray(w[k]).reshape(-1), 2e-4), k


# ---- four gloo ranks: bucket arrival order, accumulation under no_sync(), widened reduction, sharded state round trip ------------
def _rank_inputs(rank, dims, micro=0):
    from detgen import det
    dim, dv, heads, dh, nv, ffm = dims
    b, L = 2, 6
    y, vf, g = det((b, L, dim), f"r{rank}m{micro}-y"), det((b, 1, nv, dv), f"r{rank}m{micro}-vf"), det((b, L, dim), f"r{rank}m{micro}-g")
    ml = np.zeros((b, L), np.int64); ml[:, 0] = 1
    return y, vf, ml, g


def _chain_loss(blocks, order, y, vf, ml, g, dtype=torch.float32):
    from flamingo_mini_amd import functional as F
    yt, vft, mlt = torch.from_numpy(y).to(dtype), torch.from_numpy(vf).to(dtype), torch.from_numpy(ml)
    kvs = F.kv_project(vft, [m.attn.to_kv.weight for m in blocks])
    h = yt
    for i in order:
        h, _ = blocks[i](h, vft, mlt, hoisted_kv=kvs[i])
    return (h * torch.from_numpy(g).to(dtype)).sum()


ORDER_A, ORDER_B = [0, 1, 2], [2, 0, 1]


def _four_rank_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from flamingo_mini_amd.data_parallel import GradientAllReducer, ShardedAdamW
    from test_data_parallel import HP, _torch_adamw
    host, blocks, dims = _host_blocks(3, "w4")
    model = torch.nn.ModuleList(blocks)
    # (1) all-reduce with the exchange widened to float64 (reduce_dtype): mean gradients, identical on every rank
    reducer = GradientAllReducer(model, reduce_dtype=torch.float64)
    _chain_loss(blocks, ORDER_A, *_rank_inputs(rank, dims)).backward()
    reducer.finish()
    out = {"g." + k: p.grad.double().numpy().copy() for k, p in model.named_parameters()}
    reducer.close()
    model.zero_grad(set_to_none=True)
    # (2) sharded AdamW: step 1 visits the blocks in one order, step 2 in another (the buckets arrive in another order) and is made of two
    # micro-batches, the first under no_sync()
    opt = ShardedAdamW(model, update_fn=_torch_adamw, **HP)
    _chain_loss(blocks, ORDER_A, *_rank_inputs(rank, dims)).backward()
    opt.step()
    saved = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state_dict().items() if k != "buckets"}
    saved["buckets"] = {n: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()} for n, b in opt.state_dict()["buckets"].items()}
    params_after_1 = {k: p.detach().clone() for k, p in model.named_parameters()}

    def step2(o, mods):
        o.zero_grad()
        with o.no_sync():
            (_chain_loss(mods, ORDER_B, *_rank_inputs(rank, dims, 1)) / 2).backward()
        (_chain_loss(mods, ORDER_B, *_rank_inputs(rank, dims, 2)) / 2).backward()
        o.step()

    step2(opt, blocks)
    out.update({"p." + k: p.detach().double().numpy().copy() for k, p in model.named_parameters()})
    refused = 0
    opt.zero_grad()
    _chain_loss(blocks, ORDER_A, *_rank_inputs(rank, dims)).backward()
    try:                                                   # a second backward in the same step without no_sync(): refused, not silently applied
        _chain_loss(blocks, ORDER_A, *_rank_inputs(rank, dims)).backward()
    except RuntimeError as e:
        refused = int("no_sync" in str(e))
    opt.close()
    # (3) resume: fresh modules holding the parameters after step 1 + this rank's saved shards -> the same step 2
    _, blocks2, _ = _host_blocks(3, "w4")
    model2 = torch.nn.ModuleList(blocks2)
    model2.load_state_dict(params_after_1)
    opt2 = ShardedAdamW(model2, update_fn=_torch_adamw, **HP)
    opt2.load_state_dict(saved)
    step2(opt2, blocks2)
    out.update({"r." + k: p.detach().double().numpy().copy() for k, p in model2.named_parameters()})
    opt2.close()
    np.savez(os.path.join(out_dir, f"w4_{rank}.npz"), refused=refused, **out)
    dist.barrier()
    dist.destroy_process_group()


def test_four_gloo_ranks_reduce_dtype_bucket_order_accumulation_and_resume(tmp_path, clean_patches):
    world = 4
    port = _free_port()
    mp.start_processes(_four_rank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    res = [np.load(tmp_path / f"w4_{r}.npz") for r in range(world)]
    # single-process float64 reference on the oracle-backed entry points
    import host_lib
    import oracle_backend
    from detgen import xattn_params
    from flamingo_mini_amd import GatedCrossAttentionBlock
    from test_data_parallel import HP
    host_lib.uninstall()
    oracle_backend.install()
    dims = (32, 24, 2, 8, 4, 2)
    dim, dv, heads, dh, nv, ffm = dims
    blocks = []
    for i in range(3):
        m = GatedCrossAttentionBlock(dim=dim, dim_visual=dv, dim_head=dh, heads=heads, ff_mult=ffm, n_visual=nv).double()
        m.load_state_dict({k: torch.from_numpy(v).double() for k, v in xattn_params(dim, dv, heads, dh, ffm, tag=f"w4{i}").items()})
        blocks.append(m)
    model = torch.nn.ModuleList(blocks)

    def ref_loss(order, micro):
        from flamingo_mini_amd import functional as F
        tot = 0.0
        for r in range(world):
            y, vf, ml, g = _rank_inputs(r, dims, micro)
            h = torch.from_numpy(y).double()
            vft, mlt = torch.from_numpy(vf).double(), torch.from_numpy(ml)
            for i in order:
                h, _ = blocks[i](h, vft, mlt)
            tot = tot + (h * torch.from_numpy(g).double()).sum()
        return tot / world

    ref_loss(ORDER_A, 0).backward()
    for k, p in model.named_parameters():
        for r in range(world):
            assert np.array_equal(res[r]["g." + k], res[0]["g." + k]), k
        assert _close(res[0]["g." + k].reshape(-1), p.grad.numpy().reshape(-1), 2e-4), k
    opt = torch.optim.AdamW(model.parameters(), **HP)
    opt.step()
    model.zero_grad(set_to_none=True)
    ((ref_loss(ORDER_B, 1) + ref_loss(ORDER_B, 2)) / 2).backward()
    opt.step()
    for k, p in model.named_parameters():
        want = p.detach().numpy().reshape(-1)
        for r in range(world):
            assert np.array_equal(res[r]["p." + k], res[0]["p." + k]), k          # every rank holds the same parameters after the all-gather
            assert np.array_equal(res[r]["r." + k], res[r]["p." + k]), k          # resumed from the saved shards: bit-identical step 2
        assert _close(res[0]["p." + k].reshape(-1), want, 5e-4), k
    assert all(int(r["refused"]) == 1 for r in res)


def test_gradient_arena_hands_out_consecutive_slices_and_buckets_merge(clean_patches):
    """functional.GradArena + graphs._merge_arena_buckets (the piecewise step's one-collective-per-segment plumbing), on plain host tensors:
    a sizing arena only adds up what is asked for; a backed arena hands out consecutive slices and refuses what does not fit or is of another
    dtype (the caller then allocates as before); `_flat_grads` goes through the installed arena; the recorded buckets that are slices of the
    arena merge into ONE bucket spanning its used part with the owners' offsets rebased, a foreign bucket (an un-fused parameter's own
    gradient) keeps its place."""
    from flamingo_mini_amd import functional as F
    from flamingo_mini_amd.graphs import _merge_arena_buckets
    params_a = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7))]
    params_b = [torch.nn.Parameter(torch.zeros(2000))]
    sizing = F.GradArena()
    prev = F.set_grad_arena(sizing)
    try:
        fa, ga = F._flat_grads(params_a)
        fb, gb = F._flat_grads(params_b)
    finally:
        assert F.set_grad_arena(prev) is sizing
    assert sizing.buf is None and sizing.need == {(torch.float32, torch.device("cpu")): fa.numel() + fb.numel()}
    assert fa.numel() % 1024 == 0 and fb.numel() % 1024 == 0 and fa.data_ptr() != fb.data_ptr()
    arena = F.GradArena()
    arena.buf = torch.zeros(fa.numel() + fb.numel(), dtype=torch.float32)
    F.set_grad_arena(arena)
    try:
        fa2, ga2 = F._flat_grads(params_a)
        fb2, gb2 = F._flat_grads(params_b)
        fc2, _ = F._flat_grads(params_b)                                      # does not fit any more: its own allocation
        fd2, _ = F._flat_grads([torch.nn.Parameter(torch.zeros(4, dtype=torch.float64))])     # another dtype: its own allocation
    finally:
        F.set_grad_arena(None)
    base = arena.buf.data_ptr()
    assert fa2.data_ptr() == base and fb2.data_ptr() == base + fa2.numel() * 4 and arena.used == fa2.numel() + fb2.numel()
    assert not (base <= fc2.data_ptr() < base + arena.buf.numel() * 4) and fd2.dtype == torch.float64
    assert [g.shape for g in ga2] == [p.shape for p in params_a] and ga2[1].data_ptr() == base + F._flat_offsets(params_a)[0][1] * 4
    loose = torch.zeros(11)
    buckets = [(fa2, F._owners(params_a)), (loose, []), (fb2, F._owners(params_b))]
    merged = _merge_arena_buckets(buckets, arena)
    assert len(merged) == 2 and merged[1][0] is loose
    flat, owners = merged[0]
    assert flat.data_ptr() == base and flat.numel() == arena.used
    assert [(p is q, off, n) for (p, off, n), q in zip(owners, params_a + params_b)] == \
        [(True, 0, 15), (True, F._flat_offsets(params_a)[0][1], 7), (True, fa2.numel(), 2000)]
    for p, off, n in owners:                                                  # the rebased offsets address the very slices autograd adopted
        g = {id(q): t for q, t in zip(params_a + params_b, ga2 + gb2)}[id(p)]
        assert flat[off:off + n].data_ptr() == g.data_ptr()
    assert _merge_arena_buckets(buckets[:2], arena) == buckets[:2]            # fewer than two arena buckets: nothing to merge
    assert _merge_arena_buckets(buckets, None) is buckets
