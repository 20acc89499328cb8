"""GraphedTrainStep: a captured HIP graph (forward + backward + FusedAdamW) replayed n times must equal n eager steps."""
import copy

import pytest
import torch

from util import dev, rel, rnd

pytestmark = pytest.mark.gpu


class _Toy(torch.nn.Module):
    """Resampler + gated cross-attention block -> scalar loss: every kernel family of the library in one tiny step."""

    def __init__(self):
        super().__init__()
        from flamingo_mini_amd import GatedCrossAttentionBlock, PerceiverResampler
        self.resampler = PerceiverResampler(dim=64, depth=1, heads=2, dim_head=32, num_latents=8, num_time_embeds=2)
        self.block = GatedCrossAttentionBlock(dim=64, dim_visual=64, dim_head=32, heads=2, n_visual=8)
        with torch.no_grad():
            self.block.alpha_attn.fill_(0.5)
            self.block.alpha_ffw.fill_(0.5)

    def forward(self, x_f, y, media_locations):
        vf = self.resampler(x_f).unsqueeze(1)
        out, _ = self.block(y, vf, media_locations)
        return out.float().pow(2).mean()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_graph_replay_equals_eager_steps(dtype):
    from flamingo_mini_amd import FusedAdamW, GraphedTrainStep
    torch.manual_seed(0)
    eager = _Toy().cuda().to(dtype)
    graphed = copy.deepcopy(eager)
    ml = torch.zeros(2, 16, dtype=torch.int64, device="cuda")
    ml[:, 0] = 1
    batches = [dict(x_f=dev(rnd((2, 1, 24, 64), 10 + i), dtype), y=dev(rnd((2, 16, 64), 20 + i), dtype), media_locations=ml) for i in range(6)]
    opt_e = FusedAdamW(eager.parameters(), lr=1e-2)
    losses_e = []
    for b in batches:
        eager.zero_grad(set_to_none=True)
        loss = eager(**b)
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss))
    # the graphed twin: 2 eager warm-up steps on batches 0, 1 (inside the constructor), then replays on batches 2..5
    opt_g = FusedAdamW(graphed.parameters(), lr=1e-2, capturable=True)

    step = GraphedTrainStep(graphed, opt_g, batches[0], warmup=1, loss_fn=lambda out: out)
    # constructor ran exactly one eager step on batch 0; continue with replays on batches 1..5
    losses_g = [None]
    for b in batches[1:]:
        losses_g.append(float(step(b)))
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    for le, lg in zip(losses_e[1:], losses_g[1:]):
        assert abs(le - lg) <= tol * max(1.0, abs(le)), (losses_e, losses_g)
    for (n, pe), (_, pg) in zip(eager.named_parameters(), graphed.named_parameters()):
        assert rel(pg, pe) < (1e-4 if dtype == torch.float32 else 3e-2), n
    assert {float(s["step"]) for s in opt_g.state_dict()["state"].values()} == {6.0}


def test_graph_replay_under_autocast_equals_eager_steps_under_autocast():
    """fp32 parameters, torch.autocast(bf16) around the forward (the reference's mixed-precision recipe): GraphedTrainStep(autocast=...) captures the
    casts of the parameters with the step; replays equal eager autocast steps."""
    from flamingo_mini_amd import FusedAdamW, GraphedTrainStep
    torch.manual_seed(0)
    eager = _Toy().cuda()
    graphed = copy.deepcopy(eager)
    ml = torch.zeros(2, 16, dtype=torch.int64, device="cuda")
    ml[:, 0] = 1
    batches = [dict(x_f=dev(rnd((2, 1, 24, 64), 10 + i)), y=dev(rnd((2, 16, 64), 20 + i)), media_locations=ml) for i in range(5)]
    opt_e = FusedAdamW(eager.parameters(), lr=1e-2)
    losses_e = []
    for b in batches:
        eager.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = eager(**b)
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss))
    opt_g = FusedAdamW(graphed.parameters(), lr=1e-2, capturable=True)
    with GraphedTrainStep(graphed, opt_g, batches[0], warmup=1, loss_fn=lambda out: out, autocast=torch.bfloat16) as step:
        losses_g = [None] + [float(step(b)) for b in batches[1:]]
    for le, lg in zip(losses_e[1:], losses_g[1:]):
        assert abs(le - lg) <= 3e-2 * max(1.0, abs(le)), (losses_e, losses_g)
    for (n, pe), (_, pg) in zip(eager.named_parameters(), graphed.named_parameters()):
        assert pg.dtype == torch.float32 and rel(pg, pe) < 3e-2, n


class _ToyHoisted(torch.nn.Module):
    """Resampler + two gated blocks on K / V projected up front (the training layout of FlamingoModel): exercises the deferred,
    grouped weight gradients and every kind of gradient bucket the reducer sees."""

    def __init__(self):
        super().__init__()
        from flamingo_mini_amd import GatedCrossAttentionBlock, PerceiverResampler
        self.resampler = PerceiverResampler(dim=64, depth=1, heads=2, dim_head=32, num_latents=8, num_time_embeds=2)
        self.blocks = torch.nn.ModuleList(GatedCrossAttentionBlock(dim=64, dim_visual=64, dim_head=32, heads=2, n_visual=8) for _ in range(2))
        self.loose = torch.nn.Parameter(torch.ones(64))          # an un-fused trainable parameter (the token embedding's role)
        with torch.no_grad():
            for b in self.blocks:
                b.alpha_attn.fill_(0.5)
                b.alpha_ffw.fill_(-0.25)

    def forward(self, x_f, y, media_locations):
        from flamingo_mini_amd import functional as F
        vf = self.resampler(x_f).unsqueeze(1)
        kvs = F.kv_project(vf, [b.attn.to_kv.weight for b in self.blocks])
        h = y * self.loose
        for b, kv in zip(self.blocks, kvs):
            h, _ = b(h, vf, media_locations, hoisted_kv=kv)
        return h.float().pow(2).mean()


@pytest.fixture
def one_rank_rccl():
    import os
    import socket
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_rccl_reducer_on_one_rank_eager_and_captured(one_rank_rccl, dtype):
    """The NCCL (= RCCL) branch of GradientAllReducer on a 1-rank group: side-stream all-reduces (ReduceOp.AVG) issued from inside
    backward, joined by finish(); then the same step captured into a HIP graph WITH its collectives and replayed.  On one rank the
    average is the identity, so all three variants must produce the same losses and parameters."""
    from flamingo_mini_amd import FusedAdamW, GraphedTrainStep
    from flamingo_mini_amd.data_parallel import GradientAllReducer
    torch.manual_seed(0)
    plain = _ToyHoisted().cuda().to(dtype)
    reduced, graphed = copy.deepcopy(plain), copy.deepcopy(plain)
    ml = torch.zeros(2, 16, dtype=torch.int64, device="cuda"); ml[:, 0] = 1
    batches = [dict(x_f=dev(rnd((2, 1, 24, 64), 30 + i), dtype), y=dev(rnd((2, 16, 64), 40 + i), dtype), media_locations=ml) for i in range(5)]

    def eager_steps(model, reducer):
        opt = FusedAdamW(model.parameters(), lr=1e-2)
        out = []
        for b in batches:
            model.zero_grad(set_to_none=True)
            loss = model(**b)
            loss.backward()
            if reducer is not None:
                reducer.finish()
            opt.step()
            out.append(float(loss))
        return out

    base = eager_steps(plain, None)
    r1 = GradientAllReducer(reduced, force_collectives=True)
    assert r1.cuda and r1.active
    calls = []
    orig = r1._reduce_async
    r1._reduce_async = lambda flat, owners: (calls.append(flat.numel()), orig(flat, owners))[1]
    with_rccl = eager_steps(reduced, r1)
    r1.close()
    # per step: the resampler layer by layer (the reducer switches ITS model to ff_resampler_layer_*: final norm, one bucket per layer, latents +
    # time embedding), the to_kv bucket, 2 blocks, the loose parameter
    assert len(calls) == 5 * (reduced.resampler.depth + 2 + 4)
    assert not reduced.resampler.layerwise                      # ... and close() has switched it back
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    for a, b in zip(base, with_rccl):
        assert abs(a - b) <= tol * max(1.0, abs(a)), (base, with_rccl)
    for (n, pa), (_, pb) in zip(plain.named_parameters(), reduced.named_parameters()):
        assert rel(pb, pa) < (1e-5 if dtype == torch.float32 else 2e-2), n

    r2 = GradientAllReducer(graphed, force_collectives=True)
    opt_g = FusedAdamW(graphed.parameters(), lr=1e-2, capturable=True)
    step = GraphedTrainStep(graphed, opt_g, batches[0], warmup=1, loss_fn=lambda out: out, reducer=r2)     # one eager step on batch 0, then the capture
    replayed = [None] + [float(step(b)) for b in batches[1:]]
    r2.close()
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    for a, b in zip(base[1:], replayed[1:]):
        assert abs(a - b) <= tol * max(1.0, abs(a)), (base, replayed)
    for (n, pa), (_, pb) in zip(plain.named_parameters(), graphed.named_parameters()):
        assert rel(pb, pa) < (1e-4 if dtype == torch.float32 else 3e-2), n


@pytest.mark.parametrize("master", [None, torch.float32], ids=["bf16-state", "fp32-master"])
def test_sharded_adamw_on_one_rccl_rank_equals_fused_adamw(one_rank_rccl, master):
    """ShardedAdamW's device path (reduce_scatter_tensor -> ff_adamw_step_mixed on the rank's slice -> all_gather_into_tensor, per bucket, on
    the side stream) on a 1-rank RCCL group must reproduce GradientAllReducer-free FusedAdamW training step for step."""
    from flamingo_mini_amd import FusedAdamW
    from flamingo_mini_amd.data_parallel import ShardedAdamW
    torch.manual_seed(0)
    dtype = torch.bfloat16
    plain = _ToyHoisted().cuda().to(dtype)
    sharded = copy.deepcopy(plain)
    ml = torch.zeros(2, 16, dtype=torch.int64, device="cuda"); ml[:, 0] = 1
    batches = [dict(x_f=dev(rnd((2, 1, 24, 64), 50 + i), dtype), y=dev(rnd((2, 16, 64), 60 + i), dtype), media_locations=ml) for i in range(4)]
    hp = dict(lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    opt_p = FusedAdamW(plain.parameters(), master_dtype=master, **hp)
    opt_s = ShardedAdamW(sharded, master_dtype=master, force_collectives=True, **hp)
    assert opt_s.cuda and opt_s.collectives
    try:
        for b in batches:
            plain.zero_grad(set_to_none=True)
            plain(**b).backward()            # (its buckets reach opt_s's process-wide callback too and must be ignored: not its model)
            opt_p.step()
            opt_s.zero_grad()
            sharded(**b).backward()
            opt_s.finish_step()
        torch.cuda.synchronize()
    finally:
        opt_s.close()
    # the resampler layer by layer (final norm, one bucket per layer, latents + time embedding), the hoisted to_kv weights, two blocks
    assert len(opt_s.buckets) == sharded.resampler.depth + 2 + 3, list(opt_s.buckets)
    for (n, pa), (_, pb) in zip(plain.named_parameters(), sharded.named_parameters()):
        assert rel(pb, pa) < 1e-2, n


def test_sharded_adamw_refuses_autocast_over_fp32_parameters(one_rank_rccl):
    """Under torch.autocast the fused modules run on casts of their fp32 parameters, so no gradient bucket of THIS model ever reaches
    ShardedAdamW - whose update is per bucket.  It must say so instead of silently not updating those parameters."""
    from flamingo_mini_amd.data_parallel import ShardedAdamW
    torch.manual_seed(0)

    class ToyLoose(_Toy):                   # ... plus one un-fused trainable parameter (the token embedding's role)
        def __init__(self):
            super().__init__()
            self.loose = torch.nn.Parameter(torch.ones(64))

        def forward(self, x_f, y, media_locations):
            return super().forward(x_f, y * self.loose, media_locations)

    model = ToyLoose().cuda()
    ml = torch.zeros(2, 16, dtype=torch.int64, device="cuda"); ml[:, 0] = 1
    batch = dict(x_f=dev(rnd((2, 1, 24, 64), 70)), y=dev(rnd((2, 16, 64), 71)), media_locations=ml)
    opt = ShardedAdamW(model, lr=1e-2, force_collectives=True)
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(**batch)
        loss.backward()
        with pytest.raises(RuntimeError, match="did not arrive as a gradient bucket"):
            opt.finish_step()
        torch.cuda.synchronize()
    finally:
        opt.close()


@pytest.mark.parametrize("mode,expect", [("drop", 0), ("nograd", 0), ("keep", 3)])
def test_capture_after_an_eager_forward_of_the_same_model(mode, expect):
    """tools/capture_after_eager.py in a child process (the failure this guards against was a segmentation fault inside the runtime):
    an eager forward with gradients enabled before the capture is fine once its outputs are gone (the hooks keep no conditioning),
    a forward under no_grad always is, and outputs that are still alive make GraphedTrainStep refuse (exit code 3) instead of crashing."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "capture_after_eager.py"), mode], capture_output=True, text=True, timeout=600)
    assert res.returncode == expect, (res.returncode, res.stdout[-600:], res.stderr[-600:])
    assert ("refused:" in res.stdout) == (expect == 3)
    if expect == 0:
        assert "captured; replays:" in res.stdout


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_piecewise_graphs_with_eager_collectives_equal_eager_and_full_capture(one_rank_rccl, dtype):
    """graphs.PiecewiseGraphedTrainStep on the full drop-in model (h64 fixture geometry: fused kernels, hoisted K / V, deferred weight
    gradients) with the gradient exchange going through RCCL on a 1-rank group: forward | one backward sub-graph per gated layer |
    resampler backward | optimizer, all-reduces issued eagerly between the replays.  Five arms from the same initial state - eager
    launches, the whole step captured with its collectives, the piecewise replay with host-paced and with stream-ordered collectives, and
    with the optimizer split into per-segment sub-graphs that run beside the backward of the layers below - must produce the same losses
    and parameters."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_model_plumbing import H64, build_h64
    from flamingo_mini_amd import FusedAdamW, GraphedTrainStep
    from flamingo_mini_amd.data_parallel import GradientAllReducer
    from flamingo_mini_amd.graphs import PiecewiseGraphedTrainStep
    base, z, batch = build_h64(dtype, "cuda")
    arms = {"eager": base, "full": copy.deepcopy(base), "piecewise": copy.deepcopy(base), "piecewise_stream_paced": copy.deepcopy(base),
            "piecewise_overlapped_optimizer": copy.deepcopy(base)}
    n_steps, losses, finals = 4, {}, {}
    for name, model in arms.items():
        params = [p for p in model.parameters_trainable()]
        opt = FusedAdamW(params, capturable=name != "eager", **H64["adamw"])
        reducer = GradientAllReducer(model, force_collectives=True)
        assert reducer.active and all(h.xattn_block.wgrad_group == 4 for h in model.flamingo.get_modified_layers())
        if name == "eager":
            out = []
            for _ in range(n_steps):
                model.zero_grad(set_to_none=True)
                loss = model(**batch).loss
                loss.backward()
                reducer.finish()
                opt.step()
                out.append(float(loss))
        else:
            cls = GraphedTrainStep if name == "full" else PiecewiseGraphedTrainStep
            kw = {} if name == "full" else {"segment_layers": 1, "pace": "stream" if name.endswith("stream_paced") else "host",
                                            "overlap_optimizer": name.endswith("overlapped_optimizer")}
            step = cls(model, opt, batch, warmup=1, reducer=reducer, **kw)        # (the constructor's warm-up is training step 1)
            if name.endswith("overlapped_optimizer"):      # one optimizer sub-graph per segment in which a gradient became final
                assert sum(g is not None for g in step._opt_pieces) >= 3 and step._opt_graph is None
            if name.startswith("piecewise"):
                assert len(step.graphs) == 1 + 3 and sum(len(b) for b in step.segment_buckets) >= 4      # forward + 3 backward segments; blocks, to_kv, resampler, embedding
            out = [None] + [float(step()) for _ in range(n_steps - 1)]
        torch.cuda.synchronize()
        reducer.close()
        losses[name] = out
        finals[name] = {k: p.detach().float().clone() for k, p in model.named_parameters() if p.requires_grad}
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    for name in ("full", "piecewise", "piecewise_stream_paced", "piecewise_overlapped_optimizer"):
        for a, b in zip(losses["eager"][1:], losses[name][1:]):
            assert abs(a - b) <= tol * max(1.0, abs(a)), (name, losses)
        for k, v in finals["eager"].items():
            assert rel(finals[name][k], v) < (1e-4 if dtype == torch.float32 else 3e-2), (name, k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_piecewise_overlapped_optimizer_without_a_reducer_equals_eager_steps(dtype):
    """The single-GPU form of PiecewiseGraphedTrainStep(overlap_optimizer=True): no reducer, the per-segment AdamW sub-graphs run on a side
    stream beside the backward segments below them.  Same losses and parameters as eager steps; the tied token embedding (gradient from the
    head in the top segment AND from the lookup in the bottom one) must be updated once, after the later of the two."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_model_plumbing import H64, build_h64
    from flamingo_mini_amd import FusedAdamW
    from flamingo_mini_amd.graphs import PiecewiseGraphedTrainStep
    base, z, batch = build_h64(dtype, "cuda")
    arms = {"eager": base, "overlapped": copy.deepcopy(base)}
    n_steps, losses, finals = 5, {}, {}
    for name, model in arms.items():
        opt = FusedAdamW([p for p in model.parameters_trainable()], capturable=name != "eager", **H64["adamw"])
        if name == "eager":
            out = []
            for _ in range(n_steps):
                model.zero_grad(set_to_none=True)
                loss = model(**batch).loss
                loss.backward()
                opt.step()
                out.append(float(loss))
        else:
            step = PiecewiseGraphedTrainStep(model, opt, batch, warmup=1, segment_layers=1, overlap_optimizer=True)
            pieces = [g for g in step._opt_pieces if g is not None]
            assert len(pieces) >= 3
            out = [None] + [float(step()) for _ in range(n_steps - 1)]
            torch.cuda.current_stream().synchronize()       # (stream semantics: the calling stream has waited for the side stream)
        torch.cuda.synchronize()
        losses[name] = out
        finals[name] = {k: p.detach().float().clone() for k, p in model.named_parameters() if p.requires_grad}
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    for a, b in zip(losses["eager"][1:], losses["overlapped"][1:]):
        assert abs(a - b) <= tol * max(1.0, abs(a)), losses
    for k, v in finals["eager"].items():
        assert rel(finals["overlapped"][k], v) < (1e-4 if dtype == torch.float32 else 3e-2), k
