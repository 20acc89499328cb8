"""FusedAdamW (ff_adamw_step) against the numpy AdamW rule and against torch.optim.AdamW on the same device."""
import numpy as np
import pytest
import torch

from oracle import flamingo_oracle as O
from util import as64, dev, rel, rnd

pytestmark = pytest.mark.gpu
SHAPES = [(1,), (1280,), (513, 7), (5120, 1280), (64, 1024), (3,)]      # includes the 1-element alphas and ragged tails


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_fused_adamw_matches_oracle_and_torch(dtype):
    from flamingo_mini_amd import FusedAdamW
    hp = dict(lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    ours = [torch.nn.Parameter(dev(rnd(s, 10 + i), dtype)) for i, s in enumerate(SHAPES)]
    theirs = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    ref = [(as64(p), np.zeros(p.shape), np.zeros(p.shape)) for p in ours]
    opt_a, opt_b = FusedAdamW(ours, **hp), torch.optim.AdamW(theirs, fused=True, **hp)
    for step in range(1, 5):
        for i, (a, b) in enumerate(zip(ours, theirs)):
            g = dev(rnd(a.shape, 100 * step + i, 0.5), dtype)
            a.grad, b.grad = g, g.clone()
            p64, m64, v64 = ref[i]
            ref[i] = O.adamw_step(p64, as64(g), m64, v64, step, lr=hp["lr"], beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.05)
            if dtype == torch.bfloat16:     # the kernel stores p, m, v in bf16 after every step: mirror that rounding in the oracle
                ref[i] = tuple(as64(torch.as_tensor(t).to(torch.bfloat16)) for t in ref[i])
        opt_a.step()
        opt_b.step()
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    for i, (a, b) in enumerate(zip(ours, theirs)):
        assert rel(a, ref[i][0]) < tol, SHAPES[i]
        assert rel(opt_a.state[a]["exp_avg"], ref[i][1]) < tol and rel(opt_a.state[a]["exp_avg_sq"], ref[i][2]) < tol
        assert rel(a, b) < tol, SHAPES[i]                                   # and torch's own fused AdamW
    sd = opt_a.state_dict()                                                 # same state layout as torch.optim.AdamW
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0


def test_fused_adamw_capturable_and_graph_replay():
    """capturable=True keeps the step count on the device: (a) eager steps equal the host-step mode (device powf vs host powf: ulps),
    (b) a captured HIP graph replayed n times equals n eager steps, (c) state_dict() reports the device step count."""
    from flamingo_mini_amd import FusedAdamW
    shapes = [(129,), (64, 40), (8191,)]

    def make():
        ps = [torch.nn.Parameter(dev(rnd(s, 10 + i))) for i, s in enumerate(shapes)]
        gs = [dev(rnd(s, 20 + i, 0.1)) for i, s in enumerate(shapes)]
        return ps, gs

    p_host, grads = make()
    p_cap, _ = make()
    p_graph, _ = make()
    o_host = FusedAdamW(p_host, lr=1e-2, weight_decay=0.1)
    o_cap = FusedAdamW(p_cap, lr=1e-2, weight_decay=0.1, capturable=True)
    o_graph = FusedAdamW(p_graph, lr=1e-2, weight_decay=0.1, capturable=True)
    for ps in (p_host, p_cap, p_graph):
        for p, g in zip(ps, grads):
            p.grad = g.clone()
    for _ in range(4):
        o_host.step(); o_cap.step()
    for a, b in zip(p_host, p_cap):
        assert rel(a, b) < 1e-6
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        o_graph.step()                                # step 1 eagerly (allocates the state), steps 2..4 from the graph
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o_graph.step()                                # capture does not execute
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(p_host, p_graph):
        assert rel(a, b) < 1e-6
    steps = {float(s["step"]) for s in o_graph.state_dict()["state"].values()}
    assert steps == {4.0}
    assert "_step_dev" not in o_graph.state_dict()["param_groups"][0]


def test_fused_adamw_partial_steps_cover_a_step():
    """step(only=ids, advance=...) (capturable mode): three partial calls that together cover the parameters once - the first one advancing the
    device step counter - equal one whole step, eagerly and as three captured sub-graphs replayed in order (what
    graphs.PiecewiseGraphedTrainStep(overlap_optimizer=True) launches per backward segment); host-step mode refuses partial calls."""
    from flamingo_mini_amd import FusedAdamW
    shapes = [(129,), (64, 40), (8191,), (1,), (513, 7)]

    def make():
        ps = [torch.nn.Parameter(dev(rnd(s, 10 + i))) for i, s in enumerate(shapes)]
        for i, p in enumerate(ps):
            p.grad = dev(rnd(shapes[i], 20 + i, 0.1))
        return ps

    whole, parts, graphed = make(), make(), make()
    o_w = FusedAdamW(whole, lr=1e-2, weight_decay=0.1, capturable=True)
    o_p = FusedAdamW(parts, lr=1e-2, weight_decay=0.1, capturable=True)
    o_g = FusedAdamW(graphed, lr=1e-2, weight_decay=0.1, capturable=True)
    split = lambda ps: [frozenset(id(p) for p in ps[:2]), frozenset(id(p) for p in ps[2:3]), frozenset(id(p) for p in ps[3:])]
    for _ in range(3):
        o_w.step()
        for k, ids in enumerate(split(parts)):
            o_p.step(only=ids, advance=k == 0)
    for a, b in zip(whole, parts):
        assert torch.equal(a, b)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        o_g.step()                                    # step 1 eagerly (allocates the state and the device counters)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    pieces = []
    for k, ids in enumerate(split(graphed)):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            o_g.step(only=ids, advance=k == 0)
        pieces.append(g)
    for _ in range(2):
        for g in pieces:
            g.replay()
    torch.cuda.synchronize()
    for a, b in zip(whole, graphed):
        assert rel(a, b) < 1e-6
    assert {float(st["step"]) for st in o_g.state_dict()["state"].values()} == {3.0}
    with pytest.raises(ValueError):
        FusedAdamW(make(), lr=1e-2).step(only=frozenset(), advance=True)


def test_fused_adamw_fp32_master_weights_keep_small_updates():
    """master_dtype=fp32 (what the reference's `--fp16` autocast training keeps, training/train.sh:24): the fp32 master copy follows the
    float64 AdamW rule and the bf16 parameter is its rounding; with bf16-only storage the same small steps (lr 1e-4 on weights ~1) are
    below the weight's bf16 resolution and vanish."""
    from flamingo_mini_amd import FusedAdamW
    hp = dict(lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    shapes = [(1280,), (257, 9), (1,)]
    base = [dev(np.sign(rnd(s, 70 + i)) * (1.0 + 0.25 * np.abs(rnd(s, 80 + i))), torch.bfloat16) for i, s in enumerate(shapes)]
    with_master = [torch.nn.Parameter(t.clone()) for t in base]
    bf16_only = [torch.nn.Parameter(t.clone()) for t in base]
    opt_m = FusedAdamW(with_master, master_dtype=torch.float32, **hp)
    opt_b = FusedAdamW(bf16_only, **hp)
    ref = [(as64(t), np.zeros(t.shape), np.zeros(t.shape)) for t in base]
    n_steps = 40
    for step in range(1, n_steps + 1):
        for i, (a, b) in enumerate(zip(with_master, bf16_only)):
            g = dev(np.abs(rnd(a.shape, 1000 * step + i, 0.5)) + 0.1, torch.bfloat16)          # same sign every step: the updates add up
            a.grad, b.grad = g, g.clone()
            ref[i] = O.adamw_step(*ref[i][:1], as64(g), *ref[i][1:], step, lr=hp["lr"], beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0)
        opt_m.step()
        opt_b.step()
    for i, (a, b) in enumerate(zip(with_master, bf16_only)):
        st = opt_m.state[a]
        assert st["master"].dtype == torch.float32 and st["exp_avg"].dtype == torch.float32 and st["exp_avg_sq"].dtype == torch.float32
        assert rel(st["master"], ref[i][0]) < 2e-6 and rel(st["exp_avg"], ref[i][1]) < 1e-5 and rel(st["exp_avg_sq"], ref[i][2]) < 5e-5
        assert torch.equal(a.detach(), st["master"].to(torch.bfloat16))                        # the parameter IS the rounded master copy
        moved_master = float((st["master"].double() - base[i].double()).abs().mean())
        assert moved_master > 0.5 * n_steps * hp["lr"]                                         # ~ n_steps * lr, as AdamW with steady gradients does
        assert float((b.detach().double() - base[i].double()).abs().mean()) < 0.1 * moved_master   # bf16 storage: (almost) every step rounded away
    fp32_state = [torch.nn.Parameter(t.clone()) for t in base]                                # moments in fp32, no master copy
    opt_s = FusedAdamW(fp32_state, state_dtype=torch.float32, **hp)
    for p in fp32_state:
        p.grad = torch.ones_like(p)
    opt_s.step()
    assert all(opt_s.state[p]["exp_avg"].dtype == torch.float32 and "master" not in opt_s.state[p] for p in fp32_state)


def test_learning_rate_schedule_survives_graph_replay():
    """capturable=True keeps lr in a device scalar: changing group["lr"] between replays of a captured step must act like eager steps."""
    from flamingo_mini_amd import FusedAdamW
    shapes = [(130,), (33, 40)]
    lrs = [1e-2, 5e-3, 2e-2, 1e-3]

    def make():
        ps = [torch.nn.Parameter(dev(rnd(s, 10 + i))) for i, s in enumerate(shapes)]
        for i, p in enumerate(ps):
            p.grad = dev(rnd(p.shape, 20 + i, 0.1))
        return ps

    p_e, p_g = make(), make()
    o_e, o_g = FusedAdamW(p_e, lr=lrs[0]), FusedAdamW(p_g, lr=lrs[0], capturable=True)
    for lr in lrs:
        o_e.param_groups[0]["lr"] = lr
        o_e.step()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        o_g.step()                                    # step 1 (lr = lrs[0]) eagerly
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o_g.step()
    for lr in lrs[1:]:
        o_g.param_groups[0]["lr"] = lr
        o_g.sync_device_hyperparams()
        graph.replay()
    torch.cuda.synchronize()
    for a, b in zip(p_e, p_g):
        assert rel(b, a) < 1e-6


@pytest.mark.parametrize("mode", ["master", "state"])
def test_fused_adamw_mixed_precision_state_dict_round_trip(mode):
    """save -> load -> step in mixed precision: torch's Optimizer.load_state_dict casts state tensors to the parameter's dtype;
    the fp32 moments / master copies of bf16 parameters must come back as fp32 and the resumed run must equal the uninterrupted one."""
    from flamingo_mini_amd import FusedAdamW
    kw = dict(master_dtype=torch.float32) if mode == "master" else dict(state_dtype=torch.float32)
    shapes = [(257,), (64, 40), (1,)]

    def make():
        return [torch.nn.Parameter(dev(rnd(s, 10 + i), torch.bfloat16)) for i, s in enumerate(shapes)]

    grads = [[dev(rnd(s, 100 * st + i, 0.3), torch.bfloat16) for i, s in enumerate(shapes)] for st in range(4)]
    straight, first = make(), make()
    o_straight, o_first = FusedAdamW(straight, lr=1e-2, weight_decay=0.1, **kw), FusedAdamW(first, lr=1e-2, weight_decay=0.1, **kw)
    for st in range(4):
        for p, g in zip(straight, grads[st]):
            p.grad = g.clone()
        o_straight.step()
    for st in range(2):
        for p, g in zip(first, grads[st]):
            p.grad = g.clone()
        o_first.step()
    import copy
    import io
    buf = io.BytesIO()
    torch.save(dict(opt=o_first.state_dict(), params=[p.detach().clone() for p in first]), buf)
    buf.seek(0)
    ck = torch.load(buf)
    resumed = [torch.nn.Parameter(t.clone()) for t in ck["params"]]
    o_res = FusedAdamW(resumed, lr=1e-2, weight_decay=0.1, **kw)
    o_res.load_state_dict(copy.deepcopy(ck["opt"]))
    for p in resumed:
        st = o_res.state[p]
        assert st["exp_avg"].dtype == torch.float32 and st["exp_avg_sq"].dtype == torch.float32
        assert ("master" in st) == (mode == "master") and (mode != "master" or st["master"].dtype == torch.float32)
    for st in range(2, 4):
        for p, g in zip(resumed, grads[st]):
            p.grad = g.clone()
        o_res.step()
    for a, b in zip(straight, resumed):
        assert torch.equal(a, b)
    for a, b in zip(straight, resumed):
        for k in ("exp_avg", "exp_avg_sq") + (("master",) if mode == "master" else ()):
            assert torch.equal(o_straight.state[a][k], o_res.state[b][k]), k
    assert float(o_res.state_dict()["state"][0]["step"]) == 4.0
