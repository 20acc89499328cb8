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
