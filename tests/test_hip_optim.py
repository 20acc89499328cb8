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
