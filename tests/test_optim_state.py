"""FusedAdamW checkpoint plumbing that needs no GPU: load_state_dict must keep the fp32 moments / master copies of bf16 parameters
in fp32 (torch's Optimizer.load_state_dict casts floating-point state to the parameter's dtype).  The save -> load -> step
round trip on the device is tests/test_hip_optim.py::test_fused_adamw_mixed_precision_state_dict_round_trip."""
import copy

import torch


def _filled(params, **kw):
    from flamingo_mini_amd import FusedAdamW
    opt = FusedAdamW(params, **kw)
    g = torch.Generator().manual_seed(0)
    for p in params:                       # the state FusedAdamW._buckets creates on the device, written by hand
        st = opt.state[p]
        sdt = torch.float32 if (p.dtype == torch.bfloat16 and opt.state_dtype is not None) else p.dtype
        st["step"] = torch.tensor(3.0)
        st["exp_avg"] = torch.randn(p.shape, generator=g).to(sdt)
        st["exp_avg_sq"] = torch.rand(p.shape, generator=g).to(sdt)
        if p.dtype == torch.bfloat16 and opt.master_dtype is not None:
            st["master"] = p.detach().float() + 1e-4          # differs from the bf16 rounding: a cast through bf16 would lose it
    return opt


def test_load_state_dict_keeps_fp32_state_of_bf16_parameters():
    from flamingo_mini_amd import FusedAdamW
    params = [torch.nn.Parameter(torch.randn(5, 3).bfloat16()), torch.nn.Parameter(torch.randn(4))]
    for kw in (dict(master_dtype=torch.float32), dict(state_dtype=torch.float32), dict()):
        src = _filled(params, **kw)
        sd = copy.deepcopy(src.state_dict())
        fresh = [torch.nn.Parameter(p.detach().clone()) for p in params]
        dst = FusedAdamW(fresh, **kw)
        dst.load_state_dict(sd)
        for p, q in zip(params, fresh):
            assert set(src.state[p]) == set(dst.state[q])
            for k, a in src.state[p].items():
                b = dst.state[q][k]
                assert a.dtype == b.dtype and torch.equal(a, b), (kw, k, a.dtype, b.dtype)
    # a mixed-precision checkpoint loaded by an optimizer built WITHOUT master copies: moments keep the checkpoint's precision, masters go
    sd = copy.deepcopy(_filled(params, master_dtype=torch.float32).state_dict())
    fresh = [torch.nn.Parameter(p.detach().clone()) for p in params]
    plain = FusedAdamW(fresh)
    plain.load_state_dict(sd)
    assert "master" not in plain.state[fresh[0]] and plain.state[fresh[0]]["exp_avg"].dtype == torch.float32
