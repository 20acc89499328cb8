#!/usr/bin/env python
"""Element-wise comparison of the resident-operand fused cross-attention kernels with the previous ones on the same inputs:
    FLAMINGO_FUSION_LIB=debug FF_XATTN_RES=0 python tools/res_compare.py run /tmp/old.pt
    python tools/res_compare.py run /tmp/new.pt ; python tools/res_compare.py diff /tmp/old.pt /tmp/new.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np
import torch

if sys.argv[1] == "run":
    from detgen import det, xattn_params
    from test_hip_modules import build_block
    from util import dev
    out = {}
    for (b, L, nv, dim) in [(2, 32, 40, 768), (3, 20, 64, 1280)]:
        dtype = torch.bfloat16
        dv, heads, dh, ffm = 256, 8, 64, 2
        m = build_block(xattn_params(dim, dv, heads, dh, ffm, tag=f"res{L}{nv}"), dim, dv, heads, dh, nv, ffm, "gelu", dtype)
        ml = np.zeros((b, L), np.int64); ml[0, 0] = 1; ml[1, min(3, L - 1)] = 1
        if b > 2:
            ml[2, [1, L - 2]] = 1
        yd = dev(det((b, L, dim), "res-y"), dtype).requires_grad_(True)
        vfd = dev(det((b, 1, nv, dv), "res-vf"), dtype).requires_grad_(True)
        dyd = dev(det((b, L, dim), "res-dy"), dtype)
        o, kv = m(yd, vfd, torch.as_tensor(ml).cuda(), output_kv=True)
        o.backward(dyd)
        tag = f"{L}-{nv}."
        out.update({tag + "out": o.detach().float().cpu(), tag + "dy": yd.grad.float().cpu(), tag + "dvf": vfd.grad.float().cpu()})
        out.update({tag + "g." + k: p.grad.float().cpu() for k, p in m.named_parameters()})
    torch.save(out, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k] - b[k]).double()
        n = float(d.norm() / max(float(a[k].double().norm()), 1e-30))
        nz = int((d != 0).sum())
        print(f"{k:32s} rel diff {n:.3e}  differing elements {nz}/{d.numel()}  max |diff| {float(d.abs().max()):.3e}")
