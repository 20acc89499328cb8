#!/usr/bin/env python
"""One GEMM shape of the library in isolation without host launch overhead: a HIP graph of NB back-to-back launches, each on its own B
(weight) buffer - NB x bytes beyond the 256 MB on-die cache, so weights arrive cold as in the model - and either one shared A buffer (warm
activations, the model's case) or rotating ones; replayed and timed with events.  us per launch include the gap between graph nodes.
    python tools/gemm_graph_bench.py M N K [a_layout b_layout [tile [stages]]]        env: COLD_A=1 rotates A as well"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flamingo_mini_amd import functional as F

args = [int(a) for a in sys.argv[1:]]
M, N, K = args[:3]
al, bl = (args[3], args[4]) if len(args) >= 5 else (0, 0)
tile = args[5] if len(args) >= 6 else 0
stages = args[6] if len(args) >= 7 else 0
dt = torch.bfloat16
nb = max(4, min(48, int(400e6 / (N * K * 2))))
uniq = int(os.environ.get("UNIQUE", "0"))     # round 6: UNIQUE=1 = the SAME weight buffer in every one of the nb launches of the graph (warm: L2 / Infinity
                                              # Cache), UNIQUE=4 = four buffers in rotation, ...: separates operand latency from request rate (0 = all different: cold)
Bs = [torch.randn((N, K) if bl == 0 else (K, N), device="cuda", dtype=dt) * 0.05 for _ in range(uniq if uniq > 0 else nb)]
if uniq > 0:
    Bs = [Bs[i % uniq] for i in range(nb)]
na = nb if os.environ.get("COLD_A", "0") == "1" else 1
As = [torch.randn((M, K) if al == 0 else (K, M), device="cuda", dtype=dt) for _ in range(na)]


EPI = os.environ.get("EPI", "")      # "": plain; "act": gelu + aux_out (FFW up-projection); "act_bwd": gelu' from aux_in, gated (its data gradient)
# COLD_H=1 (round 6): one pre-activation buffer per launch, as in the model (each block's H was written a forward pass ago: it comes from HBM)
Hs = [torch.randn(M, N, device="cuda", dtype=dt) for _ in range(nb if os.environ.get("COLD_H", "0") == "1" else 1)] if EPI.startswith("act_bwd") else None
gate = torch.tensor([0.5], device="cuda", dtype=dt)
R = torch.randn(M, N, device="cuda", dtype=dt) if EPI == "res" else None


def run():
    for i, B in enumerate(Bs):
        kw = dict(a_layout=al, b_layout=bl, tile=tile, stages=stages)
        if EPI in ("act", "act_sqrelu"):
            F.gemm(As[i % na], B, act="gelu" if EPI == "act" else "sqrelu", want_aux_out=True, **kw)
        elif EPI in ("act_bwd", "act_bwd_sqrelu"):
            F.gemm(As[i % na], B, act_bwd="gelu" if EPI == "act_bwd" else "sqrelu", aux_in=Hs[i % len(Hs)], gate=gate, **kw)
        elif EPI == "res":
            F.gemm(As[i % na], B, residual=R, gate=gate, **kw)
        else:
            F.gemm(As[i % na], B, **kw)


run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    run()
    with torch.cuda.graph(g, stream=side):
        run()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 10
for _ in range(reps):
    g.replay()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (reps * nb)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FF_") or k in ("EPI", "COLD_H", "COLD_A", "UNIQUE"))
print(f"{M}x{N}x{K} a{al} b{bl} tile {tile or 'auto'} [{tag}]: {us:7.2f} us   {2.0 * M * N * K / us / 1e6:6.0f} TFLOP/s   ({nb} B buffers, {na} A)", flush=True)
