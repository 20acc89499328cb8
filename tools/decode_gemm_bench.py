#!/usr/bin/env python
"""Decode-shaped products (M <= 32 rows) in isolation, cold weights: a HIP graph of NB back-to-back launches, each on a different weight
buffer (NB x weight bytes > the 256 MB on-die cache), replayed and timed with events - no host launch overhead in the figure.
    python tools/decode_gemm_bench.py [tile ...]        default: 3216 (weight-streaming kernel) and 3264 (32 x 64 tiles)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flamingo_mini_amd import functional as F

tiles = [int(a) for a in sys.argv[1:]] or [3216, 3264]
M = int(os.environ.get("ROWS", "32"))
shapes = [(5120, 1280, "ffw up"), (1280, 5120, "ffw down"), (512, 1280, "to_q"), (1280, 512, "to_out")]
dt = torch.bfloat16
print(f"M = {M} rows; us per launch incl. the inter-kernel gap of a graph replay; GB/s = weight bytes / that")
for N, K, what in shapes:
    nb = max(4, min(48, int(400e6 / (N * K * 2))))
    Bs = [torch.randn(N, K, device="cuda", dtype=dt) * 0.05 for _ in range(nb)]
    A = torch.randn(M, K, device="cuda", dtype=dt)
    for tile in tiles:
        def run():
            for B in Bs:
                F.gemm(A, B, tile=tile)
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
        reps = 20
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * nb)
        print(f"{what:9s} N={N:5d} K={K:5d} tile {tile:6d}: {us:6.2f} us   {N * K * 2 / us * 1e-3:7.1f} GB/s   ({nb} weight buffers)", flush=True)
