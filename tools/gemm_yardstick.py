#!/usr/bin/env python
"""hipBLASLt beside the hand-written GEMMs, shape by shape ("the honest yard-stick to report beside").

Reads the per-shape table bench.py writes (`--gemm-table`: every GEMM shape one training step of the fusion path launches) and times each row
twice with the SAME method - a HIP graph of back-to-back launches on rotating weight buffers (more bytes than the 256 MiB on-die cache, so the
weights arrive cold as in the model), replayed and timed with events, plain products without epilogues:
  * this library's kernel with the plan the model uses (ff_gemm through flamingo_mini_amd.functional.gemm; grouped rows as nz launches),
  * torch.matmul on the same operands in the same layouts (hipBLASLt's default heuristic through PyTorch).
Writes the table with two extra columns to stdout:
    python tools/gemm_yardstick.py profiles/r04_gemm_table.txt > profiles/r04_gemm_yardstick.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from flamingo_mini_amd import functional as F


def graph_us(run, reps=8):
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()
        with torch.cuda.graph(g, stream=side):
            n = run()
    torch.cuda.synchronize()
    for _ in range(2):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


def main():
    path = sys.argv[1]
    dt = torch.bfloat16
    rows = [l.split() for l in open(path).read().strip().splitlines()[1:]]
    print("M N K nz aL bL tile splitK launches/step in-model_us ff_isolated_us hipblaslt_us ff/hipblaslt   (isolated columns: plain products on cold weights, no epilogue; rows with nz>1 time the nz products as nz separate launches on BOTH sides and report their sum, while in-model_us is the model's ONE grouped launch)")
    for r in rows:
        M, N, K, nz, al, bl, tile, sk = (int(v) for v in r[:8])
        per_step, in_model_us = r[8], r[9]
        nb = max(3, min(32, int(400e6 / (N * K * 2 * nz))))
        Bs = [[torch.randn((N, K) if bl == 0 else (K, N), device="cuda", dtype=dt) * 0.05 for _ in range(nz)] for _ in range(nb)]
        As = [torch.randn((M, K) if al == 0 else (K, M), device="cuda", dtype=dt) for _ in range(nz)]

        def ours():
            for group in Bs:
                for A, B in zip(As, group):
                    F.gemm(A, B, a_layout=al, b_layout=bl)
            return nb

        def blas():
            for group in Bs:
                for A, B in zip(As, group):
                    torch.matmul(A if al == 0 else A.t(), B.t() if bl == 0 else B)
            return nb

        try:
            t_ff, t_bl = graph_us(ours), graph_us(blas)      # per GROUP of nz problems (what one launch of the model's grouped kernel covers)
            print(M, N, K, nz, al, bl, tile, sk, per_step, in_model_us, f"{t_ff:.1f}", f"{t_bl:.1f}", f"{t_ff / t_bl:.2f}", flush=True)
        except Exception as e:
            print(M, N, K, nz, al, bl, tile, sk, per_step, in_model_us, "error", repr(e)[:80], flush=True)
        del Bs, As
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
