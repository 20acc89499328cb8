"""Launch-time model of the 128x128 GEMM kernel: t(tiles, K) at split_k = 1 (fixed cost vs per-K-step cost vs grid size).
    python tools/gemm_fit.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flamingo_mini_amd import ffi, functional as F
from gemm_bench import gpu_us

lib = ffi.lib()
for al, bl in ((0, 0), (1, 1), (0, 1)):
    print(f"layouts {al}{bl}: rows = tile grid (MxN tiles of 128), columns = K; cell = us")
    Ks = (64, 128, 256, 512, 1024, 2048, 4096)
    print("   grid  tiles | " + " ".join(f"{k:7d}" for k in Ks))
    for tm, tn in ((4, 8), (8, 8), (8, 16), (8, 32), (10, 40), (16, 32), (16, 48), (32, 32), (32, 64)):
        M, N = tm * 128, tn * 128
        cells = []
        for K in Ks:
            A = torch.randn((M, K) if al == 0 else (K, M), device="cuda", dtype=torch.bfloat16)
            B = torch.randn((N, K) if bl == 0 else (K, N), device="cuda", dtype=torch.bfloat16)
            us, _, _ = gpu_us(lambda: F.gemm(A, B, a_layout=al, b_layout=bl, split_k=1, tile=128, stages=2), 12)
            cells.append(f"{us:7.1f}")
        print(f"  {tm:2d}x{tn:2d} {tm * tn:6d} | " + " ".join(cells))
