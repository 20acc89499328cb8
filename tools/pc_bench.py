"""Wall time (HIP events over many launches, operands rotated through 24 copies so they come from HBM / MALL, not L2) of the two
gated-block FFW products with and without the one-tile-per-CU producer / consumer kernel:
    FF_GEMM_PC=0 python tools/pc_bench.py ; FF_GEMM_PC=1 python tools/pc_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flamingo_mini_amd import ffi, functional as F

lib = ffi.lib()
SHAPES = [("ff1.fwd", 1024, 5120, 1280), ("ff2.fwd", 1024, 1280, 5120)]
for name, M, N, K in SHAPES:
    As = [torch.randn((M, K), device="cuda", dtype=torch.bfloat16) for _ in range(24)]
    Bs = [torch.randn((N, K), device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(24)]
    R = torch.randn((M, N), device="cuda", dtype=torch.bfloat16)
    gate = torch.tensor([0.5], device="cuda", dtype=torch.bfloat16)
    for tile, stages in [(0, 0)] + ([(128160, 4)] if os.environ.get("FF_GEMM_PC", "1") == "1" else []):
        def run(i):
            if name == "ff1.fwd":
                F.gemm(As[i % 24], Bs[i % 24], act="gelu", want_aux_out=True, tile=tile, stages=stages)
            else:
                F.gemm(As[i % 24], Bs[i % 24], residual=R, gate=gate, tile=tile, stages=stages)
        for i in range(5):
            run(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(3):
            s.record()
            for i in range(100):
                run(i)
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 100 * 1e3)
        print(f"PC={os.environ.get('FF_GEMM_PC', '1')} {name} {M}x{N}x{K} tile={tile or 'auto'} stages={stages or 'auto'}: {best:6.1f} us/call (incl. split-K reduce), {2.0 * M * N * K / best / 1e6:5.0f} TFLOP/s")
