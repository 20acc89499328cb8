"""r6 session 14: the paired 64 x 160 launch (tile code 64160) against the planned 128 x 160 tile - parity with torch first, then timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from flamingo_mini_amd import functional as F
torch.manual_seed(0)
dt = torch.bfloat16
for (M, N, K, bl, split) in [(1024, 5120, 1280, 0, 0), (1024, 5120, 1280, 1, 0), (1024, 1280, 5120, 0, 4), (1024, 1280, 5120, 1, 4), (256, 320, 192, 0, 0), (384, 160, 128, 1, 0)]:
    A = torch.randn(M, K, device="cuda", dtype=dt)
    B = (torch.randn((N, K) if bl == 0 else (K, N), device="cuda", dtype=dt) * 0.05)
    ref = (A.float() @ (B.float().t() if bl == 0 else B.float()))
    for tile in (128160, 64160):
        C = F.gemm(A, B, b_layout=bl, tile=tile, split_k=split)
        err = ((C.float() - ref).norm() / ref.norm()).item()
        print(f"{M}x{N}x{K} b{bl} split {split} tile {tile}: rel err {err:.2e}", flush=True)
        assert err < 5e-3
    C1 = F.gemm(A, B, b_layout=bl, tile=128160, split_k=split); C2 = F.gemm(A, B, b_layout=bl, tile=64160, split_k=split)
    print("   bitwise equal to the 128 x 160 plan:", bool(torch.equal(C1, C2)))
