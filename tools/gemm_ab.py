"""A/B of GEMM variants on the model's shapes with COLD operands (24 rotating copies: 375 MB > the 256 MB Infinity Cache), wall time per
call over 100 back-to-back launches (HIP events on the stream; includes a split-K reduce launch where the plan has one).
Development-build switches are read once per process, so every variant is its own process:
    FLAMINGO_FUSION_LIB=debug FF_GEMM_PF=4 python tools/gemm_ab.py [--shapes ff] """
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flamingo_mini_amd import functional as F

SHAPES = [  # name, M, N, K, a_layout, b_layout, epilogue
    ("ff1.fwd", 1024, 5120, 1280, 0, 0, "act"), ("ff2.fwd", 1024, 1280, 5120, 0, 0, "res"),
    ("ff2.dgrad", 1024, 5120, 1280, 0, 1, "actbwd"), ("ff1.dgrad", 1024, 1280, 5120, 0, 1, ""),
    ("ff2.wgrad", 1280, 5120, 1024, 1, 1, ""), ("ff1.wgrad", 5120, 1280, 1024, 1, 1, ""),
    ("out.fwd", 1024, 1280, 512, 0, 0, "res"), ("q.dgrad", 1024, 1280, 512, 0, 1, ""),
    ("rs.ff1.fwd", 2048, 4096, 1024, 0, 0, "act"), ("rs.kv.fwd", 10272, 512, 1024, 0, 0, ""), ("rs.dkv.dgrad", 10272, 1024, 512, 0, 1, ""),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--stages", type=int, default=0)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    tag = args.tag or " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("FF_GEMM"))
    nc = 24
    for name, M, N, K, al, bl, epi in SHAPES:
        if args.shapes and not any(s in name for s in args.shapes.split(",")):
            continue
        As = [torch.randn((M, K) if al == 0 else (K, M), device="cuda", dtype=torch.bfloat16) for _ in range(nc)]
        Bs = [torch.randn((N, K) if bl == 0 else (K, N), device="cuda", dtype=torch.bfloat16) * 0.05 for _ in range(nc)]
        R = torch.randn((M, N), device="cuda", dtype=torch.bfloat16)
        gate = torch.tensor([0.5], device="cuda", dtype=torch.bfloat16)
        kw = dict(a_layout=al, b_layout=bl, tile=args.tile, stages=args.stages)
        if epi == "act":
            kw.update(act="gelu", want_aux_out=True)
        elif epi == "res":
            kw.update(residual=R, gate=gate)
        elif epi == "actbwd":
            kw.update(act_bwd="gelu", aux_in=R, gate=gate)

        def run(i):
            F.gemm(As[i % nc], Bs[i % nc], **kw)
        for i in range(6):
            run(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for rep in range(5):
            s.record()
            for i in range(100):
                run(i)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 100 * 1e3)
        ts.sort()
        print(f"[{tag}] {name:12s} {M}x{N}x{K} {al}{bl}: median {ts[2]:6.1f} us  best {ts[0]:6.1f} us  {2.0 * M * N * K / ts[2] / 1e6:5.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
