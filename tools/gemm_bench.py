"""Per-shape GPU time of the fusion GEMM (HIP events bracketing each launch, via ff_gemm_profile_*), swept over block
tile / ring depth / split-K, next to torch.matmul (hipBLASLt, wall time over many launches) as a yardstick.
    python tools/gemm_bench.py [--iters 20] [--only ff] [--sweep]
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flamingo_mini_amd import ffi, functional as F

# (name, M, N, K, a_layout, b_layout) at config B: xattn block M = b*L = 1024, resampler M = 2048 / 10272
SHAPES = [
    ("xa.ff1.fwd", 1024, 5120, 1280, 0, 0), ("xa.ff2.fwd", 1024, 1280, 5120, 0, 0), ("xa.q.fwd", 1024, 512, 1280, 0, 0),
    ("xa.kv.fwd", 2048, 1024, 1024, 0, 0), ("xa.out.fwd", 1024, 1280, 512, 0, 0),
    ("xa.ff2.dgrad", 1024, 5120, 1280, 0, 1), ("xa.ff1.dgrad", 1024, 1280, 5120, 0, 1), ("xa.out.dgrad", 1024, 512, 1280, 0, 1),
    ("xa.ff2.wgrad", 1280, 5120, 1024, 1, 1), ("xa.ff1.wgrad", 5120, 1280, 1024, 1, 1),
    ("xa.kv.wgrad", 1024, 1024, 2048, 1, 1), ("xa.q.wgrad", 512, 1280, 1024, 1, 1), ("xa.out.wgrad", 1280, 512, 1024, 1, 1),
    ("rs.kv.fwd", 10272, 512, 1024, 0, 0), ("rs.ff1.fwd", 2048, 4096, 1024, 0, 0), ("rs.ff2.fwd", 2048, 1024, 4096, 0, 0),
    ("rs.kv.wgrad", 512, 1024, 10272, 1, 1), ("rs.dkv.dgrad", 10272, 1024, 512, 0, 1),
    ("square4k", 4096, 4096, 4096, 0, 0),
]


def gpu_us(fn, iters):
    lib = ffi.lib()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.ff_gemm_profile_enable(iters + 8)
    for _ in range(iters):
        fn()
    recs = (ffi.GemmProfileRecord * (iters + 8))()
    n = lib.ff_gemm_profile_read(recs, iters + 8)
    lib.ff_gemm_profile_enable(0)
    ts = sorted(recs[i].ms for i in range(n))
    return ts[len(ts) // 2] * 1e3, recs[0].tile, recs[0].split_k


def wall_us(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--cold", action="store_true", help="rotate through 24 operand copies so weights come from HBM, not cache")
    args = ap.parse_args()
    lib = ffi.lib()
    configs = [(0, 0, 0)]
    if args.sweep:
        configs += [(128, 2, 1), (6412, 2, 1), (64, 2, 1), (128, 2, 2), (128, 2, 4), (6412, 2, 2), (64, 2, 2)]
    print(f"{'shape':13s} {'M':>6s} {'N':>6s} {'K':>6s} L | " + " ".join(f"t{t}/s{s}/k{k}".rjust(12) for t, s, k in configs) + " | blaslt(wall)")
    for name, M, N, K, al, bl in SHAPES:
        if args.only and args.only not in name:
            continue
        ncopy = 24 if args.cold else 1
        As = [torch.randn((M, K) if al == 0 else (K, M), device="cuda", dtype=torch.bfloat16) for _ in range(ncopy)]
        Bs = [torch.randn((N, K) if bl == 0 else (K, N), device="cuda", dtype=torch.bfloat16) for _ in range(ncopy)]
        fl = 2.0 * M * N * K
        cells = []
        for tile, stages, split in configs:
            it = [0]

            def run():
                i = it[0] % ncopy
                it[0] += 1
                F.gemm(As[i], Bs[i], a_layout=al, b_layout=bl, split_k=split, tile=tile, stages=stages)
            us, t_used, s_used = gpu_us(run, args.iters)
            cells.append(f"{us:5.1f}/{fl / us / 1e6:4.0f}TF".rjust(12))
        a2 = As[0] if al == 0 else As[0].t()
        b2 = Bs[0].t() if bl == 0 else Bs[0]
        ref = wall_us(lambda: torch.matmul(a2, b2), args.iters)
        print(f"{name:13s} {M:6d} {N:6d} {K:6d} {al}{bl} | " + " ".join(cells) + f" | {ref:5.1f}/{fl / ref / 1e6:4.0f}TF")


if __name__ == "__main__":
    main()
