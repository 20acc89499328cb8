"""Per-workgroup phase timeline of the bf16 GEMM kernel (debug build: -DFF_GEMM_TIMELINE).
    hipcc ... -DFF_GEMM_TIMELINE -> tools/_dbg/libflamingo_fusion_timeline.so ;  python tools/gemm_timeline.py
Phases (100 MHz clock, 10 ns ticks): 0 entry, 1 addressing done, 2 first operand tile landed, 3 K loop done,
4 fp32 tile parked in LDS, 5 epilogue stores issued.
"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flamingo_mini_amd import ffi, functional as F
ffi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_dbg", "libflamingo_fusion_timeline.so")
from gemm_bench import gpu_us

lib = ffi.lib()
rd = C.CDLL(ffi.LIB_PATH).ff_debug_timeline_read
rd.argtypes = [C.c_void_p, C.c_int]

CASES = [("tiny 4x8 K=64", 512, 1024, 64, 0, 0, 128, 1, 2), ("4x8 K=1024", 512, 1024, 1024, 0, 0, 128, 1, 2),
         ("xa.ff1.fwd", 1024, 5120, 1280, 0, 0, 128, 1, 2), ("xa.ff1.fwd ns3", 1024, 5120, 1280, 0, 0, 128, 1, 3),
         ("xa.ff1.fwd 6412", 1024, 5120, 1280, 0, 0, 6412, 1, 2), ("xa.ff1.fwd 6412 ns3", 1024, 5120, 1280, 0, 0, 6412, 1, 3),
         ("xa.ff1.fwd 6412 ns4", 1024, 5120, 1280, 0, 0, 6412, 1, 4),
         ("xa.ff2.wgrad", 1280, 5120, 1024, 1, 1, 128, 1, 2), ("xa.ff2.wgrad ns3", 1280, 5120, 1024, 1, 1, 128, 1, 3),
         ("xa.ff2.fwd split5", 1024, 1280, 5120, 0, 0, 128, 5, 2), ("xa.ff2.fwd split5 ns3", 1024, 1280, 5120, 0, 0, 128, 5, 3),
         ("xa.q.fwd 64 s2", 1024, 512, 1280, 0, 0, 64, 2, 2), ("xa.q.fwd 64 s2 ns3", 1024, 512, 1280, 0, 0, 64, 2, 3),
         ("xa.q.fwd 64 s2 ns4", 1024, 512, 1280, 0, 0, 64, 2, 4), ("xa.q.fwd 64 s1 ns4", 1024, 512, 1280, 0, 0, 64, 1, 4),
         ("xa.kv.fwd 64", 2048, 1024, 1024, 0, 0, 64, 1, 2), ("xa.kv.fwd 64 ns4", 2048, 1024, 1024, 0, 0, 64, 1, 4),
         ("xa.kv.fwd 128 ns3", 2048, 1024, 1024, 0, 0, 128, 1, 3),
         ("xa.q.wgrad 64 s2", 512, 1280, 1024, 1, 1, 64, 2, 2), ("xa.q.wgrad 64 s2 ns4", 512, 1280, 1024, 1, 1, 64, 2, 4)]
for name, M, N, K, al, bl, tile, split, stages in CASES:
    A = torch.randn((M, K) if al == 0 else (K, M), device="cuda", dtype=torch.bfloat16)
    B = torch.randn((N, K) if bl == 0 else (K, N), device="cuda", dtype=torch.bfloat16)
    run = lambda: F.gemm(A, B, a_layout=al, b_layout=bl, split_k=split, tile=tile, stages=stages)
    us, _, _ = gpu_us(run, 10)
    torch.cuda.synchronize(); run(); torch.cuda.synchronize()
    te = tile if tile != 6412 else 64
    nb = -(-M // te) * -(-N // (128 if tile != 64 else 64)) * split
    nb = min(nb, 16384)
    buf = np.zeros((nb, 8), dtype=np.uint64)
    assert rd(buf.ctypes.data, nb) == 0
    t = buf[:, :6].astype(np.int64)
    last = 5 if split == 1 else 3
    t0 = t[:, 0].min()
    ph = lambda a, b: np.median(t[:, b] - t[:, a]) / 100.0
    print(f"{name:20s} wgs {nb:5d} event {us:6.1f} us | span {((t[:, last].max() - t0) / 100.0):6.2f} | start spread p50 {np.median(t[:, 0] - t0) / 100:5.2f} max {(t[:, 0].max() - t0) / 100:5.2f}"
          f" | addr {ph(0, 1):5.2f} first-tile {ph(1, 2):5.2f} kloop {ph(2, 3):5.2f}" + (f" park {ph(3, 4):5.2f} epi {ph(4, 5):5.2f}" if split == 1 else "")
          + f" | wg life p50 {np.median(t[:, last] - t[:, 0]) / 100:5.2f} max {(t[:, last] - t[:, 0]).max() / 100:5.2f}")
