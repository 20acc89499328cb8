"""Phase timeline of ln_bwd_fused_kernel (debug build): 0 entry | 1 arguments | 2 rows done | 3 cross-wave sums | 4 partials written."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from flamingo_mini_amd import ffi, functional as F
ffi.LIB_PATH = os.path.join(ROOT, "tools", "_dbg", "libflamingo_fusion_timeline.so")
lib = ffi.lib()
rd = C.CDLL(ffi.LIB_PATH).ff_debug_ln_timeline_read
rd.argtypes = [C.c_void_p, C.c_int]
for rows, cols in ((1024, 1280), (2048, 1024)):
    x = torch.randn(rows, cols, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn_like(x); res = torch.randn_like(x)
    g = torch.randn(cols, device="cuda", dtype=torch.bfloat16); b = torch.zeros_like(g)
    y, mean, rstd = F.layernorm_fwd(x, g, b)
    for it in range(3):
        F.layernorm_bwd(dy, x, g, mean, rstd, dx_residual=res)
        torch.cuda.synchronize()
        nb = 256 if rows == 1024 else 512
        buf = np.zeros((nb, 8), dtype=np.uint64); assert rd(buf.ctypes.data, nb) == 0
        t = buf[:, :5].astype(np.int64); t0 = t[:, 0].min()
        print(rows, cols, f"span {(t[:, 4].max() - t0) / 100:.2f} us start-spread {(t[:, 0].max() - t0) / 100:.2f} | " +
              " ".join(f"{i}->{i + 1}: {np.median(t[:, i + 1] - t[:, i]) / 100:5.2f}" for i in range(4)))
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        F.layernorm_bwd(dy, x, g, mean, rstd, dx_residual=res)
    e.record(); torch.cuda.synchronize()
    print("   wall per call (fused + final kernel, incl. allocations)", s.elapsed_time(e) / 20 * 1e3, "us")
