"""Per-workgroup phase timeline of the fused cross-attention kernels (debug build: tools/build_timeline.sh, -DFF_XA_TIMELINE).
forward : 0 entry | 1 key ranges + K/V DMA issued | 2 LayerNorm statistics done | 3 projection done | 4 Q parked + stored | 5 attention | 6 stores
backward: 0 entry | 1 prefetch issued | 2 projection done | 3 dO parked | 4 dQ loop | 5 dQ stored | 6 dK / dV done
    tools/build_timeline.sh && python tools/xattn_timeline.py
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
from flamingo_mini_amd import ffi, functional as F
ffi.LIB_PATH = os.path.join(ROOT, "tools", "_dbg", "libflamingo_fusion_timeline.so")
from flamingo_mini_amd import GatedCrossAttentionBlock
from detgen import xattn_params

lib = ffi.lib()
rd = C.CDLL(ffi.LIB_PATH).ff_debug_xa_timeline_read
rd.argtypes = [C.c_void_p, C.c_int]
b, L, dim, dv, H = 32, 32, 1280, 1024, 8
blk = GatedCrossAttentionBlock(dim=dim, dim_visual=dv)
blk.load_state_dict({k: torch.from_numpy(v) for k, v in xattn_params(dim, dv, 8, 64, 4, tag="tl").items()})
blk = blk.to(torch.bfloat16).cuda()
y = torch.randn(b, L, dim, device="cuda", dtype=torch.bfloat16, requires_grad=True)
vf = torch.randn(b, 1, 64, dv, device="cuda", dtype=torch.bfloat16)
ml = torch.zeros(b, L, dtype=torch.long, device="cuda"); ml[:, 0] = 1
nb = b * H


def show(tag):
    torch.cuda.synchronize()
    buf = np.zeros((nb, 8), dtype=np.uint64)
    assert rd(buf.ctypes.data, nb) == 0
    t = buf[:, :7].astype(np.int64)
    t0 = t[:, 0].min()
    ph = " ".join(f"{i}->{i + 1}: {np.median(t[:, i + 1] - t[:, i]) / 100:5.2f} (max {(t[:, i + 1] - t[:, i]).max() / 100:5.2f})" for i in range(6))
    print(f"{tag}: span {(t[:, 6].max() - t0) / 100:6.2f} us, start spread {(t[:, 0].max() - t0) / 100:5.2f}, wg life p50 {np.median(t[:, 6] - t[:, 0]) / 100:5.2f} max {(t[:, 6] - t[:, 0]).max() / 100:5.2f} | {ph}")


for it in range(3):
    kv = F.kv_project(vf, [blk.attn.to_kv.weight])[0]
    out, _ = blk(y, vf, ml, hoisted_kv=kv)
    show(f"fwd[{it}]")
    out.backward(torch.ones_like(out))
    show(f"bwd[{it}]")
