"""Per-workgroup phase timeline of the RESIDENT fused cross-attention kernels with the in-launch exchange (round 5; debug build:
tools/build_timeline.sh, -DFF_XA_TIMELINE; 100 MHz clock, 16 slots per workgroup).
forward : 0 entry | 1 rows / K / V / first weight tiles landed | 2 LayerNorm done | 3 q projection done | 4 attention done, O stored (+ drained)
          | 5 every head of the sample has arrived | 6 O of all heads staged in LDS (+ first Wo tiles) | 7 out-projection done, tile parked | 8 epilogue stored
          | 9 phase 3 (LayerNorm behind the product: statistics exchanged through the second counter bank, normalised rows stored)
backward: 0 entry | 1 operands landed | 2 dO projection done | 3 dQ done (+ published) | 4 dK / dV stored | 5 every head has arrived
          | 6 dQ of all heads staged | 7 d LN(y) product done, tile parked | 8 stored
    tools/build_timeline.sh && python tools/xattn_res_timeline.py
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


def show(tag, last):
    torch.cuda.synchronize()
    buf = np.zeros((nb, 16), dtype=np.uint64)
    assert rd(buf.ctypes.data, nb) == 0
    t = buf[:, :last + 1].astype(np.int64)
    t0 = t[:, 0].min()
    ph = " ".join(f"{i}->{i + 1}: {np.median(t[:, i + 1] - t[:, i]) / 100:5.2f} (max {(t[:, i + 1] - t[:, i]).max() / 100:5.2f})" for i in range(last))
    print(f"{tag}: span {(t[:, last].max() - t0) / 100:6.2f} us, start spread {(t[:, 0].max() - t0) / 100:5.2f}, wg life p50 {np.median(t[:, last] - t[:, 0]) / 100:5.2f} "
          f"max {(t[:, last] - t[:, 0]).max() / 100:5.2f} | {ph}")


for exchange in (False, True, False, True):
    F.use_sync_exchange = exchange
    last = 9 if exchange else 4          # 8 -> 9: phase 3 (the LayerNorm behind the product, second counter bank) done and stored
    for it in range(3):
        kv = F.kv_project(vf, [blk.attn.to_kv.weight])[0]
        out, _ = blk(y, vf, ml, hoisted_kv=kv)
        if it:
            show(f"exchange={exchange} fwd[{it}]", last)
        out.backward(torch.ones_like(out))
        if it:
            show(f"exchange={exchange} bwd[{it}]", last)
