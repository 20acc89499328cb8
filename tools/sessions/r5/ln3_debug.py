"""Development aid (session r5s17): one gated block forward at a small resident-kernel shape; dumps the library's `saved` buffer regions so that two runs
(development build, FF_XATTN_LN3 = 0 / 1) can be compared region by region."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from flamingo_mini_amd import functional as F, ffi
torch.manual_seed(0)
b, L, d, dv, H, dh, ffm, nv = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), 256, 8, 64, 2, 64
dt = torch.bfloat16
g = torch.Generator().manual_seed(1)
def r(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).to(dt).cuda()
params = [torch.tensor([0.5]).to(dt).cuda(), torch.tensor([0.3]).to(dt).cuda(), 1 + r(d, sc=0.1), r(d, sc=0.1), r(H * dh, d, sc=0.05), r(2 * H * dh, dv, sc=0.05),
          r(d, H * dh, sc=0.05), 1 + r(d, sc=0.1), r(d, sc=0.1), r(ffm * d, d, sc=0.05), r(d, ffm * d, sc=0.05)]
y = r(b, L, d); vf = r(b, 1, nv, dv)
ml = torch.zeros(b, L, dtype=torch.int64).cuda(); ml[:, 0] = 1
tt = F.text_time(ml)
lib = ffi.lib()
desc = F._xattn_desc(y, 1, nv, dv, (H, dh, ffm, "gelu"), tt)
saved = F._empty_bytes(lib.ff_xattn_saved_bytes(desc), y.device); saved.zero_()
scratch = F._empty_bytes(lib.ff_xattn_scratch_bytes(desc), y.device)
out = torch.empty_like(y)
ffi.check(lib.ff_xattn_block_fwd(desc, y.data_ptr(), vf.data_ptr(), tt.data_ptr(), ffi.ptr_array(params), None, None, out.data_ptr(), saved.data_ptr(), saved.numel(),
                                 scratch.data_ptr(), scratch.numel(), ffi.stream_handle(y.device)), "fwd")
torch.cuda.synchronize()
M = b * L; inner = H * dh; Nk = nv
def al(n): return (n + 255) // 256 * 256
off = 0; regions = {}
for name, n in [("KV", b * Nk * 2 * inner * 2), ("mean_a", M * 4), ("rstd_a", M * 4), ("yn", M * d * 2), ("Qs", M * inner * 2), ("lse", b * H * L * 4), ("O", M * inner * 2),
                ("attn_out", M * d * 2), ("y1", M * d * 2), ("mean_f", M * 4), ("rstd_f", M * 4), ("xn_f", M * d * 2), ("Hpre", M * ffm * d * 2), ("Aact", M * ffm * d * 2), ("ffw_out", M * d * 2)]:
    regions[name] = (off, n); off = al(off + n)
np.save(sys.argv[1] + "_saved.npy", saved.cpu().numpy()); np.save(sys.argv[1] + "_out.npy", out.float().cpu().numpy())
json.dump(regions, open(sys.argv[1] + "_regions.json", "w"))
print("status", F.sync_exchange_status(), "out finite", bool(torch.isfinite(out.float()).all()), "saved bytes", saved.numel(), "layout end", off)
