"""Which hog sizes deny the fused cross-attention launches co-residency (r6 session 11)?  For each (held CUs, batch): one forward + backward of a
gated block while tests/helpers/cu_hog holds that many CUs; prints the time it took, whether the hog was still running at the end, the status word."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
from detgen import det, xattn_params
from util import cu_hog, dev
from flamingo_mini_amd import GatedCrossAttentionBlock, functional as F

hog = cu_hog()
dtype = torch.bfloat16
p = xattn_params(1280, 256, 8, 64, 2, tag="probe")
m = GatedCrossAttentionBlock(dim=1280, dim_visual=256, dim_head=64, heads=8, ff_mult=2, act="gelu", n_visual=64)
m.load_state_dict({k: torch.as_tensor(np.asarray(v, np.float64)).float() for k, v in p.items()})
m = m.to(dtype).cuda()
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
for held, b in [(0, 32), (208, 32), (224, 32), (200, 64), (128, 64)]:
    ml = torch.zeros((b, 32), dtype=torch.long, device="cuda"); ml[:, 0] = 1
    y = dev(det((b, 32, 1280), "pr-y"), dtype).requires_grad_(True)
    vf = dev(det((b, 1, 64, 256), "pr-vf"), dtype)
    dy = dev(det((b, 32, 1280), "pr-dy"), dtype)
    out, _ = m(y, vf, ml); out.backward(dy); torch.cuda.synchronize()        # warm
    side = torch.cuda.Stream()
    if held:
        assert hog.cu_hog_launch(held, 128 * 1024, 2500.0, sink.data_ptr(), side.cuda_stream) == 0
        time.sleep(0.05)
    t0 = time.perf_counter()
    e = torch.cuda.Event(enable_timing=True); e2 = torch.cuda.Event(enable_timing=True)
    e.record()
    a = torch.ones(1024, device="cuda") + 1          # a trivial stock kernel first: does ANYTHING of this stream run beside the hog?
    e2.record(); e2.synchronize()
    t_trivial = time.perf_counter() - t0
    out, _ = m(y, vf, ml); out.backward(dy)
    torch.cuda.current_stream().synchronize()
    took = time.perf_counter() - t0
    held_still = held and not side.query()
    st = F.sync_exchange_status()
    side.synchronize()
    w = F._status_word()
    raw = [int(buf.view(torch.int32)[w]) for buf in F._sync_buffers.values()]
    st = f"{st} (raw words {raw})"
    for buf in F._sync_buffers.values():
        buf.view(torch.int32)[w] = 0
    torch.cuda.synchronize()
    print(f"held {held:3d} CUs, batch {b:2d}: trivial kernel done after {t_trivial * 1e3:7.1f} ms, block fwd+bwd after {took * 1e3:8.1f} ms, hog still running: {bool(held_still)}, status word {st}", flush=True)
