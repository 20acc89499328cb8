"""PerceiverResampler and GatedCrossAttentionBlock on the MI355X (through the drop-in modules -> autograd Functions ->
C ABI -> HIP kernels) against (a) the golden vectors produced by the reference and (b) the numpy oracle at sizes the
fixtures do not cover."""
import glob
import os

import numpy as np
import pytest
import torch

from detgen import det, resampler_params, xattn_params
from oracle import flamingo_oracle as O
from util import GOLDEN, TOL, as64, dev, gate_grad_ok, rel

pytestmark = pytest.mark.gpu


def build_resampler(p, dim, depth, heads, dim_head, q, nte, ff_mult, act, dtype):
    from flamingo_mini_amd import PerceiverResampler
    m = PerceiverResampler(dim=dim, depth=depth, dim_head=dim_head, heads=heads, num_latents=q, num_time_embeds=nte, ff_mult=ff_mult, act=act)
    m.load_state_dict({k: torch.as_tensor(np.asarray(v, np.float64)).float() for k, v in p.items()}, strict=True)
    return m.to(dtype).cuda()


def build_block(p, dim, dv, heads, dim_head, n_visual, ff_mult, act, dtype):
    from flamingo_mini_amd import GatedCrossAttentionBlock
    m = GatedCrossAttentionBlock(dim=dim, dim_visual=dv, dim_head=dim_head, heads=heads, ff_mult=ff_mult, act=act, n_visual=n_visual)
    m.load_state_dict({k: torch.as_tensor(np.asarray(v, np.float64)).float() for k, v in p.items()}, strict=True)
    return m.to(dtype).cuda()


def _act(name):
    return "sqrelu" if "sqrelu" in name else ("relu" if "relu" in name else "gelu")


@pytest.mark.parametrize("layerwise", [False, True], ids=["stack-level", "layer-by-layer"])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "rs_*.npz"))), ids=os.path.basename)
def test_resampler_matches_reference_golden_fp32(path, layerwise):
    """The reference's vectors through ff_resampler_fwd / _bwd (the whole stack in one call) and through SURVEY 8-b2's per-layer export set
    (ff_resampler_prologue_* / _layer_* / _epilogue_*, one call and one autograd node per layer: PerceiverResampler.layerwise)."""
    z = np.load(path)
    name = os.path.basename(path)[:-4]
    dim, depth, heads, dim_head, q, nte, ff_mult = [int(v) for v in z["meta"]]
    xshape = tuple(int(v) for v in z["xshape"])
    if "x" in z.files:
        p = {k[2:]: z[k] for k in z.files if k.startswith("p.")}
        x, dy = z["x"], z["dy"]
    else:
        p = resampler_params(dim, depth, heads, dim_head, q, nte, ff_mult, tag=name)
        x, dy = det(xshape, name + "x"), det(z["y"].shape, name + "dy")
    m = build_resampler(p, dim, depth, heads, dim_head, q, nte, ff_mult, _act(name), torch.float32)
    m.layerwise = layerwise
    xd = dev(x).requires_grad_(True)
    y = m(xd)
    assert rel(y, z["y"]) < TOL[torch.float32]["out"]
    y.backward(dev(dy))
    assert rel(xd.grad.reshape(z["dx"].shape), z["dx"]) < TOL[torch.float32]["grad"]
    for k, prm in m.named_parameters():
        assert rel(prm.grad, z["g." + k]) < TOL[torch.float32]["grad"], k


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "xa_*.npz"))), ids=os.path.basename)
def test_xattn_block_matches_reference_golden_fp32(path):
    z = np.load(path)
    name = os.path.basename(path)[:-4]
    dim, dv, heads, dim_head, n_visual, ff_mult, b, L, N = [int(v) for v in z["meta"]]
    if "y" in z.files:
        p = {k[2:]: z[k] for k in z.files if k.startswith("p.")}
        y, vf, dy = z["y"], z["vf"], z["dy"]
    else:
        p = xattn_params(dim, dv, heads, dim_head, ff_mult, tag=name)
        y, vf, dy = det((b, L, dim), name + "y"), det((b, N, n_visual, dv), name + "vf"), det((b, L, dim), name + "dy")
    m = build_block(p, dim, dv, heads, dim_head, n_visual, ff_mult, _act(name), torch.float32)
    yd, vfd = dev(y).requires_grad_(True), dev(vf).requires_grad_(True)
    ml = torch.as_tensor(z["ml"]).cuda()
    out, kv = m(yd, vfd, ml, previous_kv=None, output_kv=True)
    t = TOL[torch.float32]
    assert rel(out, z["y_out"]) < t["out"]
    assert kv[0].shape == z["k"].shape and rel(kv[0], z["k"]) < t["out"] and rel(kv[1], z["v"]) < t["out"]
    out.backward(dev(dy))
    assert rel(yd.grad, z["dy_in"]) < t["grad"]
    assert rel(vfd.grad, z["dvf"]) < t["grad"]
    for k, prm in m.named_parameters():
        ref = z["g." + k]
        if ref.size == 1:
            assert gate_grad_ok(prm.grad.float().cpu().numpy(), ref, t["grad"], z["gs." + k]), (k, float(prm.grad), float(ref.reshape(-1)[0]))
        else:
            assert rel(prm.grad, ref) < t["grad"], k
    # cached decode: last token against the K/V returned above (reference :88-92,102-104)
    with torch.no_grad():
        out_c, _ = m(yd[:, -1:].detach(), None, ml, previous_kv=(kv[0].detach(), kv[1].detach()), output_kv=False)
    assert rel(out_c, z["y_out_cached_last"]) < t["out"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_resampler_vs_oracle_vit_l_shape(dtype):
    """ViT-L/14 geometry (257 tokens, dim 1024, 8x64 heads, 64 latents), two frames, depth 2, batch 3."""
    dim, depth, heads, dh, q, nte, ffm = 1024, 2, 8, 64, 64, 4, 4
    p = resampler_params(dim, depth, heads, dh, q, nte, ffm, tag="vitl")
    m = build_resampler(p, dim, depth, heads, dh, q, nte, ffm, "gelu", dtype)
    xd = dev(det((3, 2, 257, dim), "vitl-x"), dtype).requires_grad_(True)
    dyd = dev(det((3, q, dim), "vitl-dy"), dtype)
    y = m(xd)
    y.backward(dyd)
    p64 = {k: as64(v) for k, v in m.state_dict().items()}
    yr, cache = O.resampler_fwd(as64(xd), p64)
    dxr, gr = O.resampler_bwd(as64(dyd), cache, p64)
    t = TOL[dtype]
    assert rel(y, yr) < t["out"]
    assert rel(xd.grad, dxr) < t["grad"]
    for k, prm in m.named_parameters():
        assert rel(prm.grad, gr[k]) < t["grad"], k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("act", ["gelu", "sqrelu"])
def test_xattn_block_vs_oracle_gpt2_large_shape(dtype, act):
    """gpt2-large geometry (dim 1280, dim_visual 1024), 3 images, L = 96, batch 4, mixed text_time incl. quirk rows."""
    dim, dv, heads, dh, nv, ffm, b, L, N = 1280, 1024, 8, 64, 64, 4, 4, 96, 3
    p = xattn_params(dim, dv, heads, dh, ffm, tag="g2l")
    m = build_block(p, dim, dv, heads, dh, nv, ffm, act, dtype)
    ml = np.zeros((b, L), np.int64)
    ml[0, [0, 30, 61]] = 1
    ml[1, [10, 50]] = 1
    ml[2, [0, 1, 2, 3]] = 1       # four tags, three images -> uniform rows from token 3 on
    yd = dev(det((b, L, dim), "g2l-y"), dtype).requires_grad_(True)
    vfd = dev(det((b, N, nv, dv), "g2l-vf"), dtype).requires_grad_(True)
    dyd = dev(det((b, L, dim), "g2l-dy"), dtype)
    out, _ = m(yd, vfd, torch.as_tensor(ml).cuda())
    out.backward(dyd)
    p64 = {k: as64(v) for k, v in m.state_dict().items()}
    outr, _, cache = O.gated_xattn_block_fwd(as64(yd), as64(vfd), ml, p64, act=act)
    dyr, dvfr, gr = O.gated_xattn_block_bwd(as64(dyd), cache, p64, act=act)
    t = TOL[dtype]
    assert rel(out - yd, outr - as64(yd)) < t["out"]          # error on the block's delta, not hidden by the residual
    assert rel(yd.grad, dyr) < t["grad"]
    assert rel(vfd.grad, dvfr) < t["grad"]
    # natural scale of the two gate gradients from the oracle's cache (cache[2] = attn_out, [3] = ffw_out, [4] / [5] = tanh of the gates)
    dy2 = as64(dyd)
    dy1 = dy2 + O.feedforward_bwd(dy2 * cache[5], cache[1], p64, "ffw.", act, {})
    gscale = {"alpha_ffw": float(np.linalg.norm(dy2 * cache[3])) * float(1.0 - cache[5][0] ** 2),
              "alpha_attn": float(np.linalg.norm(dy1 * cache[2])) * float(1.0 - cache[4][0] ** 2)}
    for k, prm in m.named_parameters():
        if gr[k].size == 1:
            assert gate_grad_ok(prm.grad.float().cpu().numpy(), gr[k], t["grad"], gscale[k]), (k, float(prm.grad), float(gr[k].reshape(-1)[0]))
        else:
            assert rel(prm.grad, gr[k]) < t["grad"], k


def test_zero_gates_make_the_block_an_identity():
    """Fresh init: alpha = 0 -> y passes through and only the alphas receive gradient."""
    from flamingo_mini_amd import GatedCrossAttentionBlock
    torch.manual_seed(0)
    m = GatedCrossAttentionBlock(dim=128, dim_visual=64, heads=2, dim_head=32, n_visual=8).cuda()
    y = torch.randn(2, 8, 128, device="cuda", requires_grad=True)
    vf = torch.randn(2, 1, 8, 64, device="cuda")
    ml = torch.zeros(2, 8, dtype=torch.long, device="cuda"); ml[:, 0] = 1
    out, _ = m(y, vf, ml)
    assert torch.equal(out, y)
    out.square().sum().backward()
    nz = {k for k, prm in m.named_parameters() if float(prm.grad.abs().max()) > 0}
    assert nz == {"alpha_attn", "alpha_ffw"}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_hoisted_kv_projection_equals_per_layer_projection(dtype):
    """functional.kv_project + blocks consuming their slice == every block projecting K / V itself: same outputs, same d y,
    d visual_features (summed over the layers) and parameter gradients.  6 layers -> one full group of 4 and a ragged one."""
    from flamingo_mini_amd import functional as F
    dim, dv, heads, dh, nv, ffm, b, L, N, layers = 256, 128, 4, 32, 16, 2, 3, 24, 2, 6
    ml = np.zeros((b, L), np.int64)
    ml[0, [0, 9]] = 1
    ml[1, [4]] = 1
    ml[2, [0, 1, 2]] = 1
    mlt = torch.as_tensor(ml).cuda()

    def run(hoisted):
        blocks = [build_block(xattn_params(dim, dv, heads, dh, ffm, tag=f"hk{i}"), dim, dv, heads, dh, nv, ffm, "gelu", dtype) for i in range(layers)]
        y = dev(det((b, L, dim), "hk-y"), dtype).requires_grad_(True)
        vf = dev(det((b, N, nv, dv), "hk-vf"), dtype).requires_grad_(True)
        kvs = F.kv_project(vf, [m.attn.to_kv.weight for m in blocks]) if hoisted else [None] * layers
        outs, h = [], y
        for m, kv in zip(blocks, kvs):
            h, _ = m(h, vf, mlt, hoisted_kv=kv)
            outs.append(h)
        (h * dev(det((b, L, dim), "hk-dy"), dtype)).sum().backward()
        return h.detach(), y.grad, vf.grad, [dict(m.named_parameters()) for m in blocks]

    out_a, dy_a, dvf_a, prm_a = run(False)
    out_b, dy_b, dvf_b, prm_b = run(True)
    t = TOL[dtype]
    assert rel(out_b, out_a) < t["out"] and rel(dy_b, dy_a) < t["grad"] and rel(dvf_b, dvf_a) < t["grad"]
    # natural scale of every block's two gate gradients (util.gate_grad_ok) from the float64 oracle chain on the same inputs
    p64 = [{k: as64(v.to(dtype)) for k, v in ((k, torch.as_tensor(np.asarray(v, np.float64)).float()) for k, v in
                                              xattn_params(dim, dv, heads, dh, ffm, tag=f"hk{i}").items())} for i in range(layers)]
    h, caches = as64(dev(det((b, L, dim), "hk-y"), dtype)), []
    vf64 = as64(dev(det((b, N, nv, dv), "hk-vf"), dtype))
    for i in range(layers):
        h, _, c = O.gated_xattn_block_fwd(h, vf64, ml, p64[i], heads=heads, dim_head=dh, n_visual=nv)
        caches.append(c)
    d, gscale = as64(dev(det((b, L, dim), "hk-dy"), dtype)), [None] * layers
    for i in reversed(range(layers)):
        c = caches[i]
        d1 = d + O.feedforward_bwd(d * c[5], c[1], p64[i], "ffw.", "gelu", {})
        gscale[i] = {"alpha_ffw": float(np.linalg.norm(d * c[3])) * float(1.0 - c[5][0] ** 2),
                     "alpha_attn": float(np.linalg.norm(d1 * c[2])) * float(1.0 - c[4][0] ** 2)}
        d, _, _ = O.gated_xattn_block_bwd(d, c, p64[i], heads=heads, dim_head=dh)
    for i, (pa, pb) in enumerate(zip(prm_a, prm_b)):
        for k in pa:
            if pa[k].numel() == 1:
                assert gate_grad_ok(pb[k].grad.float().cpu().numpy(), pa[k].grad.float().cpu().numpy(), t["grad"], gscale[i][k]), (i, k, float(pa[k].grad), float(pb[k].grad))
            else:
                assert rel(pb[k].grad, pa[k].grad) < t["grad"], k


def test_deferred_weight_gradients_accumulate_correctly():
    """The four weight-gradient GEMMs of hoisted-K/V blocks are deferred and grouped (functional._WgradQueue): their gradient tensors
    are handed to autograd unfilled and completed by the end of backward().  A second backward() onto existing .grad (accumulation)
    must therefore not defer; 2 accumulated passes == 2 x one pass, and the queue is empty after every backward()."""
    from flamingo_mini_amd import functional as F
    dim, dv, heads, dh, nv, ffm, b, L, N, layers = 256, 128, 4, 32, 16, 2, 2, 24, 1, 5
    ml = torch.zeros((b, L), dtype=torch.long, device="cuda"); ml[:, 0] = 1
    blocks = [build_block(xattn_params(dim, dv, heads, dh, ffm, tag=f"acc{i}"), dim, dv, heads, dh, nv, ffm, "gelu", torch.float32) for i in range(layers)]
    y = dev(det((b, L, dim), "acc-y"))
    vf = dev(det((b, N, nv, dv), "acc-vf"))
    g = dev(det((b, L, dim), "acc-dy"))

    def backward_once():
        kvs = F.kv_project(vf, [m.attn.to_kv.weight for m in blocks])
        h = y
        for m, kv in zip(blocks, kvs):
            h, _ = m(h, vf, ml, hoisted_kv=kv)
        (h * g).sum().backward()
        assert not F._wgrad_queue.pending

    backward_once()
    once = [{k: p.grad.clone() for k, p in m.named_parameters()} for m in blocks]
    backward_once()                                             # accumulates onto the existing .grad
    for m, ref in zip(blocks, once):
        for k, p in m.named_parameters():
            assert rel(p.grad, 2.0 * ref[k]) < 1e-5, k


@pytest.mark.parametrize("b,L,nv,dim", [(3, 20, 64, 1280), (2, 32, 40, 768), (5, 7, 64, 1536), (4, 32, 64, 1024)],
                         ids=["L20", "L32-40keys", "L7-dim1536", "L32-dim1024"])
def test_resident_fused_kernels_bf16_vs_oracle(b, L, nv, dim):
    """The resident-operand kernels (xa_qattn_fwd_res / xa_dattn_bwd_res: bf16, 64-wide heads, <= 32 tokens, <= 64 keys per sample) at ragged
    sizes - rows past the text, fewer than 64 keys, zero rows (tokens before the media tag) and uniform rows (a second tag without a second
    image), both ring depths - forward, backward and the cached single-token call, against the oracle."""
    dtype = torch.bfloat16
    dv, heads, dh, ffm = 256, 8, 64, 2
    p = xattn_params(dim, dv, heads, dh, ffm, tag=f"res{L}{nv}")
    m = build_block(p, dim, dv, heads, dh, nv, ffm, "gelu", dtype)
    ml = np.zeros((b, L), np.int64)
    ml[0, 0] = 1
    ml[1, min(3, L - 1)] = 1                   # tokens 0..2 see nothing
    if b > 2:
        ml[2, [1, L - 2]] = 1                  # the second tag has no image: uniform rows at the end
    yd = dev(det((b, L, dim), "res-y"), dtype).requires_grad_(True)
    vfd = dev(det((b, 1, nv, dv), "res-vf"), dtype).requires_grad_(True)
    dyd = dev(det((b, L, dim), "res-dy"), dtype)
    mlt = torch.as_tensor(ml).cuda()
    out, kv = m(yd, vfd, mlt, output_kv=True)
    out.backward(dyd)
    p64 = {k: as64(v) for k, v in m.state_dict().items()}
    outr, _, cache = O.gated_xattn_block_fwd(as64(yd), as64(vfd), ml, p64, n_visual=nv)
    dyr, dvfr, gr = O.gated_xattn_block_bwd(as64(dyd), cache, p64)
    t = TOL[dtype]
    assert rel(out - yd, outr - as64(yd)) < t["out"]
    assert rel(yd.grad, dyr) < t["grad"] and rel(vfd.grad, dvfr) < t["grad"]
    for k in ("attn.to_q.weight", "attn.to_kv.weight", "attn.to_out.weight", "attn.norm.weight", "attn.norm.bias", "ffw.1.weight"):
        assert rel(dict(m.named_parameters())[k].grad, gr[k]) < t["grad"], k
    with torch.no_grad():                      # the decode call: one token per sequence against the cached keys / values
        out_c, _ = m(yd[:, -1:].detach(), None, mlt, previous_kv=(kv[0].detach(), kv[1].detach()))
    want = outr[:, -1:] - as64(yd)[:, -1:]
    assert rel(out_c - yd[:, -1:].detach(), want) < t["out"]


@pytest.mark.parametrize("b,L,nv,dim", [(32, 32, 64, 1280), (5, 20, 64, 768), (3, 7, 40, 1536), (64, 32, 64, 1024), (2, 32, 64, 256), (40, 1, 64, 1280)],
                         ids=["config-B", "L20-dim768", "L7-40keys-dim1536", "b64-dim1024", "dim256", "one-token-b40"])
def test_in_launch_exchange_equals_separate_launches_and_oracle(b, L, nv, dim):
    """Round 5: with a sync buffer (ff_xattn_desc.sync) `to_out` + gate + residual run inside the fused LN -> q -> attention launch and
    d LN(y) = d q . Wq inside the fused attention-backward launch - the eight (sample, head) workgroups of a sample exchange their tiles of
    O / d Q through per-sample arrival counters (write-through stores, sc1 loads) instead of through a kernel boundary.
    (1) Same results as the separate launches up to the order of one fp32 sum, and within the bf16 tolerances of the oracle;
    (2) a hand-off must not read stale lines, whatever else the chip is doing and however warm the consumer's caches are: the same call, again
        and again, next to a stream of unrelated matmuls that take CUs away from the launch, gives the SAME BITS every time (MI355X guide:
        test every hand-off under uneven load, checking every word);
    (3) no arrival wait ever timed out (the buffer's status word);
    (4) other contents in the same buffers give the other results (no line of the previous call is served again).
    Since phase 3 (the LayerNorm behind phase 2's output in the same launch, a second bank of counters) the same four points cover it too."""
    from flamingo_mini_amd import functional as F
    dtype = torch.bfloat16
    dv, heads, dh, ffm = 256, 8, 64, 2
    p = xattn_params(dim, dv, heads, dh, ffm, tag=f"xch{L}{nv}{dim}")
    m = build_block(p, dim, dv, heads, dh, nv, ffm, "gelu", dtype)
    ml = np.zeros((b, L), np.int64)
    ml[:, 0] = 1
    if L > 4:
        ml[1, 0] = 0; ml[1, 3] = 1                  # tokens 0..2 of sample 1 see nothing
        if b > 2:
            ml[2, L - 2] = 1                        # a second tag without a second image: uniform rows
    yd = dev(det((b, L, dim), "xch-y"), dtype).requires_grad_(True)
    vfd = dev(det((b, 1, nv, dv), "xch-vf"), dtype).requires_grad_(True)
    dyd = dev(det((b, L, dim), "xch-dy"), dtype)
    mlt = torch.as_tensor(ml).cuda()

    def run():
        for t_ in (yd, vfd, *m.parameters()):
            t_.grad = None
        out, _ = m(yd, vfd, mlt)
        out.backward(dyd)
        torch.cuda.synchronize()
        return [out.detach().clone(), yd.grad.clone(), vfd.grad.clone()] + [q.grad.clone() for q in m.parameters()]

    names = ["out", "dy", "dvf"] + [k for k, _ in m.named_parameters()]
    assert F.use_sync_exchange
    try:
        F.use_sync_exchange = False
        separate = run()
    finally:
        F.use_sync_exchange = True
    fused = run()
    assert F.sync_exchange_status() == 0
    for k, a_, b_ in zip(names, fused, separate):
        if a_.numel() > 1:
            assert rel(a_, b_) < 3e-3, k            # (bf16 roundings of sums taken in another order; measured ~1e-3)
    p64 = {k: as64(v) for k, v in m.state_dict().items()}
    outr, _, cache = O.gated_xattn_block_fwd(as64(yd), as64(vfd), ml, p64, n_visual=nv)
    dyr, dvfr, gr = O.gated_xattn_block_bwd(as64(dyd), cache, p64)
    t = TOL[dtype]
    assert rel(fused[0] - yd.detach(), outr - as64(yd)) < t["out"]
    assert rel(fused[1], dyr) < t["grad"] and rel(fused[2], dvfr) < t["grad"]
    for k, g_ in zip(names[3:], fused[3:]):
        if gr[k].size > 1:
            assert rel(g_, gr[k]) < t["grad"], k
    side = torch.cuda.Stream()
    a_ = torch.randn(4096, 4096, device="cuda", dtype=dtype)
    for i in range(20):
        if i % 2 == 0:
            with torch.cuda.stream(side):
                for _ in range(3):
                    a_ = (a_ @ a_).clamp_(-1, 1)
        again = run()
        for k, x_, y_ in zip(names, again, fused):
            assert torch.equal(x_, y_), (i, k)
    assert F.sync_exchange_status() == 0
    # (4) the same buffers again with OTHER contents: a consumer that was served a line of the previous call by a cache (its own L1, or - when a
    #     sample's heads sit on different XCDs: the odd batch sizes here - a private L2 that still holds what it read last time) would reproduce
    #     the OLD results, which repeating identical inputs cannot notice
    with torch.no_grad():
        yd.mul_(-0.75).add_(0.125)
        dyd.mul_(1.5)
    moved = run()
    assert rel(moved[0], fused[0]) > 0.1 and rel(moved[1], fused[1]) > 0.1          # the inputs really changed the results
    try:
        F.use_sync_exchange = False
        separate2 = run()
    finally:
        F.use_sync_exchange = True
    for k, a_, b_ in zip(names, moved, separate2):
        if a_.numel() > 1:
            assert rel(a_, b_) < 3e-3, k
    assert F.sync_exchange_status() == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("amp", [torch.bfloat16, torch.float16], ids=["bf16-autocast", "fp16-autocast"])
def test_modules_with_fp32_parameters_under_autocast(amp):
    """torch.autocast over fp32 parameters (the reference's mixed-precision recipe): a gated block at gpt2-large geometry and a resampler,
    parameters in fp32, activations arriving in the autocast dtype.  bf16 autocast = the bf16 kernels on casts of the parameters (tolerances of
    the bf16 path against the oracle on what the kernels saw); fp16 autocast = the fp32 kernels (the only error is the fp16 rounding of the
    inputs, which the oracle is fed).  Outputs come back in the dtype the activations came in; gradients arrive on the fp32 parameters."""
    dim, dv, heads, dh, ffm, nv, b, L = 1280, 1024, 8, 64, 4, 64, 4, 32
    p = xattn_params(dim, dv, heads, dh, ffm, tag="amp")
    m = build_block(p, dim, dv, heads, dh, nv, ffm, "gelu", torch.float32)
    ml = np.zeros((b, L), np.int64); ml[:, 0] = 1; ml[1, 0] = 0; ml[1, 3] = 1
    yd = dev(det((b, L, dim), "amp-y"), amp).requires_grad_(True)
    vfd = dev(det((b, 1, nv, dv), "amp-vf"), amp).requires_grad_(True)
    dyd = dev(det((b, L, dim), "amp-dy"), amp)
    with torch.autocast("cuda", dtype=amp):
        out, _ = m(yd, vfd, torch.as_tensor(ml).cuda())
    assert out.dtype == amp
    out.backward(dyd)
    cdt = torch.bfloat16 if amp == torch.bfloat16 else torch.float32
    p64 = {k: as64(v.detach().to(cdt)) for k, v in m.state_dict().items()}                 # what the kernels saw
    outr, _, cache = O.gated_xattn_block_fwd(as64(yd.detach().to(cdt)), as64(vfd.detach().to(cdt)), ml, p64, n_visual=nv)
    dyr, dvfr, gr = O.gated_xattn_block_bwd(as64(dyd.to(cdt)), cache, p64)
    t = dict(TOL[torch.bfloat16]) if amp == torch.bfloat16 else dict(out=2e-3, grad=2e-3)    # fp16: the stored output / gradients are rounded to fp16 once
    assert rel(out.float() - yd.detach().float(), outr - as64(yd.detach().to(cdt))) < max(t["out"], 8e-3 if amp == torch.float16 else 0)
    assert rel(yd.grad, dyr) < t["grad"] and rel(vfd.grad, dvfr) < t["grad"]
    for k, prm in m.named_parameters():
        assert prm.grad is not None and prm.grad.dtype == torch.float32, k
        if gr[k].size > 1:
            assert rel(prm.grad, gr[k]) < t["grad"], k
    # the resampler: fp32 parameters, CLIP features in the autocast dtype
    rp = resampler_params(256, 2, 8, 64, 64, 4, 4, tag="amp-rs")
    rs = build_resampler(rp, 256, 2, 8, 64, 64, 4, 4, "gelu", torch.float32)
    x = dev(det((2, 2, 50, 256), "amp-x"), amp).requires_grad_(True)
    dz = dev(det((2, 64, 256), "amp-dz"), cdt)
    with torch.autocast("cuda", dtype=amp):
        z_ = rs(x)
    assert z_.dtype == cdt                                             # (the dtype the kernels computed in)
    z_.backward(dz)
    rp64 = {k: as64(v.detach().to(cdt)) for k, v in rs.state_dict().items()}
    zr, rc = O.resampler_fwd(as64(x.detach().to(cdt)), rp64)
    dxr, rg = O.resampler_bwd(as64(dz), rc, rp64)
    tr = TOL[torch.bfloat16] if amp == torch.bfloat16 else dict(out=2e-3, grad=2e-3)
    assert rel(z_, zr) < tr["out"] and rel(x.grad, dxr) < tr["grad"]
    for k, prm in rs.named_parameters():
        assert prm.grad is not None and prm.grad.dtype == torch.float32, k
        assert rel(prm.grad, rg[k]) < tr["grad"], k


def _fused_tiles_of_one_block_step(m, yd, vfd, mlt, dyd):
    """profile codes (ff_gemm_profile_record.tile) of the fused cross-attention launches of one forward + backward of block `m`"""
    from flamingo_mini_amd import ffi
    lib = ffi.lib()
    lib.ff_gemm_profile_enable(256)
    try:
        out, _ = m(yd, vfd, mlt)
        out.backward(dyd)
        torch.cuda.synchronize()
        recs = (ffi.GemmProfileRecord * 256)()
        n = lib.ff_gemm_profile_read(recs, 256)
    finally:
        lib.ff_gemm_profile_enable(0)
    return sorted(recs[i].tile for i in range(n) if recs[i].tile <= -4)


@pytest.mark.parametrize("dim,want", [(1280, [-9, -8]), (1536, [-5, -4])], ids=["dim1280-in-launch", "dim1536-separate-launches"])
def test_in_launch_exchange_engages_up_to_the_documented_width(dim, want):
    """include/flamingo_fusion.h documents the widths at which `to_out` / d LN(y) (+ their LayerNorms) run INSIDE the fused launches: dim a
    multiple of 256 up to 1280 (the header said 1536, where the backward kernel's LDS footprint is 165 184 B > 160 KiB and the
    fusion silently never engaged).  At the documented maximum the launch log must show the phase-3 tiles (-8 forward, -9 backward); one
    step beyond it the plain fused launches (-4 / -5) followed by the separate products - and the results are held to the oracle either way
    by test_in_launch_exchange_equals_separate_launches_and_oracle."""
    from flamingo_mini_amd import functional as F
    dtype = torch.bfloat16
    b, L, nv, dv = 8, 32, 64, 256
    p = xattn_params(dim, dv, 8, 64, 2, tag=f"width{dim}")
    m = build_block(p, dim, dv, 8, 64, nv, 2, "gelu", dtype)
    ml = np.zeros((b, L), np.int64); ml[:, 0] = 1
    yd = dev(det((b, L, dim), "w-y"), dtype).requires_grad_(True)
    vfd = dev(det((b, 1, nv, dv), "w-vf"), dtype).requires_grad_(True)
    dyd = dev(det((b, L, dim), "w-dy"), dtype)
    assert F.use_sync_exchange
    assert _fused_tiles_of_one_block_step(m, yd, vfd, torch.as_tensor(ml).cuda(), dyd) == want
    assert F.sync_exchange_status() == 0


@pytest.mark.parametrize("b", [64, 128], ids=["512-workgroups", "1024-workgroups"])
def test_in_launch_exchange_beside_a_persistent_kernel(b):
    """the RCCL-shaped case.  While a second stream HOLDS 32 CUs with a persistent kernel (tests/helpers/cu_hog.hip: one
    64-thread workgroup with 128 KiB of LDS per CU, spinning for the whole test - nothing of the fused kernels fits beside it), the fused
    cross-attention launches run their in-launch hand-offs at 512 and 1024 workgroups, i.e. in several dispatch rounds on the 224 CUs left.
    Every sample's eight workgroups must still meet (status word 0), the results must be the bits of the undisturbed run, and within the
    usual distance of the separate launches.  A third leg hands the SAME buffers to a run on another stream: it must get its own counters
    (functional.ensure_sync_buffer is per device AND stream), not share the first stream's."""
    from flamingo_mini_amd import functional as F
    from util import cu_hog
    dtype = torch.bfloat16
    L, nv, dim, dv, heads, dh, ffm = 32, 64, 1280, 256, 8, 64, 2
    p = xattn_params(dim, dv, heads, dh, ffm, tag="hog")
    m = build_block(p, dim, dv, heads, dh, nv, ffm, "gelu", dtype)
    ml = np.zeros((b, L), np.int64); ml[:, 0] = 1; ml[1, 0] = 0; ml[1, 3] = 1
    yd = dev(det((b, L, dim), "hog-y"), dtype).requires_grad_(True)
    vfd = dev(det((b, 1, nv, dv), "hog-vf"), dtype).requires_grad_(True)
    dyd = dev(det((b, L, dim), "hog-dy"), dtype)
    mlt = torch.as_tensor(ml).cuda()

    def run():
        for t_ in (yd, vfd, *m.parameters()):
            t_.grad = None
        out, _ = m(yd, vfd, mlt)
        out.backward(dyd)
        return [out.detach().clone(), yd.grad.clone(), vfd.grad.clone()] + [q.grad.clone() for q in m.parameters()]

    names = ["out", "dy", "dvf"] + [k for k, _ in m.named_parameters()]
    assert F.use_sync_exchange
    quiet = run()
    torch.cuda.synchronize()
    assert F.sync_exchange_status() == 0
    try:
        F.use_sync_exchange = False
        separate = run()
        torch.cuda.synchronize()
    finally:
        F.use_sync_exchange = True
    hog, side = cu_hog(), torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = hog.cu_hog_launch(32, 128 * 1024, 1500.0, sink.data_ptr(), side.cuda_stream)     # 32 CUs held for 1.5 s
    assert rc == 0, rc
    import time
    time.sleep(0.02)                                    # the persistent workgroups are resident before the first fused launch is issued
    t0 = time.perf_counter()
    for i in range(12):
        crowded = run()
        for k, x_, y_ in zip(names, crowded, quiet):
            assert torch.equal(x_, y_), (i, k)
    torch.cuda.current_stream().synchronize()
    assert not side.query(), f"the persistent kernel ended before the fused launches did ({time.perf_counter() - t0:.2f} s): the test did not test anything"
    assert F.sync_exchange_status() == 0
    for k, a_, b_ in zip(names, crowded, separate):
        if a_.numel() > 1:
            assert rel(a_, b_) < 3e-3, k
    # another stream gets its own counters, and interleaving the two streams' launches disturbs neither
    other = torch.cuda.Stream()
    assert other.cuda_stream != torch.cuda.current_stream().cuda_stream
    other.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(other):
        elsewhere = run()
    again = run()
    torch.cuda.synchronize()
    mine, theirs = F._sync_buffers[F._sync_key(yd.device)], F._sync_buffers[F._sync_key(yd.device, other)]     # (torch hands out streams from a pool:
    assert mine.data_ptr() != theirs.data_ptr()                                                                   # `other` may have had its buffer already)
    for k, x_, y_, z_ in zip(names, elsewhere, again, quiet):
        assert torch.equal(x_, z_) and torch.equal(y_, z_), k
    assert F.sync_exchange_status() == 0
    torch.cuda.synchronize()


def test_denied_co_residency_is_detected_not_silent():
    """The failure itself, not a poked word: a persistent kernel holds 208 of the 256 CUs (26 of every XCD's 32: six left, fewer than a sample's
    eight heads) while a fused forward + backward of 32 samples runs.  A sample's workgroups can no longer be resident together, the arrival waits
    give up after 50 ms each (csrc kSpinTicks / kSpinPolls: 50 ms of the constant 100 MHz clock AND 2^14 polls of their own - a poll takes ~0.2 us
    there, r6s18), and the launches compute on incomplete exchanges.  That must surface, and promptly: the call returns while the
    other kernel still holds its CUs, the status word is 1, check_sync_exchange raises - and once the CUs are free again and the word is cleared
    the same call gives the right answer.  (tools/sessions/r6/denial_probe.py: which hog sizes deny what; with >= 232 CUs held the launches
    BEFORE the fused one already wait for the other kernel to end - slow, but correct: status 0.)"""
    import time
    from flamingo_mini_amd import functional as F
    from util import cu_hog
    dtype = torch.bfloat16
    b, L, nv, dim, dv = 32, 32, 64, 1280, 256
    p = xattn_params(dim, dv, 8, 64, 2, tag="denied")
    m = build_block(p, dim, dv, 8, 64, nv, 2, "gelu", dtype)
    ml = np.zeros((b, L), np.int64); ml[:, 0] = 1
    yd = dev(det((b, L, dim), "den-y"), dtype).requires_grad_(True)
    vfd = dev(det((b, 1, nv, dv), "den-vf"), dtype)
    dyd = dev(det((b, L, dim), "den-dy"), dtype)
    mlt = torch.as_tensor(ml).cuda()

    def run():
        for t_ in (yd, *m.parameters()):
            t_.grad = None
        out, _ = m(yd, vfd, mlt)
        out.backward(dyd)
        torch.cuda.current_stream().synchronize()
        return out.detach().clone(), yd.grad.clone()

    good = run()
    assert F.sync_exchange_status() == 0
    props = torch.cuda.get_device_properties(0)
    if props.multi_processor_count != 256:
        pytest.skip("the scenario is laid out for 8 XCDs of 32 CUs")
    hog, side = cu_hog(), torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert hog.cu_hog_launch(208, 128 * 1024, 3000.0, sink.data_ptr(), side.cuda_stream) == 0      # 26 CUs of every XCD, for 3 s
    time.sleep(0.05)
    t0 = time.perf_counter()
    try:
        run()                                           # finishes (bounded waits), on garbage
        took = time.perf_counter() - t0
        # (typically 0.8 s = sixteen abandoned waits; other launches of the chain may themselves wait for the other kernel's CUs - never longer than it runs)
        assert took < 4.5, f"abandoned waits are bounded at 50 ms each and the other kernel runs 3 s; the call took {took:.2f} s"
        assert F.sync_exchange_status() == 1, f"no hand-off timed out in {took:.2f} s although 208 CUs were held"
        with pytest.raises(F.SyncExchangeTimeout):
            F.check_sync_exchange("test")
    finally:
        side.synchronize()                              # the CUs are free again
        word = F._status_word()
        for buf in F._sync_buffers.values():
            buf.view(torch.int32)[word] = 0
        F._sync_probes.clear()
        torch.cuda.synchronize()
    again = run()
    assert F.sync_exchange_status() == 0
    assert torch.equal(again[0], good[0]) and torch.equal(again[1], good[1])


def test_sync_exchange_timeout_is_raised_not_ignored():
    """The error word cannot be ignored: with the status word of a stream's sync buffer set (what a timed-out arrival wait leaves behind),
    check_sync_exchange raises, the non-blocking poll of the eager optimizers raises on its second call, and a graph-replay step raises at
    its next checkpoint."""
    from flamingo_mini_amd import functional as F
    from flamingo_mini_amd.graphs import GraphedTrainStep
    dtype = torch.bfloat16
    p = xattn_params(256, 128, 8, 64, 2, tag="timeout")
    m = build_block(p, 256, 128, 8, 64, 16, 2, "gelu", dtype)
    ml = torch.zeros((2, 8), dtype=torch.long, device="cuda"); ml[:, 0] = 1
    y = dev(det((2, 8, 256), "to-y"), dtype)
    vf = dev(det((2, 1, 16, 128), "to-vf"), dtype)

    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blk = m

        def forward(self, y, vf, ml):
            out, _ = self.blk(y, vf, ml)

            class R:
                loss = out.float().square().mean()
            return R

    model = Wrap()
    with GraphedTrainStep(model, None, dict(y=y, vf=vf, ml=ml), warmup=1, check_every=2) as step:
        step(); step()
        torch.cuda.synchronize()
        F.check_sync_exchange("clean")                                  # nothing timed out
        F.poll_sync_exchange("clean"); torch.cuda.synchronize(); F.poll_sync_exchange("clean")
        buf = next(iter(F._sync_buffers.values()))
        word = F._status_word()
        try:
            buf.view(torch.int32)[word] = 1                             # what res_await leaves behind when it gives up
            torch.cuda.synchronize()
            with pytest.raises(F.SyncExchangeTimeout):
                F.check_sync_exchange("test")
            F.poll_sync_exchange("test")                                # enqueues the probe that sees the word ...
            torch.cuda.synchronize()
            with pytest.raises(F.SyncExchangeTimeout):
                F.poll_sync_exchange("test")                            # ... and the next call reports it
            with pytest.raises(F.SyncExchangeTimeout):
                step()                                                  # replay 2 is a checkpoint of check_every = 2
        finally:
            buf.view(torch.int32)[word] = 0
            F._sync_probes.clear()
            torch.cuda.synchronize()
    assert F.sync_exchange_status() == 0


@pytest.mark.parametrize("b,L,dim,ffm,act", [(32, 1, 1280, 4, "gelu"), (4, 8, 1280, 4, "gelu"), (16, 2, 2048, 4, "gelu"), (3, 5, 256, 1, "sqrelu"),
                                             (2, 7, 768, 4, "gelu"), (8, 4, 1024, 2, "gelu"), (16, 2, 384, 4, "gelu"), (32, 1, 128, 4, "gelu"),
                                             (5, 6, 640, 2, "sqrelu")],
                         ids=["gpt2-large-decode-b32", "gpt2-large-M32", "opt-1.3b-M32", "tiny-M15", "gpt2-M14", "dim1024-M32",
                              "dim384-odd-multiple-of-128", "dim128-one-piece-per-thread", "dim640-odd-multiple-of-128"])
def test_decode_shaped_feedforward_bf16_vs_oracle(b, L, dim, ffm, act):
    """At most 32 rows (the cached decode step's shape: one token per sequence, batch <= 32) the block's feed-forward half runs on the
    weight-streaming kernels of csrc/ff_decode.hip: LayerNorm + up-projection + activation in one launch (rows resident in LDS, normalised in
    place), the down-projection with its K split over workgroups and combined inside the launch (tickets, last arriver) + tanh gate + residual.
    Forward AND backward against the oracle: the backward consumes what the decode kernels saved (statistics, normalised rows, H, act(H))."""
    dtype = torch.bfloat16
    dv, heads, dh, nv = 128, 8 if dim >= 512 else 2, 64, 16
    p = xattn_params(dim, dv, heads, dh, ffm, tag=f"dec{dim}{ffm}")
    m = build_block(p, dim, dv, heads, dh, nv, ffm, act, dtype)
    ml = np.zeros((b, L), np.int64)
    ml[:, 0] = 1
    if b > 1 and L > 2:
        ml[1, 0] = 0; ml[1, 2] = 1
    y0 = det((b, L, dim), "dec-y")
    if dim in (384, 1024):          # a massive-activation channel in column 0 (the one-pass LayerNorm statistics are shifted by the
        y0 = y0.copy()              # row's leading elements; an outlier there must not cost the variance its digits)
        y0[..., 0] += 8.0           # (8 sigma; much larger values only test the rounding of the stored bf16 output, which `out - y` then isolates)
    yd = dev(y0, dtype).requires_grad_(True)
    vfd = dev(det((b, 1, nv, dv), "dec-vf"), dtype).requires_grad_(True)
    dyd = dev(det((b, L, dim), "dec-dy"), dtype)
    mlt = torch.as_tensor(ml).cuda()
    out, kv = m(yd, vfd, mlt, output_kv=True)
    out.backward(dyd)
    p64 = {k: as64(v) for k, v in m.state_dict().items()}
    outr, _, cache = O.gated_xattn_block_fwd(as64(yd), as64(vfd), ml, p64, heads=heads, dim_head=dh, n_visual=nv, act=act)
    dyr, dvfr, gr = O.gated_xattn_block_bwd(as64(dyd), cache, p64, heads=heads, dim_head=dh, act=act)
    t = dict(TOL[dtype])
    if dim < 256:       # rows of 128 elements do not average the bf16 roundings of the LayerNorm-backward chain the way the published widths
        t["grad"] *= 1.5    # (>= 768) do: measured 1.45e-2 on d y at dim 128 (r5s3) against 0.4-0.9e-2 at every other width of this test.
                            # (The case uses GELU: with ReLU a pre-activation that rounds across zero in bf16 flips a whole gradient term -
                            # 3.8e-2 on d ffw.0.weight at this size, r5s4 - which says nothing about the kernels.)
    assert rel(out - yd, outr - as64(yd)) < t["out"]
    assert rel(yd.grad, dyr) < t["grad"] and rel(vfd.grad, dvfr) < t["grad"]
    for k, prm in m.named_parameters():
        if gr[k].size > 1:
            assert rel(prm.grad, gr[k]) < t["grad"], k
    # the in-launch combine must not depend on which slice arrives last, on cache state or on what else the chip is doing: the same call
    # again and again, next to an unrelated stream of work, gives the same bits
    side = torch.cuda.Stream()
    a_ = torch.randn(2048, 2048, device="cuda", dtype=dtype)
    with torch.no_grad():
        first = m(yd.detach(), vfd.detach(), mlt)[0].clone()
        for i in range(24):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    a_ = (a_ @ a_).clamp_(-1, 1)
            again = m(yd.detach(), vfd.detach(), mlt)[0]
            assert torch.equal(again, first), i
        out_c, _ = m(yd[:, -1:].detach(), None, mlt, previous_kv=(kv[0].detach(), kv[1].detach()))      # the decode call proper
    assert rel(out_c - yd[:, -1:].detach(), outr[:, -1:] - as64(yd)[:, -1:]) < t["out"]
    torch.cuda.synchronize()
