"""The code path bench.py times, against the oracle, at the benchmark's geometry.

Config B runs bf16 with 64-wide heads - the fused LayerNorm -> Q projection -> attention kernel (xa_qattn_fwd_kernel), its backward mirror
(xa_dattn_bwd_kernel) - with the K / V projection of all layers hoisted (ff_kv_project_*), the data-gradient-only block backward
(ff_xattn_block_bwd_kv_data), the weight gradients of four blocks per grouped launch (ff_xattn_wgrad_grouped: one full group and a ragged
one here) and the final LayerNorm / gate reductions deferred with them.  The module-level oracle tests call `block(y, vf, ml)`, i.e. the
per-layer, non-deferred entry points; this file drives six gpt2-large-geometry blocks through `functional.kv_project` + `hoisted_kv=`
exactly like FlamingoBaseModel.forward does, in bf16 and fp32, at the benchmark's batch (32 x 32 tokens, one image) and at a ragged one
(batch 5, 96 tokens = the multi-tile kernels, three images, rows with text_time 0 and beyond the last image).

Two comparisons per case, both against oracle.gated_xattn_block_fwd / _bwd (float64):
  per block  - the oracle is fed what the block actually received on the device (its input rows and the incoming gradient, both as stored
               in the run's dtype), so every block's outputs and every one of its parameter gradients are held to the plain tolerances;
  chained    - the oracle runs the six blocks on its own float64 intermediates: the end-to-end output, d y, d visual_features.
The resampler stack of config B (depth 6, 257 CLIP tokens, batch 32) gets the same treatment against oracle.resampler_fwd / _bwd.
"""
import numpy as np
import pytest
import torch

from detgen import det, resampler_params, xattn_params
from oracle import flamingo_oracle as O
from test_hip_modules import build_block, build_resampler
from util import TOL, as64, dev, rel

pytestmark = pytest.mark.gpu

DIM, DV, HEADS, DH, NV, FFM = 1280, 1024, 8, 64, 64, 4
LAYERS = 6


# Rounding of a stored activation: a block's output x_out = x_in + delta is kept in the run's dtype, so |x_out - oracle| carries the rounding
# of x_out itself (relative 2^-9 / sqrt(3) rms for bf16's 8 significand bits, 2^-25 / sqrt(3) for fp32) on top of the error of delta.  With
# small gates (tanh(alpha) ~ 0.05) |delta| << |x_out| and that term dominates: it is part of the bound, not of the kernels' error.
ROUND_RMS = {torch.bfloat16: 2.0 ** -9 / 3 ** 0.5, torch.float32: 2.0 ** -25 / 3 ** 0.5}


def run_hoisted_chain(blocks, y, vf, ml, g):
    """What FlamingoBaseModel.forward does with hoist_kv: one grouped projection of every layer's K / V, then the blocks in sequence.
    Returns the hidden states h_0 .. h_n (with .grad retained)."""
    from flamingo_mini_amd import functional as F
    kvs = F.kv_project(vf, [m.attn.to_kv.weight for m in blocks])
    hs = [y]
    for m, kv in zip(blocks, kvs):
        h, _ = m(hs[-1], vf, ml, hoisted_kv=kv)
        h.retain_grad()
        hs.append(h)
    hs[-1].backward(g)
    assert not F._wgrad_queue.pending
    return hs


def check_chain_against_oracle(blocks, hs, y, vf, ml_np, g, dtype, act="gelu", chain_factor=3.0):
    t = TOL[dtype]
    p64 = [{k: as64(v) for k, v in m.state_dict().items()} for m in blocks]
    vf64 = as64(vf)
    worst = {}

    def note(name, err, bound):      # collected, asserted together at the end (one GPU run shows every margin)
        worst[name] = max(worst.get(name, 0.0), err / bound)

    # ---- per block, on the device's own inputs ----
    dvf_sum = np.zeros_like(vf64)
    for i, m in enumerate(blocks):
        x_in, x_out = as64(hs[i]), as64(hs[i + 1])
        out_r, _, cache = O.gated_xattn_block_fwd(x_in, vf64, ml_np, p64[i], act=act)
        delta = float(np.linalg.norm(out_r - x_in))
        # (two stored activations of magnitude |x| sit between x_in and x_out: y1 and the output itself)
        note(f"block{i}.out", float(np.linalg.norm(x_out - out_r)) / delta, t["out"] + 2.5 * ROUND_RMS[dtype] * float(np.linalg.norm(out_r)) / delta)
        dy2 = as64(hs[i + 1].grad)
        dy_r, dvf_r, g_r = O.gated_xattn_block_bwd(dy2, cache, p64[i], act=act)
        # the two gate gradients are dot products over all b * L * dim elements, (1 - tanh^2 alpha) * sum(d branch_sum .* branch): their error is
        # held to the tolerance times the natural scale of such a sum, || a .* b ||_2 (cache[2] = attn_out, cache[3] = ffw_out of the oracle)
        dy1 = dy2 + O.feedforward_bwd(dy2 * cache[5], cache[1], p64[i], "ffw.", act, {})       # gradient at the attention branch's sum
        gate_scale = {"alpha_ffw": float(np.linalg.norm(dy2 * cache[3])) * float(1.0 - cache[5][0] ** 2),
                      "alpha_attn": float(np.linalg.norm(dy1 * cache[2])) * float(1.0 - cache[4][0] ** 2)}
        note(f"block{i}.dy", rel(hs[i].grad, dy_r), t["grad"])
        dvf_sum += dvf_r
        for k, prm in m.named_parameters():
            assert prm.grad is not None and bool(torch.isfinite(prm.grad.float()).all()), (i, k)
            if g_r[k].size == 1:
                # the rule of util.gate_grad_ok (|| a .* b ||_2: the noise floor of a sum of rounded products; |sum| itself: relative errors
                # that are common to all terms), written as a ratio so that the worst case can be reported
                note(f"block{i}.{k}", abs(float(prm.grad) - float(g_r[k])), t["grad"] * (gate_scale[k] + abs(float(g_r[k])) + 1e-30))
            else:
                note(f"block{i}.{k}", rel(prm.grad, g_r[k]), t["grad"])
    note("dvf(sum of per-block oracles)", rel(vf.grad, dvf_sum), t["grad"] * 1.5)
    # ---- chained: the oracle on its own float64 intermediates ----
    h = as64(y)
    caches = []
    for i in range(len(blocks)):
        h, _, c = O.gated_xattn_block_fwd(h, vf64, ml_np, p64[i], act=act)
        caches.append(c)
    dtot = float(np.linalg.norm(h - as64(y)))
    note("chain.out", float(np.linalg.norm(as64(hs[-1]) - h)) / dtot, t["out"] * chain_factor + 1.5 * ROUND_RMS[dtype] * float(np.linalg.norm(h)) / dtot)
    d, dvf = as64(g), np.zeros_like(vf64)
    for i in reversed(range(len(blocks))):
        d, dvf_i, _ = O.gated_xattn_block_bwd(d, caches[i], p64[i], act=act)
        dvf += dvf_i
    note("chain.dy", rel(y.grad, d), t["grad"] * chain_factor)
    note("chain.dvf", rel(vf.grad, dvf), t["grad"] * chain_factor)
    return worst


def _case(dtype, b, L, N, ml_np, tag, act="gelu", group=None):
    from flamingo_mini_amd import functional as F
    saved_group = F._wgrad_queue.group
    if group is not None:
        F._wgrad_queue.group = group
    try:
        _run_case(dtype, b, L, N, ml_np, tag, act)
    finally:
        F._wgrad_queue.group = saved_group


def _run_case(dtype, b, L, N, ml_np, tag, act):
    blocks = [build_block(xattn_params(DIM, DV, HEADS, DH, FFM, alpha_attn=0.5 - 0.05 * i, alpha_ffw=-0.4 - 0.06 * i, tag=f"{tag}{i}"),
                          DIM, DV, HEADS, DH, NV, FFM, act, dtype) for i in range(LAYERS)]
    y = dev(det((b, L, DIM), tag + "y"), dtype).requires_grad_(True)
    vf = dev(det((b, N, NV, DV), tag + "vf"), dtype).requires_grad_(True)
    g = dev(det((b, L, DIM), tag + "g"), dtype)
    ml = torch.as_tensor(ml_np).cuda()
    hs = run_hoisted_chain(blocks, y, vf, ml, g)
    worst = check_chain_against_oracle(blocks, hs, y, vf, ml_np, g, dtype, act=act)
    print(f"[benchpath {tag} {dtype}] worst error / bound:", {k: round(v, 3) for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:8]}, flush=True)
    bad = {k: round(v, 3) for k, v in worst.items() if not v < 1.0}
    assert not bad, f"error / bound >= 1: {bad}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_config_B_hoisted_deferred_blocks_full_batch(dtype):
    """32 sequences x 32 tokens, one image (media tag at token 0): the benchmark's shapes, the resident-operand fused kernels, all six
    blocks' weight gradients in one grouped call (the single-GPU default batches up to 12)."""
    ml = np.zeros((32, 32), np.int64); ml[:, 0] = 1
    _case(dtype, 32, 32, 1, ml, "BP")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_hoisted_deferred_blocks_ragged_batch_long_text(dtype):
    """batch 5, 96 tokens (three query tiles: the multi-tile forward and the separate d K / d V launch), three images, rows before the first
    tag (text_time 0 -> zero rows) and after a fourth tag (text_time > n_media -> uniform rows), squared-ReLU feed-forward."""
    b, L, N = 5, 96, 3
    ml = np.zeros((b, L), np.int64)
    ml[0, [0, 30, 61]] = 1
    ml[1, [10, 50]] = 1
    ml[2, [0, 1, 2, 3]] = 1
    ml[3, [95]] = 1
    ml[4, [5, 6, 40]] = 1
    _case(dtype, b, L, N, ml, "BR", act="sqrelu", group=4)      # six blocks: one grouped weight-gradient call of four and a ragged one of two


@pytest.mark.parametrize("layerwise", [False, True], ids=["stack-level", "layer-by-layer"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_config_B_resampler_full_batch(dtype, layerwise):
    """The resampler of config B at its full batch: (32, 1, 257, 1024) CLIP-L features, depth 6, grouped weight gradients over 4 + 2 layers
    (stack-level call) resp. one library call per layer (the data-parallel launch structure: ff_resampler_layer_*)."""
    dim, depth, b = 1024, 6, 32
    p = resampler_params(dim, depth, HEADS, DH, 64, 4, 4, tag="BPrs")
    m = build_resampler(p, dim, depth, HEADS, DH, 64, 4, 4, "gelu", dtype)
    m.layerwise = layerwise
    xd = dev(det((b, 1, 257, dim), "BPrs-x"), dtype).requires_grad_(True)
    dyd = dev(det((b, 64, dim), "BPrs-dy"), dtype)
    y = m(xd)
    y.backward(dyd)
    p64 = {k: as64(v) for k, v in m.state_dict().items()}
    yr, cache = O.resampler_fwd(as64(xd), p64)
    dxr, gr = O.resampler_bwd(as64(dyd), cache, p64)
    t = TOL[dtype]
    assert rel(y, yr) < t["out"]
    assert rel(xd.grad, dxr) < t["grad"]
    worst = {}
    for k, prm in m.named_parameters():
        worst[k] = rel(prm.grad, gr[k])
    bad = {k: v for k, v in worst.items() if not v < t["grad"]}
    print(f"[benchpath resampler {dtype}] worst parameter-gradient errors:", sorted(worst.items(), key=lambda kv: -kv[1])[:4], flush=True)
    assert not bad, bad


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "f32"])
def test_reused_block_and_accumulated_gradients_on_the_deferred_path(dtype):
    """The two situations in which a block must NOT defer its weight gradients, on the real kernels at config B's geometry: a module applied
    twice in one backward pass (weight sharing / activation-checkpoint recompute: the engine sums the two contributions the moment the second
    one is returned, functional._WgradQueue.seen_in_this_pass) and a second backward onto existing .grad tensors (gradient accumulation).
    Oracle: the chain m0 -> m1 -> m0 in float64; the gradients of m0 are the sum of its two uses, and after the second pass twice that."""
    from flamingo_mini_amd import functional as F
    b, L, N, tag = 4, 32, 1, "reuse"
    t = TOL[dtype]
    blocks = [build_block(xattn_params(DIM, DV, HEADS, DH, FFM, alpha_attn=0.5 - 0.1 * i, alpha_ffw=-0.4 - 0.1 * i, tag=f"{tag}{i}"),
                          DIM, DV, HEADS, DH, NV, FFM, "gelu", dtype) for i in range(2)]
    order = [0, 1, 0]
    ml_np = np.zeros((b, L), dtype=bool)
    ml_np[:, 0] = True
    ml = torch.as_tensor(ml_np).cuda()
    y = dev(det((b, L, DIM), tag + "y"), dtype).requires_grad_(True)
    vf = dev(det((b, N, NV, DV), tag + "vf"), dtype).requires_grad_(True)
    g = dev(det((b, L, DIM), tag + "g"), dtype)

    def one_pass():
        kvs = F.kv_project(vf, [m.attn.to_kv.weight for m in blocks])
        h = y
        for i in order:
            h, _ = blocks[i](h, vf, ml, hoisted_kv=kvs[i])
        h.backward(g)
        assert not F._wgrad_queue.pending
        return h

    out = one_pass()
    p64 = [{k: as64(v) for k, v in m.state_dict().items()} for m in blocks]
    vf64, h, caches = as64(vf), as64(y), []
    for i in order:
        h, _, c = O.gated_xattn_block_fwd(h, vf64, ml_np, p64[i], act="gelu")
        caches.append(c)
    assert rel(out, h) < t["out"] * 3
    d, dvf = as64(g), np.zeros_like(vf64)
    grads = [dict(), dict()]
    gscale = [dict(alpha_ffw=0.0, alpha_attn=0.0), dict(alpha_ffw=0.0, alpha_attn=0.0)]      # natural scale of the gate gradients (util.gate_grad_ok), summed over a block's uses
    for pos in reversed(range(len(order))):
        i = order[pos]
        c = caches[pos]
        d1 = d + O.feedforward_bwd(d * c[5], c[1], p64[i], "ffw.", "gelu", {})
        gscale[i]["alpha_ffw"] += float(np.linalg.norm(d * c[3])) * float(1.0 - c[5][0] ** 2)
        gscale[i]["alpha_attn"] += float(np.linalg.norm(d1 * c[2])) * float(1.0 - c[4][0] ** 2)
        d, dvf_i, g_r = O.gated_xattn_block_bwd(d, caches[pos], p64[i], act="gelu")
        dvf += dvf_i
        for k, v in g_r.items():
            grads[i][k] = grads[i].get(k, 0.0) + v
    worst = {}

    def check(scale, what):
        worst[f"{what}.dy"] = rel(y.grad, scale * d) / (t["grad"] * 3)
        worst[f"{what}.dvf"] = rel(vf.grad, scale * dvf) / (t["grad"] * 3)
        for i, m in enumerate(blocks):
            for k, prm in m.named_parameters():
                ref = scale * grads[i][k]
                if ref.size == 1:       # gate gradients: the rule of util.gate_grad_ok, written as a ratio
                    worst[f"{what}.block{i}.{k}"] = abs(float(prm.grad) - float(ref)) / (t["grad"] * 3 * (scale * gscale[i][k] + abs(float(ref))) + 1e-30)
                else:
                    worst[f"{what}.block{i}.{k}"] = rel(prm.grad, ref) / (t["grad"] * 3)

    check(1.0, "reuse")
    one_pass()                          # no zero_grad: every .grad exists now, so nothing defers and autograd accumulates
    check(2.0, "accumulate")
    print(f"[benchpath {tag} {dtype}] worst error / bound:", {k: round(v, 3) for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]}, flush=True)
    bad = {k: round(v, 3) for k, v in worst.items() if not v < 1.0}
    assert not bad, f"error / bound >= 1: {bad}"
