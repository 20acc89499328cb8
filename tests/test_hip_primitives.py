"""HIP kernels, one at a time, through the C ABI against numpy (float64) on the same (dtype-rounded) inputs."""
import numpy as np
import pytest
import torch

from oracle import flamingo_oracle as O
from util import TOL, as64, dev, rel, rnd

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.bfloat16]


def F():
    from flamingo_mini_amd import functional
    return functional


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("al,bl", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (200, 136, 72), (1024, 512, 1280), (96, 1024, 4104)])
def test_gemm_layouts(dtype, al, bl, M, N, K):
    A = dev(rnd((M, K) if al == 0 else (K, M), 1), dtype)
    B = dev(rnd((N, K) if bl == 0 else (K, N), 2), dtype)
    C = F().gemm(A, B, a_layout=al, b_layout=bl, scale=0.5)
    a, b = as64(A), as64(B)
    ref = 0.5 * (a if al == 0 else a.T) @ (b.T if bl == 0 else b)
    assert rel(C, ref) < TOL[dtype]["out"] * 0.5


@pytest.mark.parametrize("al,bl", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_every_instantiated_tile(al, bl):
    """Force each bf16 block tile (ff_gemm_desc.tile / .stages) on a ragged problem: partial tiles in M and N, a K tail, split-K."""
    M, N, K = 408, 424, 328
    A = dev(rnd((M, K) if al == 0 else (K, M), 11), torch.bfloat16)
    B = dev(rnd((N, K) if bl == 0 else (K, N), 12), torch.bfloat16)
    a, b = as64(A), as64(B)
    ref = (a if al == 0 else a.T) @ (b.T if bl == 0 else b)
    # 128002 / 128160: the 8-wave producer / consumer kernels (128 x 128 for every layout; 128 x 160 for a K-major A operand, B K-major or -
    # staged as a 128-column and a 32-column piece - N-contiguous)
    # 128168: the 128 x 160 tile with eight MFMA waves (4 x 2) + four DMA waves (round 4)
    # 256128: 256 x 128 tiles, eight MFMA + eight DMA waves (round 5: products with >= 4096 rows); with an M-major A operand (the weight
    #         gradients' d Y^T: 512-byte k-rows staged two per DMA instruction, transposing fragment reads) eight MFMA + four DMA waves
    # 256256: 256 x 256 tiles, sixteen waves that both issue the DMA and run the MFMAs, five-unit ring, the fp32 tile leaves in two halves
    #         (round 5; K-major operands)
    for tile in (128, 6412, 64, 64002, 128002, 256128) + ((128160, 128168) if al == 0 else ()) + ((256256,) if (al, bl) == (0, 0) else ()) + ((3264,) if (al, bl) == (0, 0) else ()):
        for stages in (2, 3, 4):
            for split in (1, 2):
                C = F().gemm(A, B, a_layout=al, b_layout=bl, split_k=split, tile=tile, stages=stages)
                assert rel(C, ref) < 1e-2, (tile, stages, split)


@pytest.mark.parametrize("bl", [0, 1], ids=["B-K-major", "B-N-contiguous"])
def test_gemm_large_row_tile_weight_gradient_layouts(bl):
    """The 256 x 128 tile with an M-major A operand (selectable through ff_gemm_desc.tile, not planned: measured equal to 128 x 128 on config E's
    weight gradients) at such a shape: >= 4096 output rows, >= 4096 contraction rows, ragged N and a K tail."""
    dt = torch.bfloat16
    M, N, K = 4096, 2056, 4096 + 40
    A = dev(rnd((K, M), 41, 0.5), dt)
    B = dev(rnd((N, K) if bl == 0 else (K, N), 42, 0.05), dt)
    ref = as64(A).T @ (as64(B).T if bl == 0 else as64(B))
    out = F().gemm(A, B, a_layout=1, b_layout=bl, tile=256128)
    assert rel(out, ref) < TOL[dt]["out"]
    assert rel(out, F().gemm(A, B, a_layout=1, b_layout=bl)) < 1e-3            # the planned 128 x 128 launch


@pytest.mark.parametrize("bl", [0, 1], ids=["B-K-major", "B-N-contiguous"])
def test_gemm_large_row_tile_with_the_feed_forward_epilogues(bl):
    """The 256 x 128 tile the planner picks for products with >= 4096 rows (config E: 4 x 1024 tokens), at a shape it picks it for (checked
    through ff_gemm_plan) with the epilogues of the feed-forward launches: activation + saved pre-activation, activation gradient x saved
    pre-activation x gate, gated residual - K-major and N-contiguous weight."""
    import ctypes as C
    from flamingo_mini_amd import ffi
    dt = torch.bfloat16
    M, N, K = 4096, 2048, 1088
    d = ffi.GemmDesc(ffi.DTYPE_BF16, M, N, K, 0, bl, ffi.rowmap(K), ffi.rowmap(K if bl == 0 else N), ffi.rowmap(N), 1.0, ffi.ACT_NONE, ffi.ACT_NONE, 0)
    bm, bn, sk = C.c_int(), C.c_int(), C.c_int()
    assert ffi.lib().ff_gemm_plan(d, bm, bn, sk) == 0 and (bm.value, bn.value, sk.value) == (256, 128, 1)
    gate = dev(np.array([0.7]), dt)
    g = np.tanh(as64(gate)[0])
    A = dev(rnd((M, K), 31, 0.5), dt)
    B = dev(rnd((N, K) if bl == 0 else (K, N), 32, 0.05), dt)
    R, H = dev(rnd((M, N), 33), dt), dev(rnd((M, N), 34), dt)
    acc, r, h = as64(A) @ (as64(B).T if bl == 0 else as64(B)), as64(R), as64(H)
    t = TOL[dt]["out"]
    C_, aux = F().gemm(A, B, b_layout=bl, act="gelu", want_aux_out=True)
    assert rel(aux, acc) < t and rel(C_, O.act_fwd(acc, "gelu")) < t
    C_ = F().gemm(A, B, b_layout=bl, act_bwd="gelu", aux_in=H, gate=gate)
    assert rel(C_, O.act_bwd(g * acc, h, "gelu")) < t
    C_ = F().gemm(A, B, b_layout=bl, residual=R, gate=gate)
    assert rel(C_, r + g * acc) < t
    C_ = F().gemm(A[:4000], B, b_layout=bl)                     # a partial row tile
    assert rel(C_, acc[:4000]) < t


@pytest.mark.parametrize("K", [1088 + 24, 136, 64], ids=["17-k-steps-and-a-tail", "3-k-steps", "1-k-step"])
def test_gemm_256_square_tile_with_the_feed_forward_epilogues(K):
    """The 256 x 256 tile (gemm_bf16_u16_kernel, K-major operands) with every epilogue of the feed-forward launches, on a grid with partial tiles in
    both directions, for k-loops long enough to wrap the five-unit ring, and for one and three k-steps (the ring's prologue and tail)."""
    dt = torch.bfloat16
    M, N = 1160, 1320
    gate = dev(np.array([0.7]), dt)
    g = np.tanh(as64(gate)[0])
    A = dev(rnd((M, K), 51, 0.5), dt)
    B = dev(rnd((N, K), 52, 0.05), dt)
    R, H = dev(rnd((M, N), 53), dt), dev(rnd((M, N), 54), dt)
    acc, r, h = as64(A) @ as64(B).T, as64(R), as64(H)
    t = TOL[dt]["out"]
    for tile in (256256,):
        kw = dict(tile=tile)
        assert rel(F().gemm(A, B, **kw), acc) < t
        C_, aux = F().gemm(A, B, act="gelu", want_aux_out=True, **kw)
        assert rel(aux, acc) < t and rel(C_, O.act_fwd(acc, "gelu")) < t
        C_ = F().gemm(A, B, act_bwd="gelu", aux_in=H, gate=gate, **kw)
        assert rel(C_, O.act_bwd(g * acc, h, "gelu")) < t
        C_ = F().gemm(A, B, residual=R, gate=gate, **kw)
        assert rel(C_, r + g * acc) < t
        if K > 512:
            assert rel(F().gemm(A, B, split_k=3, **kw), acc) < t
    assert torch.equal(F().gemm(A, B, tile=256256, split_k=1), F().gemm(A, B, tile=128002, split_k=1))       # every tile adds an element's products in the same order


def test_gemm_balanced_producer_consumer_tile():
    """The one-tile-per-CU kernel (128 x 160 tiles, 4 MFMA waves + 4 DMA waves) at the two shapes the planner picks it for - the gated
    block's FFW up-projection 1024 x 5120 x 1280 (256 tiles) and down-projection 1024 x 1280 x 5120 (64 tiles x split-K 4) - with their
    real epilogues, and forced onto a ragged problem (partial tiles in M and N, K tail) with every epilogue."""
    from flamingo_mini_amd import ffi
    dt = torch.bfloat16
    gate = dev(np.array([0.7]), dt)
    g = np.tanh(as64(gate)[0])
    A, B = dev(rnd((1024, 1280), 21, 0.5), dt), dev(rnd((5120, 1280), 22, 0.05), dt)
    C, aux = F().gemm(A, B, act="gelu", want_aux_out=True)
    acc = as64(A) @ as64(B).T
    assert rel(aux, acc) < TOL[dt]["out"] and rel(C, O.act_fwd(acc, "gelu")) < TOL[dt]["out"]
    A, B, R = dev(rnd((1024, 5120), 23, 0.5), dt), dev(rnd((1280, 5120), 24, 0.02), dt), dev(rnd((1024, 1280), 25), dt)
    C = F().gemm(A, B, residual=R, gate=gate)
    assert rel(C, as64(R) + g * (as64(A) @ as64(B).T)) < TOL[dt]["out"]
    M, N, K = 200, 336, 1096
    R, H = dev(rnd((M, N), 28), dt), dev(rnd((M, N), 29), dt)
    A = dev(rnd((M, K), 26, 0.5), dt)
    for bl in (0, 1):        # B K-major, and N-contiguous (the data gradients dY . W: the 160-wide tile staged as 128 + 32 columns)
        B = dev(rnd((N, K) if bl == 0 else (K, N), 27, 0.05), dt)
        acc, r, h = as64(A) @ (as64(B).T if bl == 0 else as64(B)), as64(R), as64(H)
        for tile_code in (128160, 128168):       # four / eight consumer waves
            kw = dict(b_layout=bl, tile=tile_code)
            for split_k in (1, 3):
                for act in ("gelu", "sqrelu", "relu"):
                    C, aux = F().gemm(A, B, act=act, want_aux_out=True, split_k=split_k, **kw)
                    assert rel(aux, acc) < TOL[dt]["out"] and rel(C, O.act_fwd(acc, act)) < TOL[dt]["out"]
                    C = F().gemm(A, B, act_bwd=act, aux_in=H, gate=gate, split_k=split_k, **kw)
                    assert rel(C, acc * g * O.act_bwd(np.ones_like(h), h, act)) < TOL[dt]["out"]
                C = F().gemm(A, B, residual=R, gate=gate, split_k=split_k, **kw)
                assert rel(C, r + g * acc) < TOL[dt]["out"]
                assert rel(F().gemm(A, B, scale=0.25, split_k=split_k, **kw), 0.25 * acc) < TOL[dt]["out"]
    # the data-gradient shapes the planner now hands to it: d H = d y2 . W3 (1024 x 5120 x 1280) and d xn = d H . W1 (1024 x 1280 x 5120, split-K)
    A, B = dev(rnd((1024, 1280), 31, 0.5), dt), dev(rnd((1280, 5120), 32, 0.05), dt)
    assert rel(F().gemm(A, B, b_layout=1), as64(A) @ as64(B)) < TOL[dt]["out"]
    A, B = dev(rnd((1024, 5120), 33, 0.5), dt), dev(rnd((5120, 1280), 34, 0.02), dt)
    assert rel(F().gemm(A, B, b_layout=1), as64(A) @ as64(B)) < TOL[dt]["out"]


@pytest.mark.parametrize("tile", [0, 3264], ids=["weight-streaming", "32x64-tiles"])
def test_gemm_decode_rows_bf16(tile):
    """M <= 32 rows (one decoded token per sequence) on the gated block's decode products, with their epilogues: the weight-streaming kernel
    the planner picks (gemm_bf16_rows32_kernel: 16 output columns per workgroup, K split over its waves, no tiles) and the 32 x 64 tile plan
    it replaced (still what a K that is not a multiple of 32 gets)."""
    dt = torch.bfloat16
    gate = dev(np.array([0.7]), dt)
    g = np.tanh(as64(gate)[0])
    kw = dict(tile=tile) if tile else {}
    for M in (32, 17, 5, 1):
        A, B, R = dev(rnd((M, 512), 41, 0.5), dt), dev(rnd((1280, 512), 42, 0.05), dt), dev(rnd((M, 1280), 43), dt)
        C, aux = F().gemm(A, B, residual=R, gate=gate, want_aux_out=True, **kw)              # to_out + gate + residual
        acc = as64(A) @ as64(B).T
        assert rel(aux, acc) < TOL[dt]["out"] and rel(C, as64(R) + g * acc) < TOL[dt]["out"]
        A, B = dev(rnd((M, 1280), 44, 0.5), dt), dev(rnd((5120, 1280), 45, 0.05), dt)
        C, aux = F().gemm(A, B, act="gelu", want_aux_out=True, **kw)                         # FFW up-projection
        acc = as64(A) @ as64(B).T
        assert rel(aux, acc) < TOL[dt]["out"] and rel(C, O.act_fwd(acc, "gelu")) < TOL[dt]["out"]
        A, B, R = dev(rnd((M, 5120), 46, 0.5), dt), dev(rnd((1280, 5120), 47, 0.02), dt), dev(rnd((M, 1280), 48), dt)
        C = F().gemm(A, B, residual=R, gate=gate, **kw)                                      # FFW down-projection: long K
        assert rel(C, as64(R) + g * (as64(A) @ as64(B).T)) < TOL[dt]["out"]
        A, B = dev(rnd((M, 1280), 49, 0.5), dt), dev(rnd((512, 1280), 50, 0.05), dt)
        C = F().gemm(A, B, scale=0.125, **kw)                                                # to_q * scale
        assert rel(C, 0.125 * (as64(A) @ as64(B).T)) < TOL[dt]["out"]
        # ragged: a partial last column group (N % 16 = 4), k-steps that do not divide among the waves (41), and a K tail (the tile plan)
        for N, K in ((1284, 1312), (1284, 1304)):
            A, B, H = dev(rnd((M, K), 51, 0.5), dt), dev(rnd((N, K), 52, 0.05), dt), dev(rnd((M, N), 53), dt)
            acc, h = as64(A) @ as64(B).T, as64(H)
            for act in ("gelu", "sqrelu", "relu"):
                C = F().gemm(A, B, act_bwd=act, aux_in=H, gate=gate, **kw)
                assert rel(C, acc * g * O.act_bwd(np.ones_like(h), h, act)) < TOL[dt]["out"], (M, N, K, act)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("split_k", [0, 3])
@pytest.mark.parametrize("N", [264, 132], ids=["n264", "n132-ragged"])      # 132: N % 8 == 4 -> element-wise epilogue pieces
def test_gemm_epilogues(dtype, split_k, N):
    M, K = 136, 1024
    A, B = dev(rnd((M, K), 3, 0.5), dtype), dev(rnd((N, K), 4, 0.05), dtype)
    R, H = dev(rnd((M, N), 5), dtype), dev(rnd((M, N), 6), dtype)
    gate = dev(np.array([0.7]), dtype)
    a, b, r, h, g = as64(A), as64(B), as64(R), as64(H), np.tanh(as64(gate)[0])
    acc = a @ b.T
    for act in ("gelu", "sqrelu", "relu"):
        C, aux = F().gemm(A, B, act=act, want_aux_out=True, split_k=split_k)
        assert rel(aux, acc) < TOL[dtype]["out"]
        assert rel(C, O.act_fwd(acc, act)) < TOL[dtype]["out"]
        C = F().gemm(A, B, act_bwd=act, aux_in=H, gate=gate, split_k=split_k)
        assert rel(C, acc * g * O.act_bwd(np.ones_like(h), h, act)) < TOL[dtype]["out"]
    C, aux = F().gemm(A, B, residual=R, gate=gate, want_aux_out=True, split_k=split_k)
    assert rel(aux, acc) < TOL[dtype]["out"]
    assert rel(C, r + g * acc) < TOL[dtype]["out"]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("rows,cols", [(37, 64), (1000, 1024), (64, 1280), (5, 36)])
def test_layernorm_fwd_bwd(dtype, rows, cols):
    x, g, b = dev(rnd((rows, cols), 1, 2.0), dtype), dev(1 + 0.2 * rnd((cols,), 2), dtype), dev(0.1 * rnd((cols,), 3), dtype)
    dy, res = dev(rnd((rows, cols), 4), dtype), dev(rnd((rows, cols), 5), dtype)
    y, mean, rstd = F().layernorm_fwd(x, g, b)
    yr, cache = O.layernorm_fwd(as64(x), as64(g), as64(b))
    assert rel(y, yr) < TOL[dtype]["out"]
    assert rel(mean, as64(x).mean(-1)) < 1e-5 and rel(rstd, cache[1][:, 0]) < 1e-4
    dx, dg, db = F().layernorm_bwd(dy, x, g, mean, rstd, dx_residual=res)
    dxr, dgr, dbr = O.layernorm_bwd(as64(dy), cache, as64(g))
    assert rel(dx, dxr + as64(res)) < TOL[dtype]["grad"]
    assert rel(dg, dgr) < TOL[dtype]["grad"] and rel(db, dbr) < TOL[dtype]["grad"]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_layernorm_time_embedding_addend(dtype):
    b, T, v, D = 3, 2, 5, 64
    x, tpe = dev(rnd((b * T * v, D), 1), dtype), dev(rnd((4, D), 2), dtype)
    g, be = dev(1 + 0.2 * rnd((D,), 3), dtype), dev(0.1 * rnd((D,), 4), dtype)
    y, mean, rstd = F().layernorm_fwd(x, g, be, add=tpe, add_rows_per_seg=T * v, add_div=v)
    xx = as64(x).reshape(b, T, v, D) + as64(tpe)[:T][None, :, None, :]
    yr, cache = O.layernorm_fwd(xx.reshape(-1, D), as64(g), as64(be))
    assert rel(y, yr) < TOL[dtype]["out"]
    dy = dev(rnd((b * T * v, D), 5), dtype)
    dx, dg, db = F().layernorm_bwd(dy, x, g, mean, rstd, add=tpe, add_rows_per_seg=T * v, add_div=v)
    dxr, dgr, dbr = O.layernorm_bwd(as64(dy), cache, as64(g))
    assert rel(dx, dxr) < TOL[dtype]["grad"] and rel(dg, dgr) < TOL[dtype]["grad"] and rel(db, dbr) < TOL[dtype]["grad"]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_rows_reduce_and_gate_grad(dtype):
    b, T, v, D = 4, 3, 7, 128
    x = dev(rnd((b * T * v, D), 1), dtype)
    out = F().rows_reduce(x, T * v, v)                       # d time_pos_emb pattern
    assert rel(out, as64(x).reshape(b, T, v, D).sum((0, 2))) < TOL[dtype]["grad"]
    out = F().rows_reduce(x, T * v, 1)                       # d latents pattern
    assert rel(out, as64(x).reshape(b, T * v, D).sum(0)) < TOL[dtype]["grad"]
    a, c, alpha = dev(rnd((300, 256), 2), dtype), dev(rnd((300, 256), 3), dtype), dev(np.array([0.3]), dtype)
    ga = F().gate_grad(a, c, alpha)
    ref = (as64(a) * as64(c)).sum() * (1 - np.tanh(as64(alpha)[0]) ** 2)
    assert abs(float(ga.float().cpu()) - ref) < TOL[dtype]["grad"] * max(1.0, abs(ref)) * 5


def test_text_time_matches_cumsum():
    ml = torch.tensor([[0, 0, 1, 0, 0, 1, 0, 1], [1, 0, 0, 0, 1, 0, 0, 0]])
    for t in (ml, ml.int(), ml.bool()):
        tt = F().text_time(t.cuda())
        assert tt.dtype == torch.int32 and tt.cpu().tolist() == O.text_time_of(ml.numpy()).tolist()


def _attn_ref(q, k, v, tt, n_visual):
    """numpy attention on (b, n, h, d) arrays with the reference's masking rules; returns o and a backward closure."""
    qh, kh, vh = (t.transpose(0, 2, 1, 3) for t in (q, k, v))
    sim = qh @ kh.transpose(0, 1, 3, 2)
    if tt is not None:
        allow, no_media = O.attention_masks(tt, k.shape[1] // n_visual, n_visual)
        sim = np.where(allow, sim, -np.finfo(np.float64).max)
    p = O._softmax_lastdim(sim)
    if tt is not None:
        p = np.where(no_media, 0.0, p)
    o = (p @ vh).transpose(0, 2, 1, 3)

    def bwd(do):
        doh = do.transpose(0, 2, 1, 3)
        dp = doh @ vh.transpose(0, 1, 3, 2)
        if tt is not None:
            dp = np.where(no_media, 0.0, dp)
        dv = p.transpose(0, 1, 3, 2) @ doh
        ds = p * (dp - (dp * p).sum(-1, keepdims=True))
        if tt is not None:
            ds = np.where(allow, ds, 0.0)
        return tuple(t.transpose(0, 2, 1, 3) for t in (ds @ kh, ds.transpose(0, 1, 3, 2) @ qh, dv))
    return o, bwd


@pytest.mark.parametrize("dtype,dh", [(torch.float32, 64), (torch.float32, 16), (torch.bfloat16, 64), (torch.bfloat16, 32)],
                         ids=["f32-64", "f32-16", "bf16-64", "bf16-32"])
@pytest.mark.parametrize("nq,nkv", [(64, 321), (8, 18), (64, 114)])
def test_attention_dense(dtype, dh, nq, nkv):
    b, h = 2, 3
    q, k, v = (dev(rnd((b, n, h, dh), s, sc), dtype) for n, s, sc in ((nq, 1, dh ** -0.5), (nkv, 2, 1.0), (nkv, 3, 1.0)))
    do = dev(rnd((b, nq, h, dh), 4), dtype)
    o, lse = F().attention_fwd(q, k, v)
    oref, bwd = _attn_ref(as64(q), as64(k), as64(v), None, 0)
    assert rel(o, oref) < TOL[dtype]["out"]
    dq, dk, dv = F().attention_bwd(q, k, v, o, do, lse)
    for got, ref, name in zip((dq, dk, dv), bwd(as64(do)), "qkv"):
        assert rel(got, ref) < TOL[dtype]["grad"], name


@pytest.mark.parametrize("dtype,dh,nv", [(torch.float32, 16, 8), (torch.float32, 64, 64), (torch.bfloat16, 64, 64)], ids=["f32-toy", "f32", "bf16"])
def test_attention_media_mask_quirks(dtype, dh, nv):
    """equality mask, zero rows (text_time == 0), uniform rows (text_time > n_media): SURVEY.md F2/F3."""
    b, h, L, N = 3, 2, 70, 2
    ml = np.zeros((b, L), np.int64)
    ml[0, [0, 33]] = 1
    ml[1, [5, 40, 66]] = 1          # leading zero rows; third tag with only two images -> uniform rows
    tt = O.text_time_of(ml)
    q, k, v = (dev(rnd((b, n, h, dh), s, sc), dtype) for n, s, sc in ((L, 1, dh ** -0.5), (N * nv, 2, 1.0), (N * nv, 3, 1.0)))
    do = dev(rnd((b, L, h, dh), 4), dtype)
    ttd = torch.as_tensor(tt, dtype=torch.int32).cuda()
    o, lse = F().attention_fwd(q, k, v, tt=ttd, n_visual=nv)
    oref, bwd = _attn_ref(as64(q), as64(k), as64(v), tt, nv)
    assert rel(o, oref) < TOL[dtype]["out"]
    assert float(o[2].abs().max()) == 0.0 and float(o[1, :5].abs().max()) == 0.0     # exact zeros before any image
    dq, dk, dv = F().attention_bwd(q, k, v, o, do, lse, tt=ttd, n_visual=nv)
    for got, ref, name in zip((dq, dk, dv), bwd(as64(do)), "qkv"):
        assert rel(got, ref) < TOL[dtype]["grad"], name
    assert float(dq[1, 66:].abs().max()) == 0.0                                          # uniform rows: no gradient to q


def test_online_softmax_rescale_is_exercised():
    """A key far above the rest in a LATER tile forces the running-max rescale branch (guide rule 26)."""
    b, h, dh, nq, nkv = 1, 1, 64, 64, 200
    q, k, v = rnd((b, nq, h, dh), 1, 0.3), rnd((b, nkv, h, dh), 2, 0.3), rnd((b, nkv, h, dh), 3)
    k[0, 150, 0] = q[0, 7, 0] * 40.0          # spike against query 7 in the third key tile
    qd, kd, vd = dev(q), dev(k), dev(v)
    o, lse = F().attention_fwd(qd, kd, vd)
    oref, _ = _attn_ref(as64(qd), as64(kd), as64(vd), None, 0)
    assert rel(o, oref) < 2e-5 and rel(o[0, 7], oref[0, 7]) < 2e-5
