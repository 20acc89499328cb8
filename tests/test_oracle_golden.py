"""The numpy oracle (oracle/flamingo_oracle.py) against golden vectors produced by the REFERENCE
(tests/golden/make_golden.py).  This is what pins the oracle; everything on the GPU is then checked
against the oracle and against the same vectors."""
import glob
import os

import numpy as np
import pytest

from detgen import det, resampler_params, xattn_params
from oracle import flamingo_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def load_rs(path):
    z = np.load(path)
    name = os.path.basename(path)[:-4]
    dim, depth, heads, dim_head, q, nte, ff_mult = [int(v) for v in z["meta"]]
    xshape = tuple(int(v) for v in z["xshape"])
    if "x" in z.files:
        p = {k[2:]: z[k] for k in z.files if k.startswith("p.")}
        x, dy = z["x"], z["dy"]
        tol = 1e-11
    else:
        p = {k: v.astype(np.float64) for k, v in resampler_params(dim, depth, heads, dim_head, q, nte, ff_mult, tag=name).items()}
        x = det(xshape, name + "x").astype(np.float64)
        dy = det(z["y"].shape, name + "dy").astype(np.float64)
        assert abs(sum(np.abs(v).sum() for v in p.values()) - z["digest"][0]) < 1e-6 * z["digest"][0]
        tol = 2e-6  # outputs stored as float32
    act = "gelu" if "gelu" in name or "geom" in name else ("sqrelu" if "sqrelu" in name else "relu")
    return z, p, x, dy, dict(heads=heads, dim_head=dim_head, act=act), tol


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "rs_*.npz"))), ids=os.path.basename)
def test_resampler_oracle_matches_reference(path):
    z, p, x, dy, kw, tol = load_rs(path)
    y, cache = O.resampler_fwd(x, p, **kw)
    assert rel(y, z["y"]) < tol
    dx, grads = O.resampler_bwd(dy, cache, p, **kw)
    assert rel(dx.reshape(z["dx"].shape), z["dx"]) < tol
    gkeys = [k[2:] for k in z.files if k.startswith("g.")]
    assert sorted(gkeys) == sorted(grads.keys()) == sorted(p.keys())
    for k in gkeys:
        assert rel(grads[k], z["g." + k]) < tol, k


def load_xa(path):
    z = np.load(path)
    name = os.path.basename(path)[:-4]
    dim, dv, heads, dim_head, n_visual, ff_mult, b, L, N = [int(v) for v in z["meta"]]
    if "y" in z.files:
        p = {k[2:]: z[k] for k in z.files if k.startswith("p.")}
        y, vf, dy = z["y"], z["vf"], z["dy"]
        tol = 1e-11
    else:
        p = {k: v.astype(np.float64) for k, v in xattn_params(dim, dv, heads, dim_head, ff_mult, tag=name).items()}
        y = det((b, L, dim), name + "y").astype(np.float64)
        vf = det((b, N, n_visual, dv), name + "vf").astype(np.float64)
        dy = det((b, L, dim), name + "dy").astype(np.float64)
        tol = 2e-6
    act = "sqrelu" if "sqrelu" in name else ("relu" if "relu" in name else "gelu")
    return z, p, y, vf, dy, dict(heads=heads, dim_head=dim_head, act=act), n_visual, tol


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "xa_*.npz"))), ids=os.path.basename)
def test_xattn_block_oracle_matches_reference(path):
    z, p, y, vf, dy, kw, n_visual, tol = load_xa(path)
    out, (k, v), cache = O.gated_xattn_block_fwd(y, vf, z["ml"], p, n_visual=n_visual, **kw)
    assert rel(out, z["y_out"]) < tol
    assert rel(k, z["k"]) < tol and rel(v, z["v"]) < tol
    dyin, dvf, grads = O.gated_xattn_block_bwd(dy, cache, p, **kw)
    assert rel(dyin, z["dy_in"]) < tol
    assert rel(dvf, z["dvf"]) < tol
    gkeys = [k_[2:] for k_ in z.files if k_.startswith("g.")]
    assert sorted(gkeys) == sorted(grads.keys()) == sorted(p.keys())
    for k_ in gkeys:
        assert rel(grads[k_], z["g." + k_]) < tol, k_
    # cached-decode path: last token against reused K/V
    out_c, _, _ = O.gated_xattn_block_fwd(y[:, -1:], None, z["ml"], p, n_visual=n_visual, previous_kv=(k, v), **kw)
    assert rel(out_c, z["y_out_cached_last"]) < tol
    assert rel(out_c, out[:, -1:]) < 1e-9


def test_mask_quirks_pinned():
    """F2/F3 of SURVEY.md: equality mask, zero rows before any image, uniform rows past the last image."""
    tt = O.text_time_of(np.array([[0, 0, 1, 0, 0, 1, 0, 1]]))
    assert tt.tolist() == [[0, 0, 1, 1, 1, 2, 2, 3]]
    allow, no_media = O.attention_masks(tt, n_media=2, n_visual=4)
    assert allow[0, 0, 2].tolist() == [True] * 4 + [False] * 4      # == not >=
    assert allow[0, 0, 5].tolist() == [False] * 4 + [True] * 4
    assert not allow[0, 0, 7].any() and not allow[0, 0, 0].any()
    assert no_media[0, 0, :, 0].tolist() == [True, True] + [False] * 6


def test_flop_model_matches_survey():
    assert abs(O.resampler_flops_fwd(32, 1, 257, 1024) / 1e9 - 369.3) < 0.1
    assert abs(O.xattn_block_flops_fwd(32, 32, 1280, 1024) / 1e9 - 33.96) < 0.01


# ---------------------------------------------------------------------------------------------------
# The torch CPU restatement (oracle/torch_port.py: what bench.py times as `cpu_baseline`, all host cores, backward by autograd)
# is held to the same reference-generated vectors.
# ---------------------------------------------------------------------------------------------------
def _t(a, grad=False):
    import torch
    return torch.from_numpy(np.asarray(a, np.float64)).requires_grad_(grad)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "rs_*.npz"))), ids=os.path.basename)
def test_resampler_torch_port_matches_reference(path):
    from oracle import torch_port as TP
    z, p, x, dy, kw, tol = load_rs(path)
    pt = {k: _t(v, True) for k, v in p.items()}
    xt = _t(x, True)
    y = TP.resampler(xt, pt, **kw)
    assert rel(y.detach().numpy(), z["y"]) < tol
    y.backward(_t(dy))
    assert rel(xt.grad.numpy().reshape(z["dx"].shape), z["dx"]) < tol
    for k in pt:
        assert rel(pt[k].grad.numpy(), z["g." + k]) < tol, k


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(G, "xa_*.npz"))), ids=os.path.basename)
def test_xattn_block_torch_port_matches_reference(path):
    import torch
    from oracle import torch_port as TP
    z, p, y, vf, dy, kw, n_visual, tol = load_xa(path)
    pt = {k: _t(v, True) for k, v in p.items()}
    yt, vft = _t(y, True), _t(vf, True)
    ml = torch.from_numpy(np.asarray(z["ml"], np.int64))
    out, (k, v) = TP.gated_xattn_block(yt, vft, ml, pt, n_visual=n_visual, **kw)
    assert rel(out.detach().numpy(), z["y_out"]) < tol and rel(k.detach().numpy(), z["k"]) < tol and rel(v.detach().numpy(), z["v"]) < tol
    out.backward(_t(dy))
    assert rel(yt.grad.numpy(), z["dy_in"]) < tol and rel(vft.grad.numpy(), z["dvf"]) < tol
    for k_ in pt:
        assert rel(pt[k_].grad.numpy(), z["g." + k_]) < tol, k_
    with torch.no_grad():
        out_c, _ = TP.gated_xattn_block(yt[:, -1:], None, ml, pt, n_visual=n_visual, previous_kv=(k, v), **kw)
    assert rel(out_c.numpy(), z["y_out_cached_last"]) < tol


@pytest.mark.skipif(not os.path.isdir(os.environ.get("FLAMINGO_REFERENCE", "/root/reference")),
                    reason="regenerating a fixture imports the reference (build container only)")
def test_interchange_fixture_regenerates_bit_for_bit(tmp_path):
    """VERDICT r04 (fixture hygiene): tests/golden/make_interchange.py pins every BLAS pool to one thread and refuses nondeterministic torch
    algorithms, so a fresh run in the build container reproduces the committed file byte for byte (a child process: the generator patches
    transformers' from_pretrained and sets thread limits)."""
    import subprocess
    import sys
    env = dict(os.environ, FLAMINGO_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(G, "make_interchange.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(os.path.join(G, "interchange_gpt2_tiny.npz"), "rb") as f, open(tmp_path / "interchange_gpt2_tiny.npz", "rb") as g:
        assert f.read() == g.read()
