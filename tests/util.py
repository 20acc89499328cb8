"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerances (relative L2 error against the fp64 oracle / reference vectors), set from MEASURED errors (tools/tol_report.py over a
# `FF_TOL_REPORT=... pytest -m gpu` run on an MI355X, round 2) with a 1.3-2x margin - the kernels are bitwise deterministic, so the
# margin only has to absorb a different tile / split-K plan:
#   fp32 path (MFMA fp32 FMA chains + __expf): measured <= 2.7e-6 everywhere (BASELINE.json's logits target is 1e-3)  -> 5e-6 / 1e-5
#   bf16 path (inputs / activations rounded to bf16 at every kernel boundary): primitives <= 2.7e-3, modules at real geometry
#              <= 6.7e-3 on outputs (six stacked resampler layers, config D) and <= 9.7e-3 on gradients                -> 8e-3 / 1.2e-2
#              The reference's own bf16-vs-fp32 deviation is 6.6e-3 (resampler) / 2.9e-3 (xattn block).
TOL = {torch.float32: dict(out=5e-6, grad=1e-5), torch.bfloat16: dict(out=8e-3, grad=1.2e-2)}
# A THIRD class: the WHOLE drop-in model in bf16 against the reference's float64 vectors - the stock CLIP and GPT-2 stacks run in
# bf16 on PyTorch-ROCm here too, so every stock op rounds on top of the fusion path's own error and the module tolerances above do not apply.
# Measured on an MI355X (rounds 2 and 4): logits 8.9e-3 (tiny GPT-2 fixture) / 6.5e-3 (h64 fixture, step 1), worst parameter gradient
# 1.5e-2 / 1.1e-2, loss |difference| <= 1.2e-2; step 2 of the h64 fixture (weights have left the bf16 grid) is held to twice these bounds,
# measured 2.0e-2 / 3.6e-2.  Used by tests/test_model_plumbing.py::test_full_gpt2_model_bf16_on_hip and ::test_h64_bf16_two_steps_...
TOL_FULL_BF16 = dict(out=1.5e-2, grad=2.5e-2, loss=3e-2)


def gate_grad_ok(got, ref, tol, scale) -> bool:
    """THE rule for the scalar tanh-gate gradients (alpha_attn, alpha_ffw), used by every test that checks one:
            |got - ref| <= tol * (scale + |ref|)
    A gate gradient is a dot product over all b * L * dim elements, (1 - tanh^2 alpha) * sum(d branch_sum .* branch): `scale` is the
    natural size of such a sum of rounded products, (1 - tanh^2 alpha) * || d branch_sum .* branch ||_2 (what it would be if the terms
    did not cancel), |ref| covers relative errors common to all terms.  The golden fixtures carry the scale next to every gate gradient
    (`gs.<name>`, tests/golden/make_golden.py: GateProbe); oracle-based tests compute it from the oracle's cache."""
    return abs(float(np.asarray(got, np.float64).reshape(-1)[0]) - float(np.asarray(ref, np.float64).reshape(-1)[0])) <= \
        tol * (float(np.asarray(scale).reshape(-1)[0]) + abs(float(np.asarray(ref, np.float64).reshape(-1)[0])))


def rel(a, b) -> float:
    a = np.asarray(a.detach().double().cpu().numpy() if torch.is_tensor(a) else a, np.float64)
    b = np.asarray(b.detach().double().cpu().numpy() if torch.is_tensor(b) else b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    r = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    if os.environ.get("FF_TOL_REPORT"):      # tools/tol_report.py: the measured errors the tolerances above are set from
        with open(os.environ["FF_TOL_REPORT"], "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{r:.3e}\n")
    return r


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a, np.float64)).to(dtype).cuda()


def rnd(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return (g.standard_normal(shape) * scale).astype(np.float32)


def as64(t):
    """what the kernel actually saw (after rounding to its dtype), as float64 numpy"""
    return t.detach().double().cpu().numpy()


def build_cu_hog() -> str:
    """Compile tests/helpers/cu_hog.hip -> libcu_hog.so if stale (hipcc cross-compiles without a GPU; __graft_entry__.build() calls this so
    that the helper travels to the GPU box prebuilt)."""
    import subprocess
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers")
    src, so = os.path.join(here, "cu_hog.hip"), os.path.join(here, "libcu_hog.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        hipcc = next(c for c in ("/opt/rocm/bin/hipcc", "hipcc") if c == "hipcc" or os.path.exists(c))
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, src], check=True, capture_output=True)
    return so


def cu_hog():
    """ctypes handle of tests/helpers/libcu_hog.so (a kernel that holds CUs for a given time - a test helper, not part of the product
    library).  `lib.cu_hog_launch(blocks, lds_bytes, milliseconds, sink_ptr, stream)` returns a hipError_t."""
    import ctypes
    lib = ctypes.CDLL(build_cu_hog())
    lib.cu_hog_launch.restype = ctypes.c_int
    lib.cu_hog_launch.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    return lib
