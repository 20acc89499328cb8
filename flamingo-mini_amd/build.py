"""Build libflamingo_fusion.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m flamingo_mini_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to <package>/csrc/_obj, the library next to this file so it
travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libflamingo_fusion.so")
SOURCES = ["ff_api.hip", "ff_gemm.hip", "ff_rowwise.hip", "ff_attention.hip", "ff_xattn_fused.hip", "ff_optim.hip", "ff_loss.hip", "ff_elementwise.hip", "ff_decode.hip"]
HEADERS = ["ff_common.h", "ff_internal.h", "ff_gemm_tiles.h", "ff_attention_core.h", os.path.join("..", "..", "include", "flamingo_fusion.h")]
ARCH = "gfx950"
# kernarg preload: leading scalar kernel arguments arrive in SGPRs at wave launch (gfx940+); kernels fall back to loads on old firmware
# -fvisibility=hidden: only what include/flamingo_fusion.h declares (under its visibility pragma) is exported - no mangled C++ internals
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=16"] + os.environ.get("FF_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (needs ROCm >= 7.0 for gfx950)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, debug: bool = False) -> str:
    """debug=True: the development build libflamingo_fusion_debug.so (-DFF_DEBUG: A/B switches read from the environment), selected with
    FLAMINGO_FUSION_LIB=debug; the shipped library reads no environment variable."""
    hipcc = _hipcc()
    obj_dir = os.path.join(CSRC, "_obj_debug" if debug else "_obj")
    lib_path = LIB_PATH.replace(".so", "_debug.so") if debug else LIB_PATH
    flags = FLAGS + (["-DFF_DEBUG"] if debug else [])
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers) or not os.path.exists(o.replace(".o", ".resources.txt")):      # (objects of a build that kept no resource remarks)
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr[-4000:]}")
        # the compiler's register / scratch / occupancy remarks of every kernel of the unit: tools/kernel_resources.py prints them and
        # tests/test_cabi.py holds the kernels whose co-residency the launch plans assume to their budgets
        with open(o.replace(".o", ".resources.txt"), "w") as f:
            f.write("\n".join(l for l in r.stderr.splitlines() if "kernel-resource-usage" in l) + "\n")
        if verbose:
            print("compiled", os.path.basename(s))
        return o

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib_path, objs + [os.path.join(CSRC, "exports.map")]):
        r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", f"-Wl,--version-script={os.path.join(CSRC, 'exports.map')}", "-o", lib_path] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print("linked", lib_path)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, debug="--debug" in sys.argv))
