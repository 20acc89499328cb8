#!/bin/bash
ulimit -c 0
# Round 5, session 17 (debugging phase 3): the `saved` regions of one block forward with and without it
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
for shape in "8 32 256" "3 20 1280"; do
  set -- $shape; n=$1_$2_$3
  FF_XATTN_LN3=0 python tools/sessions/r5/ln3_debug.py /tmp/off_$n $shape 2>&1 | tail -1
  FF_XATTN_LN3=1 python tools/sessions/r5/ln3_debug.py /tmp/on_$n $shape 2>&1 | tail -1
  python - $n <<'P'
import numpy as np, json, sys
n = sys.argv[1]
a, b = np.load(f"/tmp/off_{n}_saved.npy"), np.load(f"/tmp/on_{n}_saved.npy")
reg = json.load(open(f"/tmp/off_{n}_regions.json"))
for k, (o, sz) in reg.items():
    x, y = a[o:o + sz], b[o:o + sz]
    if k in ("mean_a", "rstd_a", "lse", "mean_f", "rstd_f"):
        xf, yf = x.view(np.float32), y.view(np.float32)
    else:
        xf = (x.view(np.uint16).astype(np.uint32) << 16).view(np.float32); yf = (y.view(np.uint16).astype(np.uint32) << 16).view(np.float32)
    d = np.abs(xf - yf); bad = ~np.isfinite(yf)
    print(f"{n} {k:9s} max|off-on| {np.nanmax(d):10.4g}  rel {np.linalg.norm(np.nan_to_num(xf - yf)) / (np.linalg.norm(xf) + 1e-30):9.3g}  nonfinite(on) {int(bad.sum())}  first off/on: {xf[:4]} / {yf[:4]}")
P
done
