#!/bin/bash
ulimit -c 0
# Round 5, session 11: resident fused cross-attention kernels with an XCD holding 4 heads x batch / 4 samples instead of 8 heads x batch / 8 samples
# (development build, FF_XATTN_XCD_SPLIT): parity, same-box A/B of the step; config E with the planned 256 x 256 tile; planned-tile parity
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 400 python -m pytest tests/test_hip_primitives.py -q -p no:cacheprovider -k "gemm" > $out/pytest_release.txt 2>&1; echo "pytest (release) rc=$?"; tail -n 2 $out/pytest_release.txt
( export FLAMINGO_FUSION_LIB=debug FF_XATTN_XCD_SPLIT=1; timeout 600 python -m pytest tests/test_hip_modules.py tests/test_hip_benchpath.py -q -p no:cacheprovider -k "resident or in_launch or block or benchmark" > $out/pytest_split.txt 2>&1; echo "pytest (xcd split) rc=$?"; tail -n 3 $out/pytest_split.txt; grep -E "^(FAILED|ERROR)" $out/pytest_split.txt | cut -c1-300 )
B="python bench.py --no-cpu-baseline --caption-tokens 0 --companions off --steps 20 --warmup 3 --profile-steps 0"
run() { name=$1; shift; timeout 400 $B "$@" 2> $out/$name.err > $out/$name.json; python - "$out/$name.json" "$name" <<'P'
import sys, json
try:
    d = next(json.loads(l) for l in reversed(open(sys.argv[1]).read().strip().splitlines()) if l.startswith('{'))
    print(sys.argv[2], d["value"], d["unit"], d["ms_per_step"], "ms/step", "loss", d["config"].get("loss_last"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
( export FLAMINGO_FUSION_LIB=debug
  FF_XATTN_XCD_SPLIT=0 run heads8_1
  FF_XATTN_XCD_SPLIT=1 run heads4_1
  FF_XATTN_XCD_SPLIT=0 run heads8_2
  FF_XATTN_XCD_SPLIT=1 run heads4_2 ) | tee $out/xattn_xcd_split_ab.txt
run E_release --config E --steps 6 --warmup 2 | tee $out/config_E.txt
