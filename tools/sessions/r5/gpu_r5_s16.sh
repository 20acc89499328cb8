#!/bin/bash
ulimit -c 0
# Round 5, session 16: phase 3 of the resident fused cross-attention kernels - LN(y1) of the feed-forward inside the forward launch, the backward of LN(y)
# inside the backward launch (second bank of arrival counters): parity, then the step with and without it (development build, FF_XATTN_LN3)
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_modules.py tests/test_hip_benchpath.py tests/test_hip_graph.py tests/test_model_plumbing.py tests/test_hip_two_ranks.py -q -x -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 4 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-300
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
  FF_XATTN_LN3=0 run ln3_off_1
  FF_XATTN_LN3=1 run ln3_on_1
  FF_XATTN_LN3=0 run ln3_off_2
  FF_XATTN_LN3=1 run ln3_on_2 ) | tee $out/xattn_ln3_ab.txt
