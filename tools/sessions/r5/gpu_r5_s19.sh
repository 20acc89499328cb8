#!/bin/bash
ulimit -c 0
# Round 5, session 19: phase 3 - the tests the aborted run of session 18 did not reach (full models, graphs with a 1-rank exchange, two ranks), the
# piecewise-capture test three times over (the watchdog abort), then the step with and without phase 3 (development build)
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_graph.py tests/test_model_plumbing.py tests/test_hip_two_ranks.py tests/test_hip_configs.py -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 4 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-300
for i in 1 2 3; do timeout 300 python -m pytest tests/test_hip_graph.py -q -p no:cacheprovider -k "piecewise or rccl" > $out/pytest_rccl_$i.txt 2>&1; echo "rccl tests run $i rc=$?"; tail -n 1 $out/pytest_rccl_$i.txt; done
B="python bench.py --no-cpu-baseline --caption-tokens 0 --companions off --steps 20 --warmup 3 --profile-steps 0"
run() { name=$1; shift; timeout 400 $B "$@" 2> $out/$name.err > $out/$name.json; python - "$out/$name.json" "$name" <<'P'
import sys, json
try:
    d = next(json.loads(l) for l in reversed(open(sys.argv[1]).read().strip().splitlines()) if l.startswith('{'))
    print(sys.argv[2], d["value"], d["unit"], d["ms_per_step"], "ms/step", "loss", d["config"].get("loss_first"), d["config"].get("loss_last"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
( export FLAMINGO_FUSION_LIB=debug
  FF_XATTN_LN3=0 run ln3_off_1
  FF_XATTN_LN3=1 run ln3_on_1
  FF_XATTN_LN3=0 run ln3_off_2
  FF_XATTN_LN3=1 run ln3_on_2
  FF_XATTN_LN3=1 run ln3_on_nodrop --lm-dropout 0
  FF_XATTN_LN3=0 run ln3_off_nodrop --lm-dropout 0 ) | tee $out/xattn_ln3_ab.txt
