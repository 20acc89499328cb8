#!/bin/bash
# round 3, session 28: Abramowitz-Stegun erf in the bfloat16 GELU epilogues: parity, the epilogue's cost in isolation, the step
ulimit -c 0
tag=${1:-r3s28}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_primitives.py tests/test_hip_modules.py tests/test_hip_benchpath.py tests/test_hip_configs.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
for epi in "" act act_sqrelu; do ( export EPI=$epi; timeout 120 python tools/gemm_graph_bench.py 1024 5120 1280 0 0 2>&1 | grep TFLOP ) | tee -a $out/epi.txt; done
for epi in "" act_bwd act_bwd_sqrelu; do ( export EPI=$epi; timeout 120 python tools/gemm_graph_bench.py 1024 5120 1280 0 1 2>&1 | grep TFLOP ) | tee -a $out/epi.txt; done
for i in 1 2; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[fast erf]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])"
done
