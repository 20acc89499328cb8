#!/bin/bash
ulimit -c 0
tag=${1:-r3s12}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
for v in "FF_GEMM_STAGES=0" "FF_GEMM_STAGES=3" "FF_GEMM_STAGES=4" "FF_GEMM_STAGES=0" "FF_GEMM_STAGES=3"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
unset FLAMINGO_FUSION_LIB
for c in A C D E; do
  timeout 600 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --caption-tokens 0 --companions off > $out/bench_config_$c.json 2> $out/bench_config_$c.err
  python -c "
import json
d = json.loads(open('$out/bench_config_$c.json').read().strip().splitlines()[-1]); r = d['roofline']
print('config $c', d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'graph', d['config']['hip_graph'], r['kernel'], r['frac'], r['all_fusion_gemms']['tflops'])"
done
