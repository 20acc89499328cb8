#!/bin/bash
# round 3, session 17: register-staged producers in the producer / consumer GEMM kernel (FF_GEMM_RS, development build): correctness, then A/B
ulimit -c 0
tag=${1:-r3s17}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
( export FF_GEMM_RS=1; timeout 600 python -m pytest tests/test_hip_primitives.py tests/test_hip_benchpath.py -m gpu -q -p no:cacheprovider -x -k "gemm or config_B" 2>&1 | tail -3 | cut -c1-300 )
for v in "FF_GEMM_RS=0" "FF_GEMM_RS=1" "FF_GEMM_RS=0" "FF_GEMM_RS=1"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
( export FF_GEMM_RS=1; timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --caption-tokens 0 --profile-steps 3 --companions off --gemm-table $out/gemm_table_rs1.txt 2> /dev/null > $out/bench_rs1.json; head -24 $out/gemm_table_rs1.txt )
