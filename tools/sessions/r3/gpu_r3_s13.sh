#!/bin/bash
ulimit -c 0
tag=${1:-r3s13}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
( export FF_GEMM_PC64=1; timeout 300 python -m pytest tests/test_hip_primitives.py -m gpu -q -p no:cacheprovider -k "gemm" 2>&1 | tail -2 )
for v in "FF_GEMM_PC64=0" "FF_GEMM_PC64=1" "FF_GEMM_PC64=0" "FF_GEMM_PC64=1"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
