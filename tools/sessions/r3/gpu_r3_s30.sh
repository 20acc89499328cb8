#!/bin/bash
# round 3, session 30: the 12-wave 128 x 160 workgroup for split-K launches as well (FF_GEMM_NPW_SPLIT=8) or only for the launches with an epilogue (4)
ulimit -c 0
tag=${1:-r3s30}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
for v in "FF_GEMM_NPW_SPLIT=8" "FF_GEMM_NPW_SPLIT=4" "FF_GEMM_NPW_SPLIT=8" "FF_GEMM_NPW_SPLIT=4"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
