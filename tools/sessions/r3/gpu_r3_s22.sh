#!/bin/bash
# round 3, session 22: eight DMA waves (12-wave workgroups) and a 4-deep ring in the 128 x 160 producer / consumer kernel - together, which round 2 never tried
ulimit -c 0
tag=${1:-r3s22}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
( export FF_GEMM_NPW=8 FF_GEMM_STAGES=4; timeout 600 python -m pytest tests/test_hip_primitives.py tests/test_hip_benchpath.py -m gpu -q -p no:cacheprovider -x -k "gemm or config_B" 2>&1 | tail -2 | cut -c1-300 )
( export FF_GEMM_NPW=8; timeout 600 python -m pytest tests/test_hip_primitives.py -m gpu -q -p no:cacheprovider -x -k "gemm" 2>&1 | tail -1 | cut -c1-300 )
for v in "FF_GEMM_NPW=4" "FF_GEMM_NPW=8" "FF_GEMM_NPW=4 FF_GEMM_STAGES=4" "FF_GEMM_NPW=8 FF_GEMM_STAGES=4" "FF_GEMM_NPW=4" "FF_GEMM_NPW=8 FF_GEMM_STAGES=4"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
