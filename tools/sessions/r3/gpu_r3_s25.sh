#!/bin/bash
# round 3, session 25: fragment prefetch in the consumers of the 128 x 160 producer / consumer kernel (FF_GEMM_PF=1, development build)
ulimit -c 0
tag=${1:-r3s25}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
( export FF_GEMM_PF=1; timeout 600 python -m pytest tests/test_hip_primitives.py tests/test_hip_benchpath.py -m gpu -q -p no:cacheprovider -x -k "gemm or config_B" 2>&1 | tail -2 | cut -c1-300 )
( export FF_GEMM_PF=1 FF_GEMM_NPW=4 FF_GEMM_STAGES=4; timeout 600 python -m pytest tests/test_hip_primitives.py -m gpu -q -p no:cacheprovider -x -k "gemm" 2>&1 | tail -1 | cut -c1-300 )
for v in "FF_GEMM_PF=1 FF_GEMM_NPW=4 FF_GEMM_STAGES=4" "FF_GEMM_PF=1 FF_GEMM_NPW=8 FF_GEMM_STAGES=4"; do for K in 1280 5120; do for bl in 0 1; do
  ( export $v; timeout 120 python tools/gemm_graph_bench.py 1024 5120 $K 0 $bl 128160 2>&1 | grep TFLOP ) | tee -a $out/pf.txt
done; done; done
for v in "FF_GEMM_PF=0" "FF_GEMM_PF=1" "FF_GEMM_PF=0" "FF_GEMM_PF=1" "FF_GEMM_PF=1 FF_GEMM_NPW=4 FF_GEMM_STAGES=4"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
