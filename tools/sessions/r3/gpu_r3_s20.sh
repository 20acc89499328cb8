#!/bin/bash
# round 3, session 20: the weight-streaming kernel restricted to short-K decode products: tests, isolated timings, caption A/B
ulimit -c 0
tag=${1:-r3s20}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_primitives.py tests/test_model_plumbing.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2 | cut -c1-400
timeout 300 python tools/decode_gemm_bench.py 0 3216 3264 2>&1 | grep tile | tee $out/decode_gemm.txt
export FLAMINGO_FUSION_LIB=debug
for v in "FF_GEMM_ROWS32=0" "FF_GEMM_ROWS32=1" "FF_GEMM_ROWS32=0" "FF_GEMM_ROWS32=1"; do
  ( export $v; timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['caption']; print('[$v]', c['value'], 'tok/s', c['ms_per_decode_step'], 'ms/token step')" )
done
