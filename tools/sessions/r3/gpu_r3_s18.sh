#!/bin/bash
# round 3, session 18: weight-streaming decode GEMM (gemm_bf16_rows32_kernel) and the unfused decode path of the gated block: tests, caption A/B
ulimit -c 0
tag=${1:-r3s18}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_primitives.py tests/test_model_plumbing.py tests/test_hip_modules.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3 | cut -c1-400
export FLAMINGO_FUSION_LIB=debug
for v in "FF_GEMM_ROWS32=0 FF_XATTN_ROWS=0" "FF_GEMM_ROWS32=1 FF_XATTN_ROWS=0" "FF_GEMM_ROWS32=1 FF_XATTN_ROWS=1" "FF_GEMM_ROWS32=0 FF_XATTN_ROWS=0" "FF_GEMM_ROWS32=1 FF_XATTN_ROWS=1"; do
  ( export $v; timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['caption']; print('[$v]', c['value'], 'tok/s', c['ms_per_decode_step'], 'ms/token step')" )
done
