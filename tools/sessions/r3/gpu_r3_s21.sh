#!/bin/bash
# round 3, session 21: grouped weight gradients on a side stream (FF_WGRAD_STREAM=1) beside the data-gradient chain: parity, then A/B of the replayed step
ulimit -c 0
tag=${1:-r3s21}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
( export FF_WGRAD_STREAM=1; timeout 600 python -m pytest tests/test_hip_benchpath.py tests/test_hip_graph.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2 | cut -c1-300 )
for v in "FF_WGRAD_STREAM=0" "FF_WGRAD_STREAM=1" "FF_WGRAD_STREAM=0" "FF_WGRAD_STREAM=1" "FF_WGRAD_STREAM=1 FF_WGRAD_GROUP=4" "FF_WGRAD_STREAM=1 FF_WGRAD_GROUP=6"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
