#!/bin/bash
# round 3, session 31: nontemporal accesses for the saved epilogue outputs (aux_out written in forward, aux_in read once in backward): FF_EPI_NT=0/1
ulimit -c 0
tag=${1:-r3s31}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
( export FF_EPI_NT=1; timeout 300 python -m pytest tests/test_hip_primitives.py -m gpu -q -p no:cacheprovider -x -k "epilogue or balanced or decode" 2>&1 | tail -1 | cut -c1-200 )
for v in "FF_EPI_NT=0" "FF_EPI_NT=1" "FF_EPI_NT=0" "FF_EPI_NT=1"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
