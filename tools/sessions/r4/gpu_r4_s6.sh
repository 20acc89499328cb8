#!/bin/bash
# round 4, session 6: the data-gradient split-K slabs of the FFW up-projection summed inside the LayerNorm backward (no reduce launch): parity + same-box A/B
ulimit -c 0
tag=${1:-r4s6}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_modules.py tests/test_hip_benchpath.py tests/test_model_plumbing.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
export FLAMINGO_FUSION_LIB=debug
for v in "FF_FOLD_SPLITK_LN=0" "FF_FOLD_SPLITK_LN=1" "FF_FOLD_SPLITK_LN=0" "FF_FOLD_SPLITK_LN=1"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --backbone-tweaks on 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
