#!/bin/bash
ulimit -c 0
tag=${1:-r3s8}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
( export FLAMINGO_FUSION_LIB=debug FF_XATTN_RES=0; python tools/res_compare.py run /tmp/old.pt ) 2> /dev/null
( export FLAMINGO_FUSION_LIB=debug FF_XATTN_RES=1; python tools/res_compare.py run /tmp/new.pt ) 2> /dev/null
python tools/res_compare.py diff /tmp/old.pt /tmp/new.pt
timeout 600 python -m pytest tests/test_hip_modules.py tests/test_hip_benchpath.py tests/test_hip_configs.py -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^E  " $out/pytest.txt | cut -c1-200 | tail -n 8
for v in "FF_WGRAD_GROUP=12" "FF_WGRAD_GROUP=12"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v + LN hoist]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
