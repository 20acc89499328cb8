#!/bin/bash
ulimit -c 0
tag=${1:-r3s5}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_modules.py tests/test_hip_benchpath.py tests/test_hip_configs.py tests/test_model_plumbing.py -m gpu -q -s -p no:cacheprovider > $out/pytest.txt 2>&1
echo "pytest rc=$?"; grep -E "^\[benchpath|passed|failed|^FAILED|^E  " $out/pytest.txt | cut -c1-300 | tail -n 25
python -m flamingo_mini_amd.build --debug > /dev/null 2>&1
export FLAMINGO_FUSION_LIB=debug
for v in "FF_XATTN_RES=0" "FF_XATTN_RES=1" "FF_XATTN_RES=0" "FF_XATTN_RES=1"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 32 --profile-steps 2 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); ar = d.get('attention_roofline', {})
print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'], {k: (v['avg_launch_us'], v['frac']) for k, v in ar.items() if 'xattn' in k}, 'caption', d['caption']['value'], d['caption']['ms_per_decode_step'])" )
done
