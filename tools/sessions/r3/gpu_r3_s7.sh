#!/bin/bash
ulimit -c 0
tag=${1:-r3s7}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_modules.py -m gpu -q -p no:cacheprovider -k "resident or hoisted or deferred" > $out/pytest.txt 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^E  " $out/pytest.txt | cut -c1-300 | tail -n 8
for v in "FF_WGRAD_GROUP=4" "FF_WGRAD_GROUP=6" "FF_WGRAD_GROUP=12" "FF_WGRAD_GROUP=9" "FF_WGRAD_GROUP=4" "FF_WGRAD_GROUP=6" "FF_WGRAD_GROUP=12"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 2 --companions off --gemm-table $out/gemm_$v.txt 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); ar = d.get('attention_roofline', {})
print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['all_fusion_gemms'], {k: (v['avg_launch_us'], v['frac']) for k, v in ar.items() if 'xattn' in k})" )
done
head -14 "$out/gemm_FF_WGRAD_GROUP=6.txt"
