#!/bin/bash
# round 3, session 29: v_sqrt / v_rcp in the bfloat16 AdamW kernels: optimizer tests, the step, the kernel's duration (rocprofv3 --stats)
ulimit -c 0
tag=${1:-r3s29}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_optim.py tests/test_hip_graph.py tests/test_optim_state.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2 | cut -c1-300
for i in 1 2; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[fast adamw]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $R/bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 5 --warmup 2 > /dev/null 2> $out/prof.err
grep -h "adamw_kernel" $(find $out/prof -name "*kernel_stats.csv") | cut -c1-160
rm -rf $out/prof
