#!/bin/bash
ulimit -c 0
tag=${1:-r3s11}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 300 python -m pytest tests/test_hip_graph.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for o in fused sharded fused sharded; do
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --optimizer $o 2> $out/err_$o.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$o]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_first'], d['config']['loss_last'], d['config']['hip_graph'])"
done
tail -3 $out/err_sharded.txt | cut -c1-300
