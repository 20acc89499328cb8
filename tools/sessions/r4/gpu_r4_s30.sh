#!/bin/bash
ulimit -c 0
# layers per backward segment of the piecewise step (exchange granularity vs launch structure), 1-rank RCCL exchange
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3 --graph piecewise --force-collectives"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
c = d['config']
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step | host', c.get('piecewise_host_ms_per_step'))"; }
run seg4
run seg6 --segment-layers 6 --wgrad-group 6 --kv-group 6
run seg12 --segment-layers 12 --wgrad-group 12 --kv-group 12
run seg2 --segment-layers 2 --wgrad-group 2 --kv-group 2
run seg4
