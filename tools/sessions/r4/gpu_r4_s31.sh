#!/bin/bash
ulimit -c 0
# one GPU, no collectives: whole-step graph vs piecewise replay with the optimizer beside the backward, by segment size
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
c = d['config']
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', c.get('graph_mode'), 'overlapped optimizer', c.get('overlapped_optimizer'))"; }
run full --graph on
run pw4_overlap --graph piecewise --overlap-optimizer on
run pw12_overlap --graph piecewise --overlap-optimizer on --segment-layers 12
run pw36_overlap --graph piecewise --overlap-optimizer on --segment-layers 36
run full --graph on
