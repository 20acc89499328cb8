#!/bin/bash
ulimit -c 0
# is the piecewise step with a 1-rank RCCL exchange launch-bound? host issue time per step + its split into graph launches / collectives / finish
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
c = d['config']
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', c.get('graph_mode'), c.get('collectives'), '| host issue', c.get('host_issue_ms_per_step'), 'ms/step', c.get('piecewise_host_ms_per_step'))"; }
run full --graph on
run piecewise --graph piecewise
run piecewise_rccl --graph piecewise --force-collectives
run full_rccl --graph on --force-collectives
run eager_rccl --graph off --force-collectives
run piecewise_rccl --graph piecewise --force-collectives
