#!/bin/bash
ulimit -c 0
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
A="--no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3 --graph piecewise --force-collectives"
for v in ${VARIANTS:-default per_bucket events_only same_stream default}; do
  VARIANT=$v timeout 300 python tools/sessions/r4/rccl_variants.py $A 2> $out/$v.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
c = d['config']
print('$v:', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', c.get('graph_mode'), c.get('collectives'), '| host', c.get('piecewise_host_ms_per_step'))"
done
timeout 300 python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3 --graph piecewise 2>/dev/null | tail -1 | cut -c1-200
