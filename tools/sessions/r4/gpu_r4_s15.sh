#!/bin/bash
ulimit -c 0
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
A="--no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3 --graph piecewise --force-collectives"
run() { name=$1; timeout 300 python tools/sessions/r4/rccl_variants.py $A 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"; }
VARIANT=default run default
VARIANT=default SIDE_PRIORITY=-1 run default_hiprio
VARIANT=ready_wait SIDE_PRIORITY=-1 run ready_wait_hiprio
VARIANT=default GPU_MAX_HW_QUEUES=8 run default_8queues
VARIANT=ready_wait GPU_MAX_HW_QUEUES=8 run ready_wait_8queues
VARIANT=default GPU_MAX_HW_QUEUES=2 run default_2queues
VARIANT=default GPU_MAX_HW_QUEUES=8 SIDE_PRIORITY=-1 run default_8queues_hiprio
VARIANT=default run default
timeout 300 python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3 --graph piecewise 2>/dev/null | tail -1 | cut -c1-200
