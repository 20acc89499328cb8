#!/bin/bash
ulimit -c 0
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
A="--no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3 --graph piecewise --force-collectives"
run() { name=$1; timeout 300 python tools/sessions/r4/rccl_variants.py $A 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"; tail -2 $out/$name.err | cut -c1-200; }
VARIANT=host_paced_all run host_paced_all
VARIANT=host_paced_all COALESCE=1 run host_paced_all_coalesced
VARIANT=host_paced COALESCE=1 run host_paced_coalesced
VARIANT=host_paced_all COALESCE=1 run host_paced_all_coalesced
