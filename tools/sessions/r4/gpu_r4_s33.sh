#!/bin/bash
ulimit -c 0
# where kernel arguments live (HIP_FORCE_DEV_KERNARG) for a step that is ~3000 short launches
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
run() { name=$1; shift; timeout 200 env "$@" $B 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"; }
run default A=1
run dev_kernarg_1 HIP_FORCE_DEV_KERNARG=1
run dev_kernarg_0 HIP_FORCE_DEV_KERNARG=0
