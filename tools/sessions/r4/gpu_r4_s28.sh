#!/bin/bash
ulimit -c 0
# what of RCCL's / ProcessGroupNCCL's presence costs the remaining ~1.9 ms of the piecewise step at one rank? environment knobs
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3 --graph piecewise --force-collectives"
run() { name=$1; shift; timeout 300 env "$@" $B 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step')"; }
run default A=1
run no_async_error_handling TORCH_NCCL_ASYNC_ERROR_HANDLING=0
run no_monitoring TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_DUMP_ON_TIMEOUT=0
run avoid_record_streams TORCH_NCCL_AVOID_RECORD_STREAMS=1
run one_channel NCCL_MAX_NCHANNELS=1
run high_priority_stream TORCH_NCCL_HIGH_PRIORITY=1
run default A=1
