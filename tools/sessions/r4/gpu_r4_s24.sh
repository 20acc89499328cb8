#!/bin/bash
ulimit -c 0
# one collective per backward segment (functional.GradArena): parity, then the piecewise step with and without it
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_two_ranks.py -m gpu -q -p no:cacheprovider -x > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt | cut -c1-300
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
c = d['config']
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', c.get('graph_mode'), 'overlapped optimizer', c.get('overlapped_optimizer'), 'loss', c.get('loss'), '| host', c.get('piecewise_host_ms_per_step'))"; tail -1 $out/$name.err | cut -c1-200; }
run arena_on --graph piecewise --force-collectives
run arena_off --graph piecewise --force-collectives --segment-arena off
run arena_on --graph piecewise --force-collectives
run arena_on_stream_paced --graph piecewise --force-collectives --pace stream --overlap-optimizer off
run full --graph on
