#!/bin/bash
ulimit -c 0
# AdamW as per-segment sub-graphs beside the backward (PiecewiseGraphedTrainStep(overlap_optimizer=True)): parity, then the step with and without it
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_graph.py tests/test_hip_optim.py -m gpu -q -p no:cacheprovider -x > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 5 $out/pytest.txt
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
c = d['config']
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', c.get('graph_mode'), 'collectives', c.get('collectives'), 'overlapped optimizer', c.get('overlapped_optimizer'), 'loss', c.get('loss'), '| host', c.get('piecewise_host_ms_per_step'))"; tail -1 $out/$name.err | cut -c1-300; }
run full --graph on
run piecewise --graph piecewise
run piecewise_overlap --graph piecewise --overlap-optimizer on
run piecewise_rccl --graph piecewise --force-collectives
run piecewise_rccl_overlap --graph piecewise --force-collectives --overlap-optimizer on
run piecewise_overlap --graph piecewise --overlap-optimizer on
run full --graph on
