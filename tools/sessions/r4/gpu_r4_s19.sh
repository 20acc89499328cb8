#!/bin/bash
ulimit -c 0
# host-paced vs stream-ordered collectives of the piecewise step (product code), parity of both against eager / full capture
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_graph.py -m gpu -q -p no:cacheprovider -x > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
c = d['config']
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', c.get('graph_mode'), c.get('collectives'), c.get('collective_pace'), '| host', c.get('piecewise_host_ms_per_step'))"; }
run piecewise_rccl_host --graph piecewise --force-collectives --pace host
run piecewise_rccl_stream --graph piecewise --force-collectives --pace stream
run piecewise_rccl_host --graph piecewise --force-collectives --pace host
run auto_rccl --force-collectives
run piecewise --graph piecewise
run full --graph on
