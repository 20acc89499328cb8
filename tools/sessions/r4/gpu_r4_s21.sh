#!/bin/bash
ulimit -c 0
# bounded-grid ("polite") AdamW beside the backward segments: FF_ADAMW_MAX_BLOCKS in the development build
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 300 python -m pytest tests/test_hip_optim.py -m gpu -q -p no:cacheprovider -x > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 2 $out/pytest.txt
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
c = d['config']
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', c.get('graph_mode'), 'overlapped optimizer', c.get('overlapped_optimizer'), 'loss', c.get('loss'))"; }
run full --graph on
run piecewise_overlap --graph piecewise --overlap-optimizer on
export FLAMINGO_FUSION_LIB=debug
for mb in 0 1024 512 256 128; do
  FF_ADAMW_MAX_BLOCKS=$mb run piecewise_overlap_maxblocks_$mb --graph piecewise --overlap-optimizer on
done
FF_ADAMW_MAX_BLOCKS=512 run full_maxblocks_512 --graph on
FF_ADAMW_MAX_BLOCKS=2048 run full_maxblocks_2048 --graph on
