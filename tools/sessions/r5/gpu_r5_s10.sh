#!/bin/bash
ulimit -c 0
# Round 5, session 10: 256 x 256 tile, pieces spread over the k-step (256256) vs requested together (256257) vs the planned 256 x 128; one-rank RCCL
# exchange with the mean taken as an in-place SUM (no pre-multiplied-sum kernel): the launch modes of profiles/r04_launch_modes_one_rank_rccl.txt again
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_primitives.py tests/test_hip_graph.py -q -p no:cacheprovider -k "gemm or rccl or piecewise" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-300
G="timeout 120 python tools/gemm_graph_bench.py"
( for shape in "4096 16384 4096 0 0" "4096 4096 16384 0 0" "8192 8192 8192 0 0" "16384 4096 4096 1 0"; do
    for t in 256128 256257 256256; do $G $shape $t 2>/dev/null | tail -1; done
  done
  for t in 256128 256257 256256; do EPI=act $G 4096 16384 4096 0 0 $t 2>/dev/null | tail -1; done
  for t in 256128 256257 256256; do EPI=res $G 4096 4096 16384 0 0 $t 2>/dev/null | tail -1; done
) > $out/gemm_u16_spread_ab.txt 2>&1
cat $out/gemm_u16_spread_ab.txt
B2="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
( for g in "on" "piecewise" "piecewise --pace stream --overlap-optimizer off"; do timeout 300 $B2 --graph $g --force-collectives 2> /dev/null | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
print('graph=$g, gradient exchange through a 1-rank RCCL group (mean as an in-place SUM):', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', d['config']['graph_mode'], 'pace', d['config'].get('collective_pace'), 'overlapped optimizer', d['config'].get('overlapped_optimizer'))
"; done
  for g in "on" "piecewise"; do timeout 300 $B2 --graph $g 2> /dev/null | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
print('graph=$g, no collectives:', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', d['config']['graph_mode'])"; done ) > $out/launch_modes_one_rank_rccl_sum.txt
cat $out/launch_modes_one_rank_rccl_sum.txt
