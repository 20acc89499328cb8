#!/bin/bash
# round 3, session 1: the new parity tests (benchmark path, backbones, interchange, optimizer resume), the default bench line with its
# companion runs, the loss trajectories, the counter list
ulimit -c 0
tag=${1:-r3s1}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 700 python -m pytest tests/test_hip_benchpath.py tests/test_hip_backbones.py tests/test_checkpoint_interchange.py tests/test_hip_optim.py -m gpu -q -s -p no:cacheprovider > $out/pytest_new.txt 2>&1
echo "pytest rc=$?"; grep -E "benchpath|backbones|passed|failed|FAILED|Error" $out/pytest_new.txt | cut -c1-400 | tail -n 40
timeout 600 python bench.py --gemm-table $out/gemm_table.txt > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-1500 $out/bench_default.json
timeout 600 python tools/loss_trajectory.py --steps 20 --modes on:on,on:off,off:on --dropouts default,0 > $out/loss_trajectory.jsonl 2> $out/loss_trajectory.err; cat $out/loss_trajectory.jsonl | cut -c1-400
(rocprofv3 -L 2>/dev/null | grep -iE "TCC_EA0_RDREQ|TCC_EA0_WRREQ|MALL|TCC_HIT|TCC_MISS|DRAM|HBM" | head -60) > $out/counters.txt 2>&1; wc -l $out/counters.txt
