#!/bin/bash
ulimit -c 0
# Round 5, session 22: which Python line launches the stock fills / casts / cats of a step (torch.profiler with stacks, one eager step of config B)
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python tools/stock_kernel_attribution.py > $out/attribution.txt 2> $out/attribution.err; echo "rc=$?"; tail -n 4 $out/attribution.err | cut -c1-300; head -n 90 $out/attribution.txt | cut -c1-230
