#!/bin/bash
ulimit -c 0
# the short-K projection launches (1024x1280x512 to_out, 2048x512x1024 / 2048x1024x512 resampler) on other tiles than the planner's 64x64
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
for shape in "1024 1280 512 0 0" "1024 1280 512 0 1" "2048 512 1024 0 0" "2048 1024 512 0 1"; do
  for t in 64002 64 6412 128002 3264; do
    for epi in "" res; do EPI=$epi timeout 120 python tools/gemm_graph_bench.py $shape $t 2>/dev/null | tail -1; done
  done
done
