#!/bin/bash
ulimit -c 0
# r6 session 13: static issue priority for the MFMA waves of gemm_bf16_pc_kernel (s_setprio 1 / 3 before the k-loop, 0 before the epilogue) vs the
# shipped kernel, on the feed-forward and weight-gradient shapes; each line = cold weights, graph replay
out=gpurun_out/r6s13; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
for rep in 1 2; do
for shape in "1024 5120 1280 0 0" "1024 5120 1280 0 1" "1024 1280 5120 0 0" "5120 5120 1024 1 1" "2048 4096 1024 0 0" "10272 1024 512 0 1"; do
  python tools/gemm_graph_bench.py $shape 2>&1 | tail -1
  for pr in 1 3; do FLAMINGO_FUSION_LIB=$R/tools/_dbg/libflamingo_fusion_prio$pr.so python tools/gemm_graph_bench.py $shape 2>&1 | tail -1 | sed "s/\[/[setprio $pr /"; done
done
done
} > $out/consumer_prio_ab.txt 2>&1
cat $out/consumer_prio_ab.txt
