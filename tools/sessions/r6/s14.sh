#!/bin/bash
ulimit -c 0
out=gpurun_out/r6s14; mkdir -p $out; export TMPDIR=/tmp
python tools/sessions/r6/s14.py > $out/parity.txt 2>&1; echo "parity rc=$?"; tail -20 $out/parity.txt
{
for rep in 1 2; do
for shape in "1024 5120 1280 0 0" "1024 5120 1280 0 1" "1024 1280 5120 0 0" "1024 1280 5120 0 1"; do
  for tile in 128160 64160; do
    python tools/gemm_graph_bench.py $shape $tile 2>&1 | tail -1
  done
done
for epi in act act_bwd; do for tile in 128160 64160; do EPI=$epi COLD_H=1 python tools/gemm_graph_bench.py 1024 5120 1280 0 $([ $epi = act ] && echo 0 || echo 1) $tile 2>&1 | tail -1; done; done
done
} > $out/paired_ab.txt 2>&1
cat $out/paired_ab.txt
