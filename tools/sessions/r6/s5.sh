#!/bin/bash
ulimit -c 0
# r6 session 5: is the GEMM family bound by the CU's request RATE or by operand LATENCY?  The same launches with cold weights (a different
# buffer per launch, as in the model) and with warm ones (one buffer: L2 / Infinity Cache resident), at several sizes.
out=gpurun_out/r6s5; mkdir -p $out; export TMPDIR=/tmp
{
echo "# cold (default: 30+ weight buffers) vs warm (UNIQUE=1: one weight buffer in all launches of the graph; UNIQUE=4: four in rotation) weights; us per launch in a graph of back-to-back launches"
for shape in "1024 5120 1280 0 0" "1024 5120 1280 0 1" "1024 2560 1280 0 0" "1024 1280 1280 0 0" "2048 4096 1024 0 0" "1024 1280 5120 0 0"; do
  python tools/gemm_graph_bench.py $shape 2>&1 | tail -1
  UNIQUE=4 python tools/gemm_graph_bench.py $shape 2>&1 | tail -1
  UNIQUE=1 python tools/gemm_graph_bench.py $shape 2>&1 | tail -1
done
echo "# weight gradients (M-major operands), one problem: 1280 x 5120 x 1024"
python tools/gemm_graph_bench.py 1280 5120 1024 1 1 2>&1 | tail -1
UNIQUE=1 python tools/gemm_graph_bench.py 1280 5120 1024 1 1 2>&1 | tail -1
UNIQUE=1 python tools/gemm_graph_bench.py 5120 1280 1024 1 1 2>&1 | tail -1
} > $out/cold_vs_warm.txt 2>&1
cat $out/cold_vs_warm.txt
