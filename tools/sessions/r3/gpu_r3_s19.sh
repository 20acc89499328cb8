#!/bin/bash
# round 3, session 19: decode-shaped products in isolation (cold weights, graph replay); variants of the weight-streaming kernel
ulimit -c 0
tag=${1:-r3s19}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
for v in "FF_ROWS_NT=0 FF_ROWS_PAIR=2" "FF_ROWS_NT=0 FF_ROWS_PAIR=2 FF_ROWS_NW=4" "FF_ROWS_NT=0 FF_ROWS_PAIR=2 FF_ROWS_NW=16"; do
  echo "== $v"; ( export $v; timeout 300 python tools/decode_gemm_bench.py 3216 2>&1 | grep "tile" ) | tee -a $out/decode_gemm.txt
done
