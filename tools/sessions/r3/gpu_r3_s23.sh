#!/bin/bash
# round 3, session 23: where does a k-step of the 128 x 160 producer / consumer kernel go?  FF_GEMM_PCMODE (development build): 0 normal,
# 1 no fragment reads / MFMA (fill path + barriers only), 2 no DMA (consumers only), 3 fragment reads without MFMA, 4 MFMA without reads
ulimit -c 0
tag=${1:-r3s23}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
for npw in 8 4; do for st in 4 3; do for mode in 0 1 2 3 4; do
  ( export FF_GEMM_NPW=$npw FF_GEMM_STAGES=$st FF_GEMM_PCMODE=$mode; timeout 120 python tools/gemm_graph_bench.py 1024 5120 1280 0 0 128160 2>&1 | grep TFLOP ) | tee -a $out/pcmode.txt
done; done; done
for mode in 0 1 2 3 4; do
  ( export FF_GEMM_PCMODE=$mode; timeout 120 python tools/gemm_graph_bench.py 1024 5120 1280 0 1 128160 2>&1 | grep TFLOP ) | tee -a $out/pcmode.txt
  ( export FF_GEMM_PCMODE=$mode COLD_A=1; timeout 120 python tools/gemm_graph_bench.py 1024 5120 1280 0 0 128160 2>&1 | grep TFLOP ) | tee -a $out/pcmode.txt
done
