#!/bin/bash
# round 3, session 24: K sweep of the 128 x 160 producer / consumer kernel (development build, four DMA waves): slope = time per k-step,
# intercept = fixed cost; normal, fill only (PCMODE 1), consumers only (2), reads only (3), MFMA only (4)
ulimit -c 0
tag=${1:-r3s24}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug FF_GEMM_NPW=4 FF_GEMM_STAGES=3
for mode in 0 1 2 3 4; do for K in 128 640 1280 2560 5120; do
  ( export FF_GEMM_PCMODE=$mode; timeout 120 python tools/gemm_graph_bench.py 1024 5120 $K 0 0 128160 2>&1 | grep TFLOP ) | tee -a $out/ksweep.txt
done; done
