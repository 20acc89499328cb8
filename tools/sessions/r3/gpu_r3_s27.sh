#!/bin/bash
# round 3, session 27: what do the epilogues of the FFW launches cost? (plain / gelu + aux_out / gelu' from aux_in + gate / the squared-ReLU twins / gate + residual)
ulimit -c 0
tag=${1:-r3s27}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
for epi in "" act act_sqrelu res; do ( export EPI=$epi; timeout 120 python tools/gemm_graph_bench.py 1024 5120 1280 0 0 2>&1 | grep TFLOP ) | tee -a $out/epi.txt; done
for epi in "" act_bwd act_bwd_sqrelu; do ( export EPI=$epi; timeout 120 python tools/gemm_graph_bench.py 1024 5120 1280 0 1 2>&1 | grep TFLOP ) | tee -a $out/epi.txt; done
