#!/bin/bash
# round 4, session 3: (a) where a decode-kernel workgroup's time goes (stand-alone probe with phase timestamps, several geometries, the failing tiny case),
# (b) the 128 x 160 GEMM tile with eight MFMA waves against four (isolated launches, cold weights, graph replay) + its parity tests
ulimit -c 0
tag=${1:-r4s3}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
P=tools/experiments/_bin/decode_probe
for a in "32 5120 1280 20 1 1" "32 5120 1280 20 1 0" "32 5120 1280 16 1 1" "32 5120 1280 32 1 1" "32 1280 5120 20 4 0" "32 1280 5120 20 2 0" "32 1280 1280 20 1 0" \
         "15 256 256 4 1 1" "15 256 256 4 1 0" "15 256 256 16 1 1" "32 8192 2048 32 1 1" "32 2048 8192 32 4 0" "7 3072 768 12 1 1"; do
  echo "== $a"; timeout 60 $P $a 2>&1 | tail -4
done > $out/probe.txt 2>&1
cat $out/probe.txt
for t in 128160 128168 128160 128168; do python tools/gemm_graph_bench.py 1024 5120 1280 0 0 $t 2>/dev/null | tail -1; done
for t in 128160 128168; do EPI=act python tools/gemm_graph_bench.py 1024 5120 1280 0 0 $t 2>/dev/null | tail -1; done
for t in 128160 128168; do EPI=act_bwd python tools/gemm_graph_bench.py 1024 5120 1280 0 1 $t 2>/dev/null | tail -1; done
for t in 128160 128168; do python tools/gemm_graph_bench.py 1024 1280 5120 0 0 $t 2>/dev/null | tail -1; done
for t in 128160 128168; do python tools/gemm_graph_bench.py 1024 1280 5120 0 1 $t 2>/dev/null | tail -1; done
for t in 128160 128168; do python tools/gemm_graph_bench.py 2048 4096 1024 0 0 $t 2>/dev/null | tail -1; done
timeout 300 python -m pytest tests/test_hip_primitives.py -m gpu -q -p no:cacheprovider -k "tile" 2>&1 | tail -3
