#!/bin/bash
ulimit -c 0
# Round 5, session 5: the 256 x 128 tile (16-wave workgroups) against the 128 x 128 one (two 8-wave workgroups per CU) on config E's feed-forward products;
# the decode test that was red; the 8-GPU session script rehearsed on one GPU
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 300 python -m pytest tests/test_hip_primitives.py tests/test_hip_modules.py -q -p no:cacheprovider -k "gemm or decode_shaped" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-200
( for shape in "4096 16384 4096 0 0" "4096 16384 4096 0 1" "4096 4096 16384 0 0" "4096 4096 16384 0 1"; do
    for t in 128002 256128; do python tools/gemm_graph_bench.py $shape $t 2>/dev/null | tail -1; done
    ( export FLAMINGO_FUSION_LIB=debug FF_GEMM_NPW256=4; echo "[four DMA waves] $(python tools/gemm_graph_bench.py $shape 256128 2>/dev/null | tail -1)" )
  done
  for t in 128002 256128; do EPI=act python tools/gemm_graph_bench.py 4096 16384 4096 0 0 $t 2>/dev/null | tail -1; done
  for t in 128002 256128; do EPI=act_bwd python tools/gemm_graph_bench.py 4096 16384 4096 0 1 $t 2>/dev/null | tail -1; done
  for t in 128002 256128; do EPI=res python tools/gemm_graph_bench.py 4096 4096 16384 0 0 $t 2>/dev/null | tail -1; done
  for shape in "8192 8192 8192 0 0" "4096 2048 8192 0 0"; do for t in 128002 256128; do python tools/gemm_graph_bench.py $shape $t 2>/dev/null | tail -1; done; done
) > $out/gemm_t256_ab.txt 2>&1
cat $out/gemm_t256_ab.txt
timeout 600 bash tools/sessions/r5/scale8.sh $out/scale8 --rehearsal > $out/scale8_rehearsal.txt 2>&1; echo "scale8 rehearsal rc=$?"; tail -n 12 $out/scale8_rehearsal.txt
