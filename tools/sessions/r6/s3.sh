#!/bin/bash
ulimit -c 0
# r6 session 3: attribution of the feed-forward products' epilogues (second half) and the per-phase timeline of the
# fused cross-attention launches (item 5) - numbers for DESIGN.md section 8, no product change
out=gpurun_out/r6s3; mkdir -p $out; export TMPDIR=/tmp
{
echo "# 1024 x 5120 x 1280 in isolation (cold weights, graph replay, us per launch incl. the gap): what the epilogue costs, by what it does"
for bl in 0 1; do
  for epi in "" act_sqrelu act act_bwd_sqrelu act_bwd; do
    EPI=$epi python tools/gemm_graph_bench.py 1024 5120 1280 0 $bl 2>&1 | tail -1
  done
  EPI=act_bwd COLD_H=1 python tools/gemm_graph_bench.py 1024 5120 1280 0 $bl 2>&1 | tail -1
  EPI=act_bwd_sqrelu COLD_H=1 python tools/gemm_graph_bench.py 1024 5120 1280 0 $bl 2>&1 | tail -1
done
echo "# 1024 x 1280 x 5120 (split-K 4), plain and with the gated residual"
for bl in 0 1; do
  python tools/gemm_graph_bench.py 1024 1280 5120 0 $bl 2>&1 | tail -1
  EPI=res python tools/gemm_graph_bench.py 1024 1280 5120 0 $bl 2>&1 | tail -1
done
} > $out/ffw_epilogue_attribution.txt 2>&1
cat $out/ffw_epilogue_attribution.txt
bash tools/build_timeline.sh > $out/build_timeline.log 2>&1; tail -1 $out/build_timeline.log
python tools/xattn_res_timeline.py > $out/xattn_res_timeline.txt 2>&1; cat $out/xattn_res_timeline.txt | cut -c1-400
