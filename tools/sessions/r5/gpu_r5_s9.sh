#!/bin/bash
ulimit -c 0
# Round 5, session 9: the 256 x 256 tile with the five-unit ring (one and a half k-steps in flight): parity + the same isolated A/B as session 8;
# where the piecewise step idles with and without a 1-rank RCCL exchange (kernel traces -> tools/queue_gaps.py)
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_primitives.py -q -p no:cacheprovider -k "gemm" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-300
G="timeout 120 python tools/gemm_graph_bench.py"
( for shape in "4096 16384 4096 0 0" "4096 16384 4096 0 1" "4096 4096 16384 0 0" "4096 4096 16384 0 1" "8192 8192 8192 0 0"; do
    for t in 256128 256256; do $G $shape $t 2>/dev/null | tail -1; done
  done
  for shape in "4096 16384 4096 1 1" "16384 4096 4096 1 1"; do
    for t in 128002 256256; do $G $shape $t 2>/dev/null | tail -1; done
  done
  for t in 256128 256256; do EPI=act $G 4096 16384 4096 0 0 $t 2>/dev/null | tail -1; done
  for t in 256128 256256; do EPI=res $G 4096 4096 16384 0 0 $t 2>/dev/null | tail -1; done
) > $out/gemm_u16_ab.txt 2>&1
cat $out/gemm_u16_ab.txt
cd /tmp
B="python $R/bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 4 --warmup 2 --graph piecewise"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_plain -- $B > $out/trace_plain.json 2> $out/trace_plain.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_rccl -- $B --force-collectives > $out/trace_rccl.json 2> $out/trace_rccl.err
cd $R
python tools/queue_gaps.py $(find $out/trace_plain -name "*kernel_trace.csv" | head -1) $(find $out/trace_rccl -name "*kernel_trace.csv" | head -1) > $out/queue_gaps.txt 2>&1; cat $out/queue_gaps.txt | cut -c1-400
for a in plain rccl; do f=$(find $out/trace_$a -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && gzip -c $f > $out/trace_$a.kernel_trace.csv.gz; done
rm -rf $out/trace_plain $out/trace_rccl; du -sh $out
