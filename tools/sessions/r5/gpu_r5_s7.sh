#!/bin/bash
ulimit -c 0
# Round 5, session 7: the 256 x 128 tile with an M-major A operand (weight gradients of config E): parity, A/B against 128 x 128 on the weight-gradient
# shapes, config E end to end with its per-shape GEMM table, hipBLASLt beside every config-E shape
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 400 python -m pytest tests/test_hip_primitives.py -q -p no:cacheprovider -k "gemm" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-200
( for shape in "4096 16384 4096 1 1" "16384 4096 4096 1 1" "4096 16384 4096 1 0" "4096 1024 2048 1 1" "1280 5120 1024 1 1" "5120 1280 1024 1 1"; do
    for t in 128002 256128; do timeout 120 python tools/gemm_graph_bench.py $shape $t 2>/dev/null | tail -1; done
  done ) > $out/gemm_t256_a1_ab.txt 2>&1
cat $out/gemm_t256_a1_ab.txt
timeout 600 python bench.py --config E --no-cpu-baseline --caption-tokens 0 --companions off --steps 6 --warmup 2 --gemm-table $out/gemm_table_E.txt > $out/bench_config_E.json 2> $out/bench_config_E.err; echo "bench E rc=$?"
python - $out/bench_config_E.json <<'P'
import sys, json
try:
    d = next(json.loads(l) for l in reversed(open(sys.argv[1]).read().strip().splitlines()) if l.startswith('{'))
    print("config E:", d["value"], d["unit"], d["ms_per_step"], "ms/step; roofline", d.get("roofline"), "; all fusion GEMMs", d.get("all_fusion_gemms"))
except Exception as e:
    print("config E: no line", e)
P
head -3 $out/gemm_table_E.txt
timeout 500 python tools/gemm_yardstick.py $out/gemm_table_E.txt > $out/gemm_yardstick_E.txt 2> $out/gemm_yardstick_E.err; echo "yardstick rc=$?"; cat $out/gemm_yardstick_E.txt | cut -c1-200 | head -50
