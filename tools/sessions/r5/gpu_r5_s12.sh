#!/bin/bash
ulimit -c 0
# Round 5, session 12: 256 x 256 tile with the two wave groups half a k-step apart (256256) vs all waves in step (256257) vs 256 x 128; config E
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_primitives.py -q -p no:cacheprovider -k "gemm" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-300
G="timeout 120 python tools/gemm_graph_bench.py"
( for shape in "4096 16384 4096 0 0" "4096 4096 16384 0 0" "8192 8192 8192 0 0" "4096 8192 2048 0 0"; do
    for t in 256128 256257 256256; do $G $shape $t 2>/dev/null | tail -1; done
  done
  for t in 256128 256257 256256; do EPI=act $G 4096 16384 4096 0 0 $t 2>/dev/null | tail -1; done
  for t in 256128 256257 256256; do EPI=res $G 4096 4096 16384 0 0 $t 2>/dev/null | tail -1; done
) > $out/gemm_u16_groups_ab.txt 2>&1
cat $out/gemm_u16_groups_ab.txt
timeout 500 python bench.py --config E --no-cpu-baseline --caption-tokens 0 --companions off --steps 6 --warmup 2 --gemm-table $out/gemm_table_E.txt > $out/bench_config_E.json 2> $out/bench_config_E.err
python - $out/bench_config_E.json <<'P'
import sys, json
try:
    d = next(json.loads(l) for l in reversed(open(sys.argv[1]).read().strip().splitlines()) if l.startswith('{'))
    print("config E:", d["value"], d["unit"], d["ms_per_step"], "ms/step; roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
except Exception as e:
    print("config E: no line", e)
P
head -8 $out/gemm_table_E.txt
