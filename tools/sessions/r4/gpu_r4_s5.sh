#!/bin/bash
# round 4, session 5: decode kernels (tightened LayerNorm), resident forward kernel with one-pass / independent-accumulator statistics: parity + timings
ulimit -c 0
tag=${1:-r4s5}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
P=tools/experiments/_bin/decode_probe
for a in "32 5120 1280 20 1 1" "32 8192 2048 32 1 1" "7 3072 768 12 1 1"; do echo "== $a"; timeout 60 $P $a 2>&1 | tail -3; done
timeout 600 python -m pytest tests/test_hip_modules.py tests/test_hip_benchpath.py tests/test_hip_primitives.py tests/test_hip_configs.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5
python tools/decode_chain_bench.py 2>&1 | tail -1
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --companions off --gemm-table $out/gemm_table.txt > $out/bench.json 2> $out/bench.err
python - <<P
import json
d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print(d["value"], "img/s", d["ms_per_step"], "ms/step; caption", d["caption"]["value"], d["caption"]["ms_per_decode_step"], d["caption"]["library"]["library_ms_per_decode_step"])
for k, v in d["attention_roofline"].items(): print("  ", k, v["launches"], v["avg_launch_us"], v["frac"])
print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["all_fusion_gemms"])
P
