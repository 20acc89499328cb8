#!/bin/bash
ulimit -c 0   # no core files: a GPU fault must not fill the scratch disk
# bench.py on every BASELINE.json config: tools/gpu_configs.sh <tag> [configs...]
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for c in "$@"; do
  timeout 600 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --caption-tokens 0 --companions off --profile-steps 2 --gemm-table $out/gemm_$c.txt > $out/bench_$c.json 2> $out/bench_$c.err; echo "config $c rc=$?"
  python - <<P
import json
try:
    d=json.loads(open("$out/bench_$c.json").read().strip().splitlines()[-1])
    print("  ", d["metric"], d["value"], d["unit"], d["ms_per_step"], "ms/step loss", d["config"]["loss"], "graph", d["config"]["hip_graph"], d["roofline"]["kernel"] if d["roofline"] else None, d["roofline"]["frac"] if d["roofline"] else None)
except Exception as e:
    print("   no json:", e); print(open("$out/bench_$c.err").read()[-1500:])
P
done
