#!/bin/bash
ulimit -c 0   # no core files: a GPU fault must not fill the scratch disk
# A/B on ONE box: tools/gpu_ab.sh <tag> "<ENV1>" "<ENV2>" ...   (each ENV string is exported for one bench run, e.g. "FF_XATTN_FUSED=0")
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
python -m pytest tests -m gpu -q -p no:cacheprovider -x > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 4 $out/pytest.txt
fi
i=0
for envs in "$@"; do
  i=$((i+1))
  ( export $envs; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 2 --gemm-table $out/gemm_$i.txt > $out/bench_$i.json 2> $out/bench_$i.err )
  echo "== [$envs]"; python - <<P
import json
d=json.loads(open("$out/bench_$i.json").read().strip().splitlines()[-1])
print(d["value"], "img/s", d["ms_per_step"], "ms/step", "loss", d["config"]["loss"], d["roofline"]["all_fusion_gemms"])
for k,v in d.get("attention_roofline",{}).items(): print("   ",k,v["launches"], v["avg_launch_us"])
P
done
