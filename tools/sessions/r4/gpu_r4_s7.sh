#!/bin/bash
# round 4, session 7: register-prefetched key / query tiles in the stand-alone attention kernels (resampler): parity + launch times
ulimit -c 0
tag=${1:-r4s7}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_primitives.py tests/test_hip_modules.py tests/test_hip_configs.py tests/test_hip_benchpath.py -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; tail -3 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | head
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --companions off --caption-tokens 0 > $out/bench.json 2> $out/bench.err
python - <<P
import json
d = next(json.loads(l) for l in reversed(open("$out/bench.json").read().strip().splitlines()) if l.startswith("{"))
print(d["value"], "img/s", d["ms_per_step"], "ms/step")
for k, v in d["attention_roofline"].items(): print("  ", k, v["launches"], v["avg_launch_us"], v["frac"])
P
