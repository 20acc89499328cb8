#!/bin/bash
ulimit -c 0
# r6 session 8: the final tree once more - full GPU suite, smoke, default bench line (no companions)
out=gpurun_out/r6s8; mkdir -p $out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 4 $out/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -n 1 $out/smoke.txt
timeout 900 python bench.py --companions off > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python - <<P
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["sync_exchange_timeouts"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["hot_path"])
P
