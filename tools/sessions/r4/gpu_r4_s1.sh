#!/bin/bash
# round 4, session 1: the whole GPU suite on the round's Python-layer changes (hidden-visibility library, launch-structure settings, h64 two-step
# fixture through GraphedTrainStep, piecewise capture), the default bench line (north-star default: stock backbones), and the three launch
# modes of the step with the gradient exchange going through a 1-rank RCCL group.
ulimit -c 0
tag=${1:-r4s1}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=6 > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 12 $out/pytest.txt
grep -h "h64 bf16 two-step report" $out/pytest.txt
timeout 200 python -m pytest tests/test_model_plumbing.py -m gpu -q -p no:cacheprovider -k h64 -s 2>&1 | grep "report" | cut -c1-600
timeout 900 python bench.py --gemm-table $out/gemm_table.txt > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-700 $out/bench_default.json
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3 --force-collectives"
for g in on piecewise off; do
  timeout 300 $B --graph $g --bucket-timeline > $out/fc_$g.json 2> $out/fc_$g.err; echo "== force-collectives graph=$g rc=$?"
  python - <<P
import json
try:
    d = json.loads(open("$out/fc_$g.json").read().strip().splitlines()[-1])
    print(d["value"], "img/s", d["ms_per_step"], "ms/step", d["config"]["graph_mode"], d["config"]["loss_last"])
    bt = d.get("bucket_timeline")
    if bt: print("  backward", bt["backward_ms"], "exchange finished", bt["exchange_finished_ms"], "exposed", bt["exposed_communication_ms"], "buckets", len(bt["buckets"]))
except Exception as e:
    print("no line:", e); print(open("$out/fc_$g.err").read()[-1500:])
P
done
timeout 200 $B --graph on 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no collectives, graph on:', d['value'], d['ms_per_step'])" 
du -sh $out
