#!/bin/bash
ulimit -c 0
# Round 5, session 2: full GPU suite on the round's tree (ABI 4: sync exchange, per-layer resampler exports), then the 1-rank-RCCL piecewise step with
# the resampler layer by layer (six buckets, six more backward segments) against the stack-level call (one 126 MB bucket at the end of backward)
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 6 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-240
B="python bench.py --no-cpu-baseline --caption-tokens 0 --companions off --steps 12 --warmup 3 --profile-steps 0 --graph piecewise --force-collectives --bucket-timeline"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err > $out/$name.json; python - "$out/$name.json" "$name" <<'P'
import sys, json
try:
    d = next(json.loads(l) for l in reversed(open(sys.argv[1]).read().strip().splitlines()) if l.startswith('{'))
except StopIteration:
    print(sys.argv[2], "no JSON line"); sys.exit(0)
bt = d.get("bucket_timeline") or {}
c = d["config"]
print(sys.argv[2], d["value"], "images/s", d["ms_per_step"], "ms/step | mode", c["graph_mode"], "layerwise", c.get("resampler_layerwise"), "| host", c.get("piecewise_host_ms_per_step"))
print("   eager timeline step: backward", bt.get("backward_ms"), "ms, exchange finished", bt.get("exchange_finished_ms"), "exposed", bt.get("exposed_communication_ms"), len(bt.get("buckets", [])), "buckets")
for r in bt.get("buckets", [])[-12:]:
    print("      ", r)
P
}
run lw_off1 --resampler-layerwise off
run lw_on1 --resampler-layerwise on
run lw_off2 --resampler-layerwise off
run lw_on2 --resampler-layerwise on
