#!/bin/bash
ulimit -c 0
# Round 5, session 1: the in-launch exchange (to_out / d LN(y) inside the fused attention launches): parity first, then the same-box A/B of the step
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 420 python -m pytest tests/test_hip_modules.py -q -x -p no:cacheprovider -k "in_launch_exchange or resident_fused or decode_shaped" > $out/pytest_a.txt 2>&1; echo "pytest_a rc=$?"; tail -n 15 $out/pytest_a.txt
timeout 300 python -m pytest tests/test_hip_benchpath.py tests/test_hip_primitives.py -q -x -p no:cacheprovider > $out/pytest_b.txt 2>&1; echo "pytest_b rc=$?"; tail -n 5 $out/pytest_b.txt
B="python bench.py --no-cpu-baseline --caption-tokens 0 --companions off --steps 12 --warmup 3 --profile-steps 2"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err > $out/$name.json; python - "$out/$name.json" "$name" <<'P'
import sys, json
d = next(json.loads(l) for l in reversed(open(sys.argv[1]).read().strip().splitlines()) if l.startswith('{'))
ar = d.get("attention_roofline", {})
print(sys.argv[2], d["value"], "images/s", d["ms_per_step"], "ms/step |", {k: (v["avg_launch_us"], v["launches"]) for k, v in ar.items()}, "| gemm", d["roofline"].get("all_fusion_gemms"))
P
}
run off1 --sync-exchange off
run on1 --sync-exchange on
run off2 --sync-exchange off
run on2 --sync-exchange on
