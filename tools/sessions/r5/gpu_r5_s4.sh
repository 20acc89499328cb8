#!/bin/bash
ulimit -c 0
# Round 5, session 4: 16-byte write-through stores + prefetched residual rows in the in-launch exchange: timeline, parity, same-box A/B of the step
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 120 python tools/xattn_res_timeline.py > $out/timeline.txt 2>&1; echo "timeline rc=$?"; grep "exchange=True" $out/timeline.txt | tail -4
timeout 600 python -m pytest tests/test_hip_modules.py tests/test_hip_benchpath.py -q -x -p no:cacheprovider -k "in_launch_exchange or decode_shaped or resident_fused or block" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-240
B="python bench.py --no-cpu-baseline --caption-tokens 0 --companions off --steps 20 --warmup 3 --profile-steps 0"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err > $out/$name.json; python - "$out/$name.json" "$name" <<'P'
import sys, json
d = next(json.loads(l) for l in reversed(open(sys.argv[1]).read().strip().splitlines()) if l.startswith('{'))
print(sys.argv[2], d["value"], "images/s", d["ms_per_step"], "ms/step")
P
}
run off1 --sync-exchange off
run on1 --sync-exchange on
run off2 --sync-exchange off
run on2 --sync-exchange on
run off3 --sync-exchange off
run on3 --sync-exchange on
