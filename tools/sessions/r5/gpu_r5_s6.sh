#!/bin/bash
ulimit -c 0
# Round 5, session 6: the full GPU suite on the restored tree; grouped weight-gradient launches on a second stream beside the backward pass
# (same-box A/B of the step: in line / beside, groups of 12 / 6 / 4, capture stream at high priority)
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-240
B="python bench.py --no-cpu-baseline --caption-tokens 0 --companions off --steps 20 --warmup 3 --profile-steps 0"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err > $out/$name.json; python - "$out/$name.json" "$name" <<'P'
import sys, json
try:
    d = next(json.loads(l) for l in reversed(open(sys.argv[1]).read().strip().splitlines()) if l.startswith('{'))
    print(sys.argv[2], d["value"], "images/s", d["ms_per_step"], "ms/step", "side", d["config"].get("wgrad_side_stream"), "loss", d["config"].get("loss_last"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
( run inline1 --wgrad-side-stream off
  run beside12 --wgrad-side-stream on
  run beside6 --wgrad-side-stream on --wgrad-group 6
  run beside4 --wgrad-side-stream on --wgrad-group 4
  run beside12_prio --wgrad-side-stream on --capture-priority high
  run inline6 --wgrad-side-stream off --wgrad-group 6
  run inline2 --wgrad-side-stream off
  run beside12b --wgrad-side-stream on ) | tee $out/wgrad_side_stream_ab.txt
tail -n 5 $out/beside12.err | cut -c1-300
