#!/bin/bash
# round 4, session 2: the decode-shaped feed-forward kernels (csrc/ff_decode.hip): parity, the library's decode chain in isolation (A/B against the
# previous launches through the development build's switches), per-kernel averages, the caption leg.
ulimit -c 0
tag=${1:-r4s2}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_modules.py tests/test_model_plumbing.py tests/test_optim_state.py tests/test_hip_graph.py -m gpu -q -p no:cacheprovider --durations=5 > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 14 $out/pytest.txt
timeout 120 python -m pytest tests/test_model_plumbing.py -m gpu -q -p no:cacheprovider -k h64 -s 2>&1 | grep "report" | cut -c1-700
python tools/decode_chain_bench.py
for v in "FF_DECODE_FFW=0" "FF_DECODE_FFW=1 FF_DECODE_NT=1" "FF_DECODE_FFW=1 FF_DECODE_NT=0" "FF_DECODE_FFW=0" "FF_DECODE_FFW=1 FF_DECODE_NT=1"; do
  ( export FLAMINGO_FUSION_LIB=debug $v; echo "[$v] $(python tools/decode_chain_bench.py 2>&1 | tail -1)" )
done
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $R/tools/decode_chain_bench.py --eager --reps 10 > $out/prof.txt 2>&1
cd $R
python - <<P
import csv, glob
f = glob.glob("$out/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f'{r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:8.2f} us {float(r["TotalDurationNs"])/1e6:8.2f} ms')
P
rm -rf $out/prof
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --companions off 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['caption']; print('caption', c['value'], 'tok/s', c['ms_per_decode_step'], 'ms/step', c['library'])"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --companions off --backbone-tweaks on 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['caption']; print('caption (backbone tweaks on)', c['value'], 'tok/s', c['ms_per_decode_step'], 'ms/step', c['library'])"
