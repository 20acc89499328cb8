#!/bin/bash
ulimit -c 0
tag=${1:-r3s4}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_benchpath.py -m gpu -q -s -p no:cacheprovider > $out/pytest.txt 2>&1
echo "pytest rc=$?"; grep -E "^\[benchpath|passed|failed|^FAILED|^E  " $out/pytest.txt | cut -c1-400 | tail -n 12
export FLAMINGO_FUSION_LIB=debug
for v in "FF_GEMM_PC_BL1=0" "FF_GEMM_PC_BL1=1" "FF_GEMM_PC_BL1=0" "FF_GEMM_PC_BL1=1"; do
  ( export $v; timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', d['config']['loss_last'])" )
done
unset FLAMINGO_FUSION_LIB
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/capprof -- python $R/tools/caption_profile.py --eager > $out/caption_profile.txt 2> $out/caption_profile.err
cd $R
f=$(find $out/capprof -name "*kernel_stats.csv" | head -1); cp $f $out/caption_kernel_stats.csv; rm -rf $out/capprof
cat $out/caption_profile.txt | tail -3
python - <<P
import csv
rows = list(csv.DictReader(open("$out/caption_kernel_stats.csv")))
ours = sum(float(r["TotalDurationNs"]) for r in rows if "ff::" in r["Name"] or "_ZN2ff" in r["Name"])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
n_ours = sum(int(r["Calls"]) for r in rows if "ff::" in r["Name"] or "_ZN2ff" in r["Name"]); n = sum(int(r["Calls"]) for r in rows)
print(f"caption (2 x 32 tokens): GPU busy {tot/1e6:.1f} ms, library {ours/1e6:.1f} ms ({n_ours} launches of {n})")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:8.1f} us {float(r["TotalDurationNs"])/1e6:8.2f} ms')
P
