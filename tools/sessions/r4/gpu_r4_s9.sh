#!/bin/bash
ulimit -c 0
# resident-panel walk of the grouped weight-gradient launches: parity, then A/B of the super-panel budget (development build) with the per-launch
# time of the wgrad rows and a FETCH_SIZE pass each
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_benchpath.py tests/test_hip_primitives.py -m gpu -q -p no:cacheprovider -x > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt
export FLAMINGO_FUSION_LIB=debug
for kb in 0 3072 1536 0 3072; do
  ( export FF_GEMM_WALK_KB=$kb; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --caption-tokens 0 --profile-steps 2 --companions off --gemm-table $out/gemm_$kb.txt > $out/bench_$kb.json 2> $out/bench_$kb.err )
  echo "== FF_GEMM_WALK_KB=$kb"; python - <<P
import json
d=next(json.loads(l) for l in reversed(open("$out/bench_$kb.json").read().strip().splitlines()) if l.startswith("{"))
print(d["value"], "img/s", d["ms_per_step"], "ms/step", d["roofline"]["achieved"], d["roofline"]["frac"])
for l in open("$out/gemm_$kb.txt").read().strip().splitlines()[1:]:
    r = l.split()
    if int(r[3]) > 1: print("   ", l)
P
done
cd /tmp
B="python $R/bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off"
for kb in 0 3072; do
  ( export FF_GEMM_WALK_KB=$kb; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_$kb -- $B --steps 1 --warmup 1 --graph off > /dev/null 2> $out/pmc_$kb.err )
  echo "== FETCH_SIZE, FF_GEMM_WALK_KB=$kb"; python - <<P
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob("$out/pmc_$kb/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16_pc_kernel<128, 128, 1, 1" in r["Kernel_Name"]:
            a = acc[int(r["Grid_Size"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for g, (n, kb) in sorted(acc.items()): print(f"    grid {g:8d} ({g // 512} tiles) launches {n:3d}  2 x FETCH_SIZE {2 * kb / n * 1024 / 1e6:8.1f} MB per launch")
P
done
rm -rf $out/pmc_*/
du -sh $out
