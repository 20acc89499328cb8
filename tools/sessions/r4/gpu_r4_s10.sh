#!/bin/bash
ulimit -c 0
# where the 3.6 ms between the single-GPU step (41.2 ms) and the piecewise step with a 1-rank RCCL exchange (44.8 ms) go:
# launch structure (4 layers per weight-gradient group / K,V projection call), segmentation into sub-graphs, the collectives themselves
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
B="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
run() { name=$1; shift; timeout 300 $B "$@" 2> $out/$name.err | python -c "
import sys, json
d = next(json.loads(l) for l in reversed(sys.stdin.read().strip().splitlines()) if l.startswith('{'))
print('$name:', d['value'], 'images/s', d['ms_per_step'], 'ms/step, mode', d['config'].get('graph_mode'), d['config'].get('collectives'))"; }
run full --graph on
run full_group4 --graph on --wgrad-group 4 --kv-group 4
run piecewise --graph piecewise
run piecewise_group4 --graph piecewise --wgrad-group 4 --kv-group 4
run piecewise_rccl --graph piecewise --force-collectives
run full_rccl --graph on --force-collectives
run full --graph on
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $R/bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 5 --warmup 2 --graph piecewise --force-collectives > /dev/null 2> $out/prof.err
python - <<P
import csv, glob
f = glob.glob("$out/prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("piecewise + 1-rank RCCL, 7 steps traced: GPU busy %.1f ms" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    if "ccl" in n.lower() or "AllReduce" in n or "copyBuffer" in n or "fillBuffer" in n or "elementwise_kernel" in n and float(r["TotalDurationNs"]) > 3e6:
        print(f'  {n[:120]:120s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:9.1f} us {float(r["TotalDurationNs"])/1e6:8.2f} ms')
P
rm -rf $out/prof
