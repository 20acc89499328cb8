#!/bin/bash
# rocprofv3 kernel-trace of the default bench (graph replay): tools/gpu_profile.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --caption-tokens 0 > $R/$out/prof_bench.json 2> $R/$out/prof_bench.err
cd $R
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); echo "stats file: $f"
python tools/summarize_rocprof.py $f --steps-total 8 > $out/summary.md; head -45 $out/summary.md
cp $f $out/kernel_stats.csv
