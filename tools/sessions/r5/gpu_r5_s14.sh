#!/bin/bash
ulimit -c 0
# Round 5, session 14: the launch-mode row the round-end session lost (whole-step capture with a 1-rank exchange + the eager bucket-timeline step): its error
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
B2="python bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off --steps 12 --warmup 3"
timeout 300 $B2 --graph on --force-collectives --bucket-timeline > $out/full_bt.json 2> $out/full_bt.err; echo "rc=$?"; tail -n 25 $out/full_bt.err | cut -c1-300; tail -c 600 $out/full_bt.json
timeout 300 $B2 --graph on --force-collectives > $out/full.json 2> $out/full.err; echo "rc=$?"; tail -c 300 $out/full.json
