#!/bin/bash
ulimit -c 0
# Round 5, session 21: the 8-GPU session script rehearsed on one GPU with the final tree (phase 3, per-layer resampler, one-rank SUM, watchdog drain)
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 bash tools/sessions/r5/scale8.sh $out/scale8 --rehearsal > $out/scale8_rehearsal.txt 2>&1; echo "scale8 rehearsal rc=$?"; tail -n 8 $out/scale8_rehearsal.txt | cut -c1-260
for f in $out/scale8/*.err; do if grep -q "Traceback" $f; then echo "== $f"; tail -n 6 $f | cut -c1-240; fi; done
