#!/bin/bash
ulimit -c 0
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
for i in 1 2 3; do
timeout 600 python -X faulthandler -m pytest tests/test_hip_graph.py -m gpu -q -p no:cacheprovider -x > $out/pytest_$i.txt 2>&1; echo "run $i rc=$?"; grep -v "^  File" $out/pytest_$i.txt | head -8 | cut -c1-300; tail -2 $out/pytest_$i.txt | cut -c1-200
done
