#!/bin/bash
ulimit -c 0
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_two_ranks.py tests/test_hip_optim.py -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 40 $out/pytest.txt | cut -c1-300
