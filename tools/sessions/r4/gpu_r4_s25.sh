#!/bin/bash
ulimit -c 0
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
AMD_LOG_LEVEL=1 timeout 600 python -m pytest "tests/test_hip_graph.py::test_piecewise_graphs_with_eager_collectives_equal_eager_and_full_capture" -m gpu -q -p no:cacheprovider -x -s > $out/pytest.txt 2>&1; echo "pytest rc=$?"
grep -v "^  File\|^$" $out/pytest.txt | head -60 | cut -c1-400
