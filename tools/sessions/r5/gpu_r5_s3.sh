#!/bin/bash
ulimit -c 0
# Round 5, session 3: per-workgroup timeline of the resident fused kernels with / without the in-launch exchange; the tests session 2 left red
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 120 python tools/xattn_res_timeline.py > $out/timeline.txt 2>&1; echo "timeline rc=$?"; cat $out/timeline.txt | grep exchange
timeout 600 python -m pytest tests/test_hip_graph.py tests/test_hip_modules.py -q -p no:cacheprovider -k "rccl_reducer_on_one_rank or sharded_adamw_on_one_rccl or decode_shaped or golden_fp32" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 4 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-240
