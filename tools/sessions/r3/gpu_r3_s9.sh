#!/bin/bash
ulimit -c 0
tag=${1:-r3s9}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
( export FLAMINGO_FUSION_LIB=debug FF_XATTN_RES=0; python tools/res_compare.py run /tmp/old.pt ) 2> /dev/null
( export FLAMINGO_FUSION_LIB=debug FF_XATTN_RES=1; python tools/res_compare.py run /tmp/new.pt ) 2> /dev/null
python tools/res_compare.py diff /tmp/old.pt /tmp/new.pt | grep -E "out|dvf|to_q"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 > $out/pytest.txt 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^E  " $out/pytest.txt | cut -c1-200 | tail -n 8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gemm-table $out/gemm_table.txt > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-900 $out/bench_default.json
