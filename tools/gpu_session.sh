#!/bin/bash
ulimit -c 0   # no core files: a GPU fault must not fill the scratch disk
# One GPU-box session: parity tests (all, no -x), default bench with the per-shape GEMM table.  Usage: tools/gpu_session.sh <tag> [extra bench args]
tag=${1:-run}; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $out/pytest.txt 2>&1; echo "pytest rc=$?" >> $out/pytest.txt
tail -n 25 $out/pytest.txt
python bench.py --steps 12 --warmup 4 --gemm-table $out/gemm_table.txt "$@" > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
cat $out/bench.json | head -c 3000
head -40 $out/gemm_table.txt
