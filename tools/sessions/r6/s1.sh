#!/bin/bash
ulimit -c 0
# r6 session 1: the new robustness tests first (fast feedback), then the full GPU suite, then the default bench line
out=gpurun_out/r6s1; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_modules.py -m gpu -q -p no:cacheprovider -k "persistent or documented_width or timeout_is_raised" > $out/new_tests.txt 2>&1; echo "new tests rc=$?"; tail -n 30 $out/new_tests.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 25 $out/pytest.txt
timeout 900 python bench.py --steps 12 --warmup 4 --gemm-table $out/gemm_table.txt > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
head -c 2500 $out/bench.json; tail -5 $out/bench.err
