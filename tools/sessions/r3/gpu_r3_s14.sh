#!/bin/bash
# round 3, session 14: the reuse / accumulation test on the real kernels; capture after an eager forward (three variants, separate processes)
out=gpurun_out/r3s14; mkdir -p $out
timeout 600 python -m pytest tests/test_hip_benchpath.py -q -m gpu -k reused -s > $out/pytest.txt 2>&1; echo "pytest rc=$?"
tail -6 $out/pytest.txt | cut -c1-400
for mode in nograd drop keep; do
  timeout 300 python tools/capture_after_eager.py $mode > $out/capture_$mode.txt 2>&1; echo "capture_after_eager $mode rc=$?"
  tail -3 $out/capture_$mode.txt | cut -c1-300
done
