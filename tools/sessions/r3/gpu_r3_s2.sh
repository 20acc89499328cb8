#!/bin/bash
ulimit -c 0
tag=${1:-r3s2}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
for t in "tests/test_hip_benchpath.py" "tests/test_hip_backbones.py -k bf16" "tests/test_hip_backbones.py -k f32" "tests/test_checkpoint_interchange.py tests/test_hip_optim.py"; do
  n=$(echo "$t" | tr ' /' '__')
  timeout 600 python -X faulthandler -m pytest $t -m gpu -q -s -p no:cacheprovider > $out/pytest_$n.txt 2>&1
  echo "== $t rc=$?"; grep -E "^\[benchpath|^\[backbones|passed|failed|^FAILED|^E  |Segmentation|line [0-9]+ in test" $out/pytest_$n.txt | cut -c1-600 | tail -n 30
done
