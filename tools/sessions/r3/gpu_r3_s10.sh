#!/bin/bash
ulimit -c 0
tag=${1:-r3s10}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
for st in 0 3 4; do
  timeout 200 python tools/gemm_ab.py --shapes "out.fwd,q.dgrad" --tile 64 --stages $st --tag "tile64 stages=$st" 2> /dev/null
done
timeout 200 python tools/gemm_ab.py --shapes "out.fwd,q.dgrad" --tile 6412 --stages 2 --tag "tile 64x128" 2> /dev/null
timeout 200 python tools/gemm_ab.py --shapes "out.fwd" --tile 3264 --stages 0 --tag "tile 32x64 pc" 2> /dev/null
timeout 200 python tools/gemm_ab.py --shapes "out.fwd,q.dgrad" --tile 128002 --stages 0 --tag "tile 128 pc" 2> /dev/null
