#!/bin/bash
ulimit -c 0
# Round 5, session 20: per-step loss of the bench workload (stock backbones, LM dropout 0) with and without phase 3, graph replay and eager launches
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
export FLAMINGO_FUSION_LIB=debug
for v in 0 1; do
  echo "== FF_XATTN_LN3=$v"
  FF_XATTN_LN3=$v timeout 600 python tools/loss_trajectory.py --steps 24 --modes off:on,off:off --dropouts 0 2>&1 | cut -c1-400
done | tee $out/loss_trajectory_ln3.txt
