#!/bin/bash
ulimit -c 0
# r6 session 7: anatomy of a k-step with the timing builds of tools/build_pcmodes.sh (compile-time modes of gemm_bf16_pc_kernel: the producer
# side alone, the consumer side alone, fragment reads alone, MFMAs alone) - for the WEIGHT-GRADIENT kernel (128 x 128 tiles, M-major operands,
# two workgroups per CU) on a shape with the model's K and several rounds of tiles, and for the 128 x 160 feed-forward launches.
out=gpurun_out/r6s7; mkdir -p $out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{
for shape in "5120 5120 1024 1 1" "5120 5120 4096 1 1" "5120 5120 1024 0 0" "1024 5120 1280 0 0" "1024 5120 1280 0 1" "1024 1280 5120 0 0"; do
  echo "# $shape (M N K a_layout b_layout): the kernel | 1 = DMA side alone | 2 = consumers alone | 3 = fragment reads alone | 4 = MFMA alone"
  python tools/gemm_graph_bench.py $shape 2>&1 | tail -1
  for m in 1 2 3 4; do FLAMINGO_FUSION_LIB=$R/tools/_dbg/libflamingo_fusion_pcmode$m.so python tools/gemm_graph_bench.py $shape 2>&1 | tail -1 | sed "s/\[/[mode $m /"; done
  echo "# ONE weight buffer in all launches (Infinity-Cache-warm): the kernel | DMA side alone"
  UNIQUE=1 python tools/gemm_graph_bench.py $shape 2>&1 | tail -1
  UNIQUE=1 FLAMINGO_FUSION_LIB=$R/tools/_dbg/libflamingo_fusion_pcmode1.so python tools/gemm_graph_bench.py $shape 2>&1 | tail -1 | sed "s/\[/[mode 1 /"
done
} > $out/pc_timing_modes.txt 2>&1
cat $out/pc_timing_modes.txt
