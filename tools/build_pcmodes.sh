#!/bin/bash
# Timing builds of the library for the anatomy of a GEMM k-step (tools/sessions/r6/s7.sh): libflamingo_fusion_pcmode{1,2,3,4}.so under tools/_dbg/,
# each with -DFF_GEMM_PCMODE=n compiled into gemm_bf16_pc_kernel (1: DMA side alone, 2: consumers alone, 3: fragment reads alone, 4: MFMA alone).
# Results of these libraries are WRONG by construction; select one with FLAMINGO_FUSION_LIB=<path>.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_dbg /tmp/ff_pcm
O=flamingo-mini_amd/csrc/_obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -mllvm -amdgpu-kernarg-preload-count=16 -Iinclude"
for m in 1 2 3 4; do
  /opt/rocm/bin/hipcc $FLAGS -DFF_GEMM_PCMODE=$m -c flamingo-mini_amd/csrc/ff_gemm.hip -o /tmp/ff_pcm/ff_gemm_$m.o &
done
wait
for m in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ff_pcm/ff_gemm_$m.o $O/ff_api.o $O/ff_rowwise.o $O/ff_attention.o $O/ff_xattn_fused.o $O/ff_optim.o $O/ff_loss.o $O/ff_elementwise.o $O/ff_decode.o -o tools/_dbg/libflamingo_fusion_pcmode$m.so
done
ls -la tools/_dbg/libflamingo_fusion_pcmode*.so
