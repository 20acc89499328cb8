#!/bin/bash
# debug build of the library with per-workgroup GEMM phase timestamps (tools/gemm_timeline.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_dbg /tmp/ff_dbg
for f in ff_api ff_gemm ff_rowwise ff_attention ff_xattn_fused ff_optim ff_loss ff_elementwise ff_decode; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -mllvm -amdgpu-kernarg-preload-count=16 -DFF_GEMM_TIMELINE -DFF_XA_TIMELINE -Iinclude -c flamingo-mini_amd/csrc/$f.hip -o /tmp/ff_dbg/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/ff_dbg/*.o -o tools/_dbg/libflamingo_fusion_timeline.so
echo built tools/_dbg/libflamingo_fusion_timeline.so
