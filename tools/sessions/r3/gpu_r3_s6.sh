#!/bin/bash
ulimit -c 0
tag=${1:-r3s6}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_hip_primitives.py tests/test_hip_optim.py tests/test_hip_benchpath.py -m gpu -q -p no:cacheprovider -k "not resampler" > $out/pytest.txt 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^E  " $out/pytest.txt | cut -c1-300 | tail -n 12
# the borderline d visual_features error of the 40-key case: the same test on the previous kernels (development build, FF_XATTN_RES=0)
( export FLAMINGO_FUSION_LIB=debug FF_XATTN_RES=0; timeout 300 python -m pytest "tests/test_hip_modules.py::test_resident_fused_kernels_bf16_vs_oracle" -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^E       Assert" | cut -c1-200 )
export FLAMINGO_FUSION_LIB=debug
for v in "FF_GEMM_SKINNY=0" "FF_GEMM_SKINNY=1" "FF_GEMM_SKINNY=0" "FF_GEMM_SKINNY=1"; do
  ( export $v; timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --caption-tokens 32 --profile-steps 0 --companions off 2> /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$v]', d['value'], 'img/s', d['ms_per_step'], 'ms/step', 'caption', d['caption']['value'], d['caption']['ms_per_decode_step'])" )
done
