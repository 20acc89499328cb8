#!/bin/bash
# round 3, session 3: GEMM kernel variants - correctness of the new tiles, A/B of M-major 128x160 staging and the L2 pre-touch
ulimit -c 0
tag=${1:-r3s3}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_primitives.py tests/test_hip_benchpath.py -m gpu -q -s -p no:cacheprovider -x > $out/pytest.txt 2>&1
echo "pytest rc=$?"; grep -E "^\[benchpath|passed|failed|^FAILED|^E  " $out/pytest.txt | cut -c1-500 | tail -n 20
export FLAMINGO_FUSION_LIB=debug
run() { ( export $1; timeout 300 python tools/gemm_ab.py --shapes "$2" --tag "$1" ) >> $out/gemm_ab.txt 2>> $out/gemm_ab.err; }
run "FF_GEMM_PF=0 FF_GEMM_PC_BL1=0" "ff,out,q.d,rs"
run "FF_GEMM_PF=0 FF_GEMM_PC_BL1=1" "dgrad"
run "FF_GEMM_PF=4 FF_GEMM_PF_SHARE=1" "ff,rs"
run "FF_GEMM_PF=6 FF_GEMM_PF_SHARE=1" "ff,rs"
run "FF_GEMM_PF=8 FF_GEMM_PF_SHARE=1" "ff"
run "FF_GEMM_PF=4 FF_GEMM_PF_SHARE=0" "ff"
run "FF_GEMM_PF=3 FF_GEMM_PF_SHARE=1" "ff"
cat $out/gemm_ab.txt
tail -3 $out/gemm_ab.err
