#!/bin/bash
ulimit -c 0
# Round 5, session 8: the 256 x 256 tile on sixteen unified waves (gemm_bf16_u16_kernel): parity, isolated A/B against the planned tiles on config E's
# shapes, in-model A/B of the step at config B (12-block weight-gradient launches) and config E; OPT-backed decode session on the GPU
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_hip_primitives.py tests/test_model_plumbing.py -q -p no:cacheprovider -k "gemm or greedy_generate" > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt; grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-300
G="timeout 120 python tools/gemm_graph_bench.py"
( for shape in "4096 16384 4096 0 0" "4096 16384 4096 0 1" "4096 4096 16384 0 0" "4096 4096 16384 0 1" "8192 8192 8192 0 0"; do
    for t in 256128 256256; do $G $shape $t 2>/dev/null | tail -1; done
  done
  for shape in "4096 16384 4096 1 1" "16384 4096 4096 1 1" "1280 5120 1024 1 1" "2048 4096 4096 0 0"; do
    for t in 128002 256256; do $G $shape $t 2>/dev/null | tail -1; done
  done
  for t in 256128 256256; do EPI=act $G 4096 16384 4096 0 0 $t 2>/dev/null | tail -1; done
  for t in 256128 256256; do EPI=act_bwd $G 4096 16384 4096 0 1 $t 2>/dev/null | tail -1; done
  for t in 256128 256256; do EPI=res $G 4096 4096 16384 0 0 $t 2>/dev/null | tail -1; done
) > $out/gemm_u16_ab.txt 2>&1
cat $out/gemm_u16_ab.txt
B="python bench.py --no-cpu-baseline --caption-tokens 0 --companions off --steps 20 --warmup 3 --profile-steps 0"
run() { name=$1; shift; timeout 400 $B "$@" 2> $out/$name.err > $out/$name.json; python - "$out/$name.json" "$name" <<'P'
import sys, json
try:
    d = next(json.loads(l) for l in reversed(open(sys.argv[1]).read().strip().splitlines()) if l.startswith('{'))
    print(sys.argv[2], d["value"], d["unit"], d["ms_per_step"], "ms/step", "loss", d["config"].get("loss_last"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
}
export FLAMINGO_FUSION_LIB=debug
( FF_GEMM_U16=0 run B_planned1
  FF_GEMM_U16=1 run B_u16_1
  FF_GEMM_U16=0 run B_planned2
  FF_GEMM_U16=1 run B_u16_2
  FF_GEMM_U16=1 FF_GEMM_U16_MIN_TILES=256 run B_u16_min256
  FF_GEMM_U16=0 run E_planned --config E --steps 6 --warmup 2
  FF_GEMM_U16=1 run E_u16 --config E --steps 6 --warmup 2 ) | tee $out/step_u16_ab.txt
