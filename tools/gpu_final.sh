#!/bin/bash
ulimit -c 0   # no core files: a GPU fault must not fill the scratch disk
# Round-end measurement session on ONE box (round 5: + configs A / C / D / E, config E's GEMM table and yardstick; - the decode probe and the 8-MFMA-wave A/B of round 4): full parity suite, smoke, the default bench line (with its companion runs), the rocprofv3 kernel
# trace of the same command, the two PMC traffic passes and the SQ MFMA-busy pass (separate runs, --kernel-trace only), the bucket timeline of
# the data-parallel launch structure, the kernel statistics of the caption leg.
#   tools/gpu_final.sh <tag> <round>      results under gpurun_out/<tag>/ ; tools/collect_profiles.py turns them into profiles/<round>_*
tag=$1; rnd=${2:-r04}; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -n 2 $out/smoke.txt
timeout 900 python bench.py --gemm-table $out/gemm_table.txt > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-400 $out/bench_default.json
cd /tmp
B="python $R/bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- $B --steps 5 --warmup 2 > $out/prof_bench.json 2> $out/prof.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B --steps 1 --warmup 1 --graph off > /dev/null 2> $out/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B --steps 1 --warmup 1 --graph off > /dev/null 2> $out/pmc_write.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/pmc_sq -- $B --steps 1 --warmup 1 --graph off > /dev/null 2> $out/pmc_sq.err
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/bucket -- $B --wgrad-group 4 --kv-group 4 --steps 2 --warmup 2 --graph off > /dev/null 2> $out/bucket.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/capprof -- python $R/tools/caption_profile.py --eager > $out/caption_profile.txt 2> $out/caption_profile.err
cd $R
# round 4: hipBLASLt beside every GEMM shape of the step (same method for both), the decode kernels in isolation (probe with phase timestamps, the 36-block
# chain), the three launch modes of the step with the gradient exchange going through a 1-rank RCCL group + the bucket timeline, the 128 x 160 tile with 4 / 8 MFMA waves
timeout 400 python tools/gemm_yardstick.py $out/gemm_table.txt > $out/gemm_yardstick.txt 2> $out/gemm_yardstick.err; head -5 $out/gemm_yardstick.txt
( python tools/decode_chain_bench.py 2>&1 | tail -1; for v in "FF_DECODE_FFW=0" "FF_DECODE_FFW=1"; do ( export FLAMINGO_FUSION_LIB=debug $v; echo "[development build, $v] $(python tools/decode_chain_bench.py 2>&1 | tail -1)" ); done ) > $out/decode_chain.txt
cat $out/decode_chain.txt
bash tools/sessions/r5/gpu_r5_s15.sh $tag > $out/launch_modes.log 2>&1      # the launch-mode table (stderr of every run kept, a run without a JSON line is named)
head -4 $out/launch_modes_one_rank_rccl.txt
# round 5: the other configurations of BASELINE.json (stock backbones, 6 timed steps), config E with its per-shape GEMM table and hipBLASLt beside every one of its shapes
for c in A C D; do timeout 400 python bench.py --config $c --no-cpu-baseline --caption-tokens 0 --companions off --steps 6 --warmup 2 2> $out/bench_config_$c.err | tail -1 > $out/bench_config_$c.json; done
timeout 600 python bench.py --config E --no-cpu-baseline --caption-tokens 0 --companions off --steps 6 --warmup 2 --gemm-table $out/gemm_table_E.txt 2> $out/bench_config_E.err | tail -1 > $out/bench_config_E.json
timeout 500 python tools/gemm_yardstick.py $out/gemm_table_E.txt > $out/gemm_yardstick_E.txt 2> $out/gemm_yardstick_E.err; head -4 $out/gemm_yardstick_E.txt | cut -c1-200
for c in A C D E; do python -c "
import json,sys
try:
    d=json.loads(open('$out/bench_config_$c.json').read().strip().splitlines()[-1]); print('config $c:', d['value'], d['unit'], d['ms_per_step'], 'ms/step')
except Exception as e: print('config $c: no line', e)"; done
python tools/bucket_timeline.py $(find $out/bucket -name "*kernel_trace.csv" | head -1) > $out/bucket_timeline.txt 2>&1; head -20 $out/bucket_timeline.txt
python - > $out/caption_decode_kernels.txt <<P
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$out/capprof/**/*kernel_stats.csv", recursive=True)[0])))
ours = lambda r: "ff::" in r["Name"] or "_ZN2ff" in r["Name"]
tot = sum(float(r["TotalDurationNs"]) for r in rows); mine = sum(float(r["TotalDurationNs"]) for r in rows if ours(r))
n = sum(int(r["Calls"]) for r in rows); n_mine = sum(int(r["Calls"]) for r in rows if ours(r))
print(open("$out/caption_profile.txt").read().strip())
print(f"caption leg, decode steps launched one by one (2 x 32 tokens x batch 32 = 62 cached steps + 2 prompt steps): GPU busy {tot/1e6:.1f} ms, fusion library {mine/1e6:.1f} ms ({n_mine} of {n} launches)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    print(f'{"*" if ours(r) else " "} {r["Name"][:110]:110s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:8.1f} us {float(r["TotalDurationNs"])/1e6:8.2f} ms')
P
head -4 $out/caption_decode_kernels.txt
grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-220
if ! python tools/collect_profiles.py $out $rnd $out/collected > $out/collect.txt 2>&1; then   # keep the inputs (compressed) if the post-processing failed
  mkdir -p $out/raw; for f in $(find $out/prof $out/pmc_* -name "*kernel_stats.csv" -o -name "*counter_collection.csv" -o -path "*pmc_sq*" -name "*kernel_trace.csv"); do
    gzip -c $f > $out/raw/$(echo $f | sed "s#$out/##; s#/#_#g").gz; done
fi
tail -n 2 $out/collect.txt
# gpurun copies back at most 64 MiB: keep the collected summaries and the small logs, drop the raw traces
rm -rf $out/prof $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/bucket $out/capprof
du -sh $out
