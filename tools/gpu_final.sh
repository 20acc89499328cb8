#!/bin/bash
ulimit -c 0   # no core files: a GPU fault must not fill the scratch disk
# Round-end measurement session on ONE box: full parity suite, the default bench line, the rocprofv3 kernel trace of the same command,
# the two PMC traffic passes and the SQ MFMA-busy pass (separate runs, --kernel-trace only), the backbone-tweaks-off line.
#   tools/gpu_final.sh <tag>          results under gpurun_out/<tag>/ ; tools/collect_profiles.py turns them into profiles/rNN_*
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $out/pytest.txt 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -n 2 $out/smoke.txt
timeout 420 python bench.py --gemm-table $out/gemm_table.txt > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"; cut -c1-600 $out/bench_default.json
timeout 240 python bench.py --backbone-tweaks off --no-cpu-baseline --caption-tokens 0 --profile-steps 0 > $out/bench_stock_backbones.json 2> $out/bench_stock_backbones.err; cut -c1-200 $out/bench_stock_backbones.json
cd /tmp
B="python $R/bench.py --no-cpu-baseline --caption-tokens 0 --profile-steps 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- $B --steps 5 --warmup 2 > $out/prof_bench.json 2> $out/prof.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B --steps 1 --warmup 1 --graph off > /dev/null 2> $out/pmc_fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B --steps 1 --warmup 1 --graph off > /dev/null 2> $out/pmc_write.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/pmc_sq -- $B --steps 1 --warmup 1 --graph off > /dev/null 2> $out/pmc_sq.err
cd $R
grep -E "^(FAILED|ERROR)" $out/pytest.txt | cut -c1-220
if ! python tools/collect_profiles.py $out r02 $out/collected > $out/collect.txt 2>&1; then   # keep the inputs (compressed) if the post-processing failed
  mkdir -p $out/raw; for f in $(find $out/prof $out/pmc_* -name "*kernel_stats.csv" -o -name "*counter_collection.csv" -o -path "*pmc_sq*" -name "*kernel_trace.csv"); do
    gzip -c $f > $out/raw/$(echo $f | sed "s#$out/##; s#/#_#g").gz; done
fi
tail -n 2 $out/collect.txt
# gpurun copies back at most 64 MiB: keep the collected summaries and the small logs, drop the raw traces
rm -rf $out/prof $out/pmc_fetch $out/pmc_write $out/pmc_sq
du -sh $out
