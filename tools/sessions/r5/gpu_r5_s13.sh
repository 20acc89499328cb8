#!/bin/bash
ulimit -c 0
# Round 5, session 13: SQ counters of the 256 x 256 tile (both forms) and the 256 x 128 tile on config E's up-projection: where do a k-step's clocks go?
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
for t in 256128 256257 256256; do
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_a_$t -- python $R/tools/gemm_graph_bench.py 4096 16384 4096 0 0 $t > $out/pmc_a_$t.txt 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $out/pmc_b_$t -- python $R/tools/gemm_graph_bench.py 4096 16384 4096 0 0 $t > $out/pmc_b_$t.txt 2>&1
done
cd $R
python - $out <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
for t in ("256128", "256257", "256256"):
    acc = collections.defaultdict(lambda: [0, 0.0]); dur = [0, 0.0]
    for sub in ("pmc_a_", "pmc_b_"):
        for f in glob.glob(f"{out}/{sub}{t}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_bf16" in r["Kernel_Name"]:
                    a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
        for f in glob.glob(f"{out}/{sub}{t}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_bf16" in r["Kernel_Name"]:
                    dur[0] += 1; dur[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"tile {t}: {dur[0]} launches, avg {dur[1] / max(dur[0], 1):.1f} us under counter collection")
    for k in sorted(acc):
        print(f"   {k:28s} {acc[k][1] / acc[k][0]:16.0f} per launch")
P
rm -rf $out/pmc_a_* $out/pmc_b_*/; du -sh $out
