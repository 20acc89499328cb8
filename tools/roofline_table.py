#!/usr/bin/env python
"""Per-kernel roofline table of the fusion library from the committed profiles of one round:
    python tools/roofline_table.py r03 [directory] > profiles/r03_roofline_table.md
rocprofv3 kernel statistics (average duration, launches per step) x the counter traffic of the same session (HBM bytes per launch, corrected as
MI355X_MICROARCH.md prescribes) give the achieved HBM rate of every kernel; the GEMM table of the bench (HIP events) gives FLOP rates."""
import csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
DIR = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")      # (tools/collect_profiles.py points it at a session's collected/ directory)
P = lambda n: os.path.join(DIR, f"{rnd}_{n}")
stats = {r["Name"]: r for r in csv.DictReader(open(P("bench_b32_bf16_kernel_stats.csv")))}
summary = open(P("bench_b32_bf16_summary.md")).read()
steps = int(summary.split(" ms over ")[1].split(" steps")[0])
traffic = json.load(open(P("pmc_traffic.json")))["kernels"]
bench = json.load(open(P("bench_default.json")))


def short(name):
    if name.startswith("_Z"):
        import shutil, subprocess
        filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
        try:
            name = subprocess.run([filt, name], capture_output=True, text=True, timeout=10).stdout.strip() or name
        except (OSError, subprocess.SubprocessError):
            pass
    if name.startswith("_ZN2ff"):      # not demangled (the bf16 type code is newer than the filter): ff::<name><template arguments as mangled>
        import re
        m = re.match(r"_ZN2ff(\d+)", name)
        n = int(m.group(1))
        fn, rest = name[m.end():m.end() + n], name[m.end() + n:]
        targs = re.match(r"I(.*?)E+v", rest)
        args = targs.group(1) if targs else ""
        args = args.replace("DF16b", "bf16 ")
        args = re.sub(r"Lb([01])E?", lambda q: ("true " if q.group(1) == "1" else "false "), args)
        args = re.sub(r"Li(\d+)E?", lambda q: q.group(1) + " ", args).strip().replace(" ", ", ")
        name = f"ff::{fn}<{args}>" if args else f"ff::{fn}"
    name = name.replace("void ", "").split("(")[0].replace("__hip_bfloat16", "bf16")
    return name[:72]


rows = []
for k, v in traffic.items():
    r = next((s for n, s in stats.items() if n.startswith(k) or k in n), None)
    if r is None or v.get("hbm_bytes_per_launch", 0) < 1e5:
        continue
    us = float(r["AverageNs"]) / 1e3
    per_step = int(r["Calls"]) / steps
    mb = v["hbm_bytes_per_launch"] / 1e6
    rows.append((us * per_step / 1e3, short(k), per_step, us, mb, mb / us))
rows.sort(reverse=True)
print(f"# Fusion-library kernels against the HBM roofline ({rnd}: rocprofv3 durations x counter traffic of the same session)\n")
print("HBM peak 8 TB/s. Counter traffic includes Infinity-Cache hits and every re-fetch; for the GEMM kernels the MFMA roofline is the relevant one (second table).\n")
print("| kernel | launches/step | avg µs | counter MB/launch | TB/s | of 8 TB/s | ms/step |\n|---|---:|---:|---:|---:|---:|---:|")
for ms, k, n, us, mb, tb in rows:
    print(f"| `{k}` | {n:.0f} | {us:.1f} | {mb:.1f} | {tb:.2f} | {tb / 8:.2f} | {ms:.3f} |")
print("\n## GEMM shapes against the MFMA roofline (bench.py --gemm-table: HIP events around every launch of 3 eager steps; dense bf16 peak 2500 TFLOP/s)\n")
print("| M | N | K | problems | A / B layout | tile | split-K | launches/step | µs | TFLOP/s | of peak | ms/step |\n|---:|---:|---:|---:|---|---|---:|---:|---:|---:|---:|---:|")
for line in open(P("gemm_table.txt")).read().strip().splitlines()[1:]:
    M, N, K, nz, al, bl, tile, sk, n, us, tf, ms = line.split()
    print(f"| {M} | {N} | {K} | {nz} | {al} / {bl} | {tile} | {sk} | {n} | {us} | {tf} | {float(tf) / 2500:.2f} | {ms} |")
r = bench["roofline"]
print(f"\nDominant kernel by total time: `{r['kernel']}`, {r['achieved']} TFLOP/s = {r['frac']} of the peak over its {r['launches']} measured launches; "
      f"all fusion GEMMs {r['all_fusion_gemms']['tflops']} TFLOP/s, {r['all_fusion_gemms']['ms_per_step']} ms/step.")
