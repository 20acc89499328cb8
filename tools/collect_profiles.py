"""Turn one tools/gpu_final.sh session (gpurun_out/<tag>/) into the committed artefacts profiles/<round>_*:
    python tools/collect_profiles.py gpurun_out/final_r02 r02 [outdir]
(tools/gpu_final.sh runs it on the GPU box with outdir = gpurun_out/<tag>/collected and drops the raw traces, which exceed what gpurun copies back)
bench line, rocprofv3 kernel stats + summary, HBM traffic per launch (FETCH_SIZE / WRITE_SIZE passes), MFMA-busy (SQ pass)."""
import csv, glob, json, os, shutil, subprocess, sys
from collections import defaultdict

src, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)


def find(sub, pat):
    hits = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    assert hits, (sub, pat)
    return hits[0]


def short(name):
    return name.split("(")[0].replace("void ", "")


def is_ours(name):
    return "ff::" in name or "_ZN2ff" in name


stats = find("prof", "*kernel_stats.csv")
shutil.copy(stats, os.path.join(P, f"{rnd}_bench_b32_bf16_kernel_stats.csv"))
prof_line = json.loads(open(os.path.join(src, "prof_bench.json")).read().strip().splitlines()[-1])
steps_total = 2 + prof_line["warmup"] + prof_line["steps"]          # 2 eager warm-up steps inside GraphedTrainStep + the replays
with open(os.path.join(P, f"{rnd}_bench_b32_bf16_summary.md"), "w") as f:
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_rocprof.py"), stats, "--steps-total", str(steps_total)], stdout=f, check=True)

# HBM traffic: counter values are KB; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md, HBM section)
def pmc(sub):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(find(sub, "*counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if is_ours(r["Kernel_Name"]):
                a = acc[short(r["Kernel_Name"])][r["Counter_Name"]]
                a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc

fetch, write = pmc("pmc_fetch"), pmc("pmc_write")


# ---- the same counters PER SHAPE: a kernel instantiation serves several GEMM shapes, which differ in their grid; the algorithmic bytes of a
# shape can only be compared with the counter traffic of ITS launches (175 MB vs 108 MB was a comparison across shapes) ----
def pmc_by_grid(sub, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(find(sub, "*counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if is_ours(r["Kernel_Name"]) and r["Counter_Name"] == counter:
                a = acc[(short(r["Kernel_Name"]), int(r["Grid_Size"]))]
                a[0] += 1; a[1] += float(r["Counter_Value"])
    return acc


def gemm_kernel_of(tile, al, bl, split):
    """(kernel name as rocprofv3 prints it, block tile, threads per workgroup) of a row of the bench's GEMM table (ff_gemm.hip: run_bf16_dma)"""
    if tile == 128002: return f"ff::gemm_bf16_pc_kernel<128, 128, {al}, {bl}, 2, 2, 4, 4>", 128, 128, 512
    if tile == 128160 and split > 1: return f"ff::gemm_bf16_pc_kernel<128, 160, {al}, {bl}, 3, 1, 4, 4>", 128, 160, 512
    if tile == 128160: return f"ff::gemm_bf16_pc_kernel<128, 160, {al}, {bl}, 4, 1, 8, 4>", 128, 160, 768
    if tile == 64002: return f"ff::gemm_bf16_pc_kernel<64, 64, {al}, {bl}, 3, 2, 4, 4>", 64, 64, 512
    if tile == 3264: return "ff::gemm_bf16_pc_kernel<32, 64, 0, 0, 4, 2, 4, 4>", 32, 64, 512
    if tile == 256128 and al == 0: return f"ff::gemm_bf16_pc_kernel<256, 128, 0, {bl}, 3, 1, 8, 8>", 256, 128, 1024
    if tile == 256256: return f"ff::gemm_bf16_u16_kernel<{al}, {bl}, {'true' if bl == 0 else 'false'}>", 256, 256, 1024
    return None, 0, 0, 0


if os.path.exists(os.path.join(src, "gemm_table.txt")):
    fetch_g, write_g = pmc_by_grid("pmc_fetch", "FETCH_SIZE"), pmc_by_grid("pmc_write", "WRITE_SIZE")
    lines = ["M N K nz aL bL tile splitK launches/step us/launch algorithmic_MB counter_MB counter/algorithmic   "
             "(counter = (2 FETCH_SIZE + WRITE_SIZE) KiB per launch of THIS shape; algorithmic = operands read once + output written once, fp32 slabs for split-K; NOT in the algorithmic column: the epilogue's second tensor (pre-activation saved / read back, residual: +M*N*2 bytes on the M x 5120 rows and the gated outputs), and the per-XCD copies - FETCH_SIZE counts what the eight private L2s request from the fabric, so an operand shared by workgroups on all XCDs is counted up to eight times although the 256 MiB Infinity Cache, not HBM, serves the repeats: for 1024x5120x1280 the cheapest 8-way split is A x 8 + B = 34 MB of reads, which is what the counter shows)"]
    for row in open(os.path.join(src, "gemm_table.txt")).read().strip().splitlines()[1:]:
        M, N, K, nz, al, bl, tile, sk = (int(v) for v in row.split()[:8])
        per_step, us = row.split()[8], row.split()[9]
        name, bm, bn, threads = gemm_kernel_of(tile, al, bl, sk)
        if name is None:
            continue
        grid = -(-M // bm) * -(-N // bn) * sk * nz * threads
        f_ = next((v for (k, g), v in fetch_g.items() if g == grid and k.replace(" ", "") == name.replace(" ", "")), None)
        w_ = next((v for (k, g), v in write_g.items() if g == grid and k.replace(" ", "") == name.replace(" ", "")), None)
        alg = nz * ((M * K + N * K) * 2 + (M * N * 4 * sk if sk > 1 else M * N * 2)) / 1e6
        if f_ is None or w_ is None:
            lines.append(f"{M} {N} {K} {nz} {al} {bl} {tile} {sk} {per_step} {us} {alg:.1f} n/a n/a")
            continue
        cnt = (2 * f_[1] / f_[0] + w_[1] / w_[0]) * 1024 / 1e6
        lines.append(f"{M} {N} {K} {nz} {al} {bl} {tile} {sk} {per_step} {us} {alg:.1f} {cnt:.1f} {cnt / alg:.2f}")
    open(os.path.join(P, f"{rnd}_gemm_traffic.txt"), "w").write("\n".join(lines) + "\n")
traffic = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over one eager step of the default bench; counter values are KB; "
                    "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 "
                    "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated; memory-side counters include Infinity-Cache hits.", "kernels": {}}
for k in sorted(fetch, key=lambda k: -fetch[k]["FETCH_SIZE"][1]):
    n, f_kb = fetch[k]["FETCH_SIZE"]
    wn, w_kb = write.get(k, {}).get("WRITE_SIZE", [1, 0.0])
    traffic["kernels"][k] = {"launches": n, "fetch_kb_avg": round(f_kb / n, 1), "write_kb_avg": round(w_kb / max(wn, 1), 1),
                             "hbm_bytes_per_launch": int((2 * f_kb / n + w_kb / max(wn, 1)) * 1024)}
json.dump(traffic, open(os.path.join(P, f"{rnd}_pmc_traffic.json"), "w"), indent=1)

sq = pmc("pmc_sq")
dur = defaultdict(lambda: [0, 0.0])
with open(find("pmc_sq", "*kernel_trace.csv")) as f:
    for r in csv.DictReader(f):
        if is_ours(r["Kernel_Name"]):
            d = dur[short(r["Kernel_Name"])]
            d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
busy = {"_note": "rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES over one eager step of the default bench; per-launch "
                 "averages.  mfma_busy_frac_of_kernel = SQ_VALU_MFMA_BUSY_CYCLES / (avg duration * 2.4 GHz * 1024 SIMDs): the share of the chip's SIMD-cycles "
                 "during the kernel in which the MFMA pipe was busy (durations under counter collection are a few % longer than in the plain trace).", "kernels": {}}
for k in sorted(sq, key=lambda k: -sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1]):
    n = sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"][0]
    us = dur[k][1] / max(dur[k][0], 1)
    mf = sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"][1] / n
    busy["kernels"][k] = {"launches": n, "avg_us": round(us, 1), "sq_busy_cu_cycles": int(sq[k]["SQ_BUSY_CU_CYCLES"][1] / n), "sq_valu_mfma_busy_cycles": int(mf),
                          "sq_waves": int(sq[k]["SQ_WAVES"][1] / n), "mfma_busy_frac_of_kernel": round(mf / (us * 1e-6 * 2.4e9 * 1024), 3) if us else None}
json.dump(busy, open(os.path.join(P, f"{rnd}_sq_mfma_busy.json"), "w"), indent=1)

# the bench line: take `traffic` of its dominant kernel from the PMC file of the SAME session
line = next(json.loads(l) for l in reversed(open(os.path.join(src, "bench_default.json")).read().strip().splitlines()) if l.startswith("{"))
k = line["roofline"]["kernel"]
line["roofline"]["traffic"] = traffic["kernels"].get(k, {}).get("hbm_bytes_per_launch")
json.dump(line, open(os.path.join(P, f"{rnd}_bench_default.json"), "w"), indent=1)
if os.path.exists(os.path.join(src, "bench_stock_backbones.json")):      # (round 3 on: the default line carries it as `stock_backbones`)
    stock = json.loads(open(os.path.join(src, "bench_stock_backbones.json")).read().strip().splitlines()[-1])
    json.dump(stock, open(os.path.join(P, f"{rnd}_bench_stock_backbones.json"), "w"), indent=1)
for extra in ("gemm_table.txt", "bucket_timeline.txt", "caption_decode_kernels.txt", "smoke.txt", "gemm_yardstick.txt", "decode_probe.txt", "decode_chain.txt",
              "launch_modes_one_rank_rccl.txt", "gemm_ncw8_ab.txt", "bench_config_A.json", "bench_config_C.json", "bench_config_D.json", "bench_config_E.json", "gemm_table_E.txt", "gemm_yardstick_E.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(P, f"{rnd}_{extra}"))
print("dominant kernel", k, "traffic", line["roofline"]["traffic"], "frac", line["roofline"]["frac"], "value", line["value"])

# derived table: every library kernel against its roofline (tests/test_profiles_consistency.py regenerates and compares it)
table = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_table.py"), rnd, P], capture_output=True, text=True)
if table.returncode == 0:
    open(os.path.join(P, f"{rnd}_roofline_table.md"), "w").write(table.stdout)
else:
    print("roofline table failed:", table.stderr[-400:])
