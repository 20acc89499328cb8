"""The committed measurement artefacts of the latest round (profiles/rNN_*) must agree with each other: the bench line's dominant
kernel is in the rocprofv3 summary OF THE SAME SESSION with a matching average duration and launch count per step, the PMC traffic
file names it, and the JSON line carries the contract fields (tools/gpu_final.sh + tools/collect_profiles.py produce all of them)."""
import csv
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
RND = sorted(re.match(r"(r\d+)_", os.path.basename(p)).group(1) for p in glob.glob(os.path.join(P, "r*_bench_default.json")))[-1]


def _bench():
    with open(os.path.join(P, f"{RND}_bench_default.json")) as f:
        return json.load(f)


def test_bench_line_has_contract_fields():
    d = _bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    assert abs(d["value"] - d["config"]["global_batch"] / d["ms_per_step"] * 1e3) < 0.5


def test_rocprof_summary_agrees_with_the_event_timing():
    r = _bench()["roofline"]
    with open(os.path.join(P, f"{RND}_bench_b32_bf16_kernel_stats.csv")) as f:
        rows = {row["Name"].split("(")[0].replace("void ", ""): row for row in csv.DictReader(f)}
    assert r["kernel"] in rows, r["kernel"]
    rocprof_us = float(rows[r["kernel"]]["AverageNs"]) / 1e3
    # HIP events bracket the launch: they read a few us more than the kernel's own duration, never less (2 % slack for run-to-run noise)
    assert rocprof_us * 0.98 <= r["avg_launch_us"] <= rocprof_us + 6.0, (rocprof_us, r["avg_launch_us"])
    if RND >= "r02":   # same tree, same session: the launch count per step must agree too (profiled run: 2 eager warm-up steps + the replays)
        with open(os.path.join(P, f"{RND}_bench_b32_bf16_summary.md")) as f:
            steps_total = int(re.search(r"over (\d+) steps", f.read()).group(1))
        per_step_bench = r["launches"] / int(re.search(r"in (\d+) eager steps", r["measured"]).group(1))
        assert abs(int(rows[r["kernel"]]["Calls"]) / steps_total - per_step_bench) < 0.51, (rows[r["kernel"]]["Calls"], steps_total, per_step_bench)


def test_pmc_traffic_names_the_dominant_kernel():
    r = _bench()["roofline"]
    with open(os.path.join(P, f"{RND}_pmc_traffic.json")) as f:
        t = json.load(f)["kernels"]
    assert t[r["kernel"]]["hbm_bytes_per_launch"] == r["traffic"]
    algorithmic = r["avg_launch_gflop"] * 0 + 1      # operands + output of the launch, bf16: see DESIGN.md section 5 for the per-launch figure
    assert t[r["kernel"]]["hbm_bytes_per_launch"] > 20e6 * algorithmic


def test_roofline_table_is_regenerable_from_the_committed_profiles():
    """profiles/rNN_roofline_table.md is derived data: tools/roofline_table.py must reproduce it from the kernel statistics, the counter traffic
    and the GEMM table committed next to it (so the table cannot drift from the session it claims to describe)."""
    import glob
    import subprocess
    import sys
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_roofline_table.md")))
    if not tables:
        pytest.skip("no roofline table committed")
    path = tables[-1]
    rnd = os.path.basename(path).split("_")[0]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_table.py"), rnd], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-800:]
    assert out.stdout.strip() == open(path).read().strip()


def test_counter_traffic_is_compared_shape_by_shape():
    """profiles/rNN_gemm_traffic.txt (round 4 on): the FETCH_SIZE / WRITE_SIZE passes grouped by kernel instantiation AND grid, one row per GEMM shape of the
    step, next to that shape's algorithmic bytes - so a traffic ratio never mixes shapes.  The listed figures must follow from the row's own shape, the
    ratio from the two columns, every shape of the bench's GEMM table must be there, and the bench line's per-instantiation average must lie inside the
    range of the rows that instantiation serves."""
    path = os.path.join(P, f"{RND}_gemm_traffic.txt")
    if not os.path.exists(path):
        pytest.skip("no per-shape traffic table for this round")
    rows = [l.split() for l in open(path).read().strip().splitlines()[1:]]
    table = [l.split()[:8] for l in open(os.path.join(P, f"{RND}_gemm_table.txt")).read().strip().splitlines()[1:]]
    assert sorted(r[:8] for r in rows) == sorted(t for t in table if int(t[6]) in (128002, 128160, 64002, 3264, 256128, 256256))
    measured = 0
    for r in rows:
        M, N, K, nz, al, bl, tile, sk = (int(v) for v in r[:8])
        alg = nz * ((M * K + N * K) * 2 + (M * N * 4 * sk if sk > 1 else M * N * 2)) / 1e6
        assert abs(alg - float(r[10])) <= 0.06, r
        if r[11] == "n/a":
            continue
        measured += 1
        assert abs(float(r[11]) / float(r[10]) - float(r[12])) <= 0.03 * float(r[12]) + 0.01 and 0.5 < float(r[12]) < 4.0, r
    assert measured >= len(rows) - 2
    line = _bench()["roofline"]
    m = re.match(r"ff::gemm_bf16_pc_kernel<(\d+), (\d+), (\d), (\d),", line["kernel"])
    if m and line["traffic"]:
        bm, bn, al, bl = (int(v) for v in m.groups())
        code = {(128, 128): 128002, (128, 160): 128160, (64, 64): 64002, (256, 128): 256128}[(bm, bn)]
        mine = [float(r[11]) for r in rows if int(r[6]) == code and int(r[4]) == al and int(r[5]) == bl and r[11] != "n/a"]
        assert mine and min(mine) * 0.95 <= line["traffic"] / 1e6 <= max(mine) * 1.05, (line["traffic"], mine)


def test_hipblaslt_yardstick_covers_every_gemm_shape():
    path = os.path.join(P, f"{RND}_gemm_yardstick.txt")
    if not os.path.exists(path):
        pytest.skip("no yardstick table for this round")
    rows = [l.split() for l in open(path).read().strip().splitlines()[1:]]
    table = [l.split()[:8] for l in open(os.path.join(P, f"{RND}_gemm_table.txt")).read().strip().splitlines()[1:]]
    assert sorted(r[:8] for r in rows) == sorted(table)
    for r in rows:
        assert float(r[10]) > 0 and float(r[11]) > 0 and abs(float(r[10]) / float(r[11]) - float(r[12])) <= 0.02, r          # (both times are printed to 0.1 us)


def test_launch_modes_under_a_one_rank_exchange_are_ordered_as_the_design_says():
    """profiles/rNN_launch_modes_one_rank_rccl.txt (round 4 on): the single graph without collectives is the fastest mode; with the gradient exchange
    going through a 1-rank RCCL group the piecewise replay with host-paced collectives beats the stream-ordered one, and every replayed mode beats
    eager launches (DESIGN.md section 6 quotes these numbers).  Round 4 also had whole-step capture behind both piecewise forms: its collectives were
    `oneRankReduce<FuncPreMulSum>` kernels inside the graph (ReduceOp.AVG at one rank); since round 5 a one-rank exchange asks for an in-place SUM,
    launches nothing, and the captured step is as fast as the host-paced one - so from r05 on only `captured < eager` is held."""
    path = os.path.join(P, f"{RND}_launch_modes_one_rank_rccl.txt")
    if not os.path.exists(path):
        pytest.skip("no launch-mode table for this round")
    rows = [l for l in open(path) if l.startswith("graph=")]
    if not any(l.startswith("graph=piecewise --pace stream") for l in rows):
        pytest.skip("table predates the host-paced mode")
    no_coll = float(re.search(r"([\d.]+) ms/step", [l for l in rows if "no collectives" in l][0]).group(1))
    captured = float(re.search(r"([\d.]+) ms/step", [l for l in rows if l.startswith("graph=on, gradient")][0]).group(1))
    host = float(re.search(r"([\d.]+) ms/step", [l for l in rows if l.startswith("graph=piecewise, ")][0]).group(1))
    stream = float(re.search(r"([\d.]+) ms/step", [l for l in rows if l.startswith("graph=piecewise --pace stream")][0]).group(1))
    eager = float(re.search(r"([\d.]+) ms/step", [l for l in rows if l.startswith("graph=off")][0]).group(1))
    assert no_coll < host < stream < eager and no_coll < captured < eager, (no_coll, host, stream, captured, eager)
    if RND < "r05":
        assert stream < captured
