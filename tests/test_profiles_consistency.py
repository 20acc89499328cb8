"""The committed measurement artefacts must agree with each other: the bench line's dominant kernel is in the rocprofv3
summary with a matching average duration, the PMC traffic file names it, and the JSON line carries the contract fields."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def _bench():
    with open(os.path.join(P, "r01_bench_default.json")) as f:
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
    with open(os.path.join(P, "r01_bench_b32_bf16_kernel_stats.csv")) as f:
        rows = {row["Name"].split("(")[0].replace("void ", ""): row for row in csv.DictReader(f)}
    assert r["kernel"] in rows, r["kernel"]
    rocprof_us = float(rows[r["kernel"]]["AverageNs"]) / 1e3
    # HIP events bracket the launch: they read 2-4 us more than the kernel's own duration, never less
    assert rocprof_us <= r["avg_launch_us"] <= rocprof_us + 5.0, (rocprof_us, r["avg_launch_us"])


def test_pmc_traffic_names_the_dominant_kernel():
    r = _bench()["roofline"]
    with open(os.path.join(P, "r01_pmc_traffic.json")) as f:
        t = json.load(f)["kernels"]
    assert t[r["kernel"]]["hbm_bytes_per_launch"] == r["traffic"]
    assert t[r["kernel"]]["hbm_bytes_per_launch"] > 26e6 * 0.9   # >= the 26 MB the operands and the output occupy
