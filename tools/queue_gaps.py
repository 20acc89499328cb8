#!/usr/bin/env python
"""Where does a replayed training step idle?  (the ~1.5 ms per step that "RCCL's presence" costs at one rank.)

Input: rocprofv3 --kernel-trace CSVs of the same step in two arms, e.g. the piecewise replay with and without a 1-rank RCCL exchange
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- python bench.py --graph piecewise [--force-collectives] --steps 4 --warmup 2 ...
    python tools/queue_gaps.py <dir A>/**/*kernel_trace.csv <dir B>/**/*kernel_trace.csv
For every arm, over its last complete steps (a step starts at the library's text_time kernel): wall time per step, kernels and busy time per
hardware queue, and - for the queue that carries the step - the idle time between consecutive kernels, binned by gap length.  The difference
between the arms' bins says whether the extra time is many slightly longer dispatch gaps or a few long waits, and after which kernels."""
import collections
import csv
import sys

BINS = [(0, 1), (1, 2), (2, 4), (4, 8), (8, 20), (20, 100), (100, 1e9)]


def load(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    return rows


def analyse(path):
    rows = load(path)
    qkey = next((k for k in ("Queue_Id", "Queue_ID", "queue_id") if k in rows[0]), None)
    starts = [i for i, r in enumerate(rows) if "text_time_kernel" in r["Kernel_Name"]]
    assert len(starts) >= 3, f"{path}: fewer than two complete steps in the trace"
    lo, hi = starts[-3], starts[-1]                 # the last two complete steps
    steps = 2
    seg = rows[lo:hi]
    wall = (rows[hi]["s"] - rows[lo]["s"]) / 1e6 / steps
    queues = collections.defaultdict(list)
    for r in seg:
        queues[r[qkey] if qkey else "0"].append(r)
    main = max(queues, key=lambda q: len(queues[q]))
    out = {"path": path, "wall_ms": wall, "queues": {}, "main": main}
    for q, rs in queues.items():
        out["queues"][q] = (len(rs) / steps, sum(r["e"] - r["s"] for r in rs) / 1e6 / steps,
                            collections.Counter(r["Kernel_Name"].split("(")[0][:60] for r in rs).most_common(3))
    rs = queues[main]
    bins = [[0, 0.0] for _ in BINS]
    after = collections.defaultdict(float)
    prev_end = rs[0]["e"]
    for a, b in zip(rs, rs[1:]):
        gap = (b["s"] - prev_end) / 1e3         # us; kernels of one queue run in order, overlap shows as a negative gap
        prev_end = max(prev_end, b["e"])
        if gap <= 0:
            continue
        for k, (x, y) in enumerate(BINS):
            if x <= gap < y:
                bins[k][0] += 1
                bins[k][1] += gap
        if gap >= 8:
            after[a["Kernel_Name"].split("(")[0][:70]] += gap
    out["bins"] = [(c / steps, t / 1e3 / steps) for c, t in bins]
    out["busy_ms"] = sum(r["e"] - r["s"] for r in rs) / 1e6 / steps
    out["idle_ms"] = sum(t for _, t in out["bins"])
    out["after"] = sorted(((v / 1e3 / steps, k) for k, v in after.items()), reverse=True)[:8]
    return out


def main():
    arms = [analyse(p) for p in sys.argv[1:]]
    for a in arms:
        print(f"== {a['path']}")
        print(f"   wall {a['wall_ms']:.2f} ms per step; main queue {a['main']}: busy {a['busy_ms']:.2f} ms, idle between its kernels {a['idle_ms']:.2f} ms")
        for q, (n, busy, top) in sorted(a["queues"].items(), key=lambda kv: -kv[1][0]):
            print(f"   queue {q}: {n:.0f} kernels per step, {busy:.2f} ms busy; most frequent: " + "; ".join(f"{c // 2}x {k}" for k, c in top))
        print("   idle gaps of the main queue, per step:   " + "   ".join(f"[{x}-{'inf' if y > 1e8 else y} us) {c:.0f} gaps {t:.2f} ms" for (x, y), (c, t) in zip(BINS, a["bins"])))
        print("   gaps >= 8 us follow: " + "; ".join(f"{t:.2f} ms after {k}" for t, k in a["after"]))
    if len(arms) == 2:
        a, b = arms
        print(f"== difference (second - first): wall {b['wall_ms'] - a['wall_ms']:+.2f} ms, main-queue busy {b['busy_ms'] - a['busy_ms']:+.2f} ms, idle {b['idle_ms'] - a['idle_ms']:+.2f} ms")
        print("   by gap length: " + "   ".join(f"[{x}-{'inf' if y > 1e8 else y}) {cb - ca:+.0f} gaps {tb - ta:+.2f} ms" for (x, y), (ca, ta), (cb, tb) in zip(BINS, a["bins"], b["bins"])))


if __name__ == "__main__":
    main()
