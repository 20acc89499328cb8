"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace) into HBM bytes per launch of each
fusion-library kernel:   python tools/pmc_traffic.py <FETCH counter_collection.csv> <WRITE counter_collection.csv> > profiles/rNN_pmc_traffic.json
Counter values are KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section), so
hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Memory-side counters: Infinity-Cache hits are included."""
import csv, json, sys
from collections import defaultdict


def load(path):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "ff::" not in name and "_ZN2ff" not in name:
                continue
            name = name.split("(")[0].replace("void ", "")
            a = acc[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {"_note": __doc__.split("\n", 2)[2].strip().replace("\n", " "), "kernels": {}}
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    n, f_kb = fetch[k]
    w_kb = write.get(k, [0, 0.0])[1] / max(write.get(k, [1, 0.0])[0], 1)
    f_avg = f_kb / n
    out["kernels"][k] = {"launches": n, "fetch_kb_avg": round(f_avg, 1), "write_kb_avg": round(w_kb, 1),
                         "hbm_bytes_per_launch": int((2 * f_avg + w_kb) * 1024)}
print(json.dumps(out, indent=1))
