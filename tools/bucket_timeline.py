#!/usr/bin/env python
"""When, inside backward, does each gradient bucket of the data-parallel exchange become final?  (budget table)

Input: the kernel trace of eager training steps run with the data-parallel launch structure (weight-gradient groups of 4 blocks, one
K / V projection call per 4 layers):
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -- \
        python bench.py --graph off --steps 2 --warmup 2 --no-cpu-baseline --caption-tokens 0 --profile-steps 0 --companions off
    python tools/bucket_timeline.py <dir>/**/*kernel_trace.csv
The last complete step of the trace is taken.  Bucket-final events: the end of every cluster of weight-gradient launches (both operands
M-major: gemm_bf16_*<.., 1, 1, ..>) - the grouped weight gradients of 4 gated blocks, of 4 layers' to_kv, of the resampler's layers - and the
end of the backward pass (token embedding).  Times are relative to the start of backward (the shifted cross-entropy backward kernel)."""
import csv
import glob
import sys

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ce = [i for i, n in enumerate(names) if "shifted_ce_bwd" in n]
adam = [i for i, n in enumerate(names) if "adamw_kernel" in n]
assert len(ce) >= 1 and adam, "no complete step in the trace"
start = ce[-1]
end = min(i for i in adam if i > start)
t0 = int(rows[start]["Start_Timestamp"])
t_end = int(rows[end]["Start_Timestamp"])
fwd_start = None
for i in range(start, -1, -1):          # the step's first library kernel: text_time
    if "text_time_kernel" in names[i]:
        fwd_start = int(rows[i]["Start_Timestamp"])
        break


def is_wgrad(n):
    return ("gemm_bf16_pc_kernel<128, 128, 1, 1" in n) or ("gemm_bf16_dma_kernel<64, 64, 1, 1" in n) or ("gemm_bf16_dma_kernel<128, 128, 1, 1" in n)


clusters, cur = [], None
for i in range(start, end):
    if is_wgrad(names[i]):
        if cur is None:
            cur = [i, i, 0]
        cur[1] = i
        cur[2] += 1
    elif cur is not None and "ln_bwd_final" not in names[i] and "splitk_epilogue" not in names[i]:
        clusters.append(cur)
        cur = None
if cur is not None:
    clusters.append(cur)
print(f"backward {(t_end - t0) / 1e6:.2f} ms" + (f" (forward {(t0 - fwd_start) / 1e6:.2f} ms before it)" if fwd_start else "") + f", {len(clusters)} clusters of weight-gradient launches")
print("cluster  launches  final at (ms after backward start)  ms before the optimizer")
for k, (a, b, n) in enumerate(clusters):
    te = int(rows[b]["End_Timestamp"])
    print(f"{k:7d} {n:9d} {(te - t0) / 1e6:12.2f} {(t_end - te) / 1e6:30.2f}")
