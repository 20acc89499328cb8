"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` run of bench.py into a per-step, per-category table.
    python tools/summarize_rocprof.py gpurun_out/prof_x/x_kernel_stats.csv --steps-total 5 > profiles/rNN_summary.md
"""
import argparse, csv, re


INIT = "one-time: random initialisation of the model (normal_ fills, the embedding resize's Cholesky) - not part of a step"


def category(name):
    # the profiled process also builds the model: its random init runs ~850 normal_ kernels and rocsolver's small Cholesky (mean-resizing of the
    # embedding for the <EOC> row) once; they are listed apart and left out of the per-step total
    if "normal_and_transform" in name or "rocsolver" in name or "potf2" in name: return INIT
    if "ff::gemm_bf16" in name or "ff16gemm" in name or "gemm_f32" in name: return "fusion: GEMM main kernels (hand-written MFMA)"
    if "gemm_splitk" in name: return "fusion: split-K reduce + epilogue"
    if "decode_rows32" in name: return "fusion: decode-shaped feed-forward (weight streaming, <= 32 rows)"
    if "xa_qattn" in name or "xa_dattn" in name: return "fusion: LayerNorm + projection + attention of the gated blocks (one launch each way)"
    if "attn_fwd_kernel" in name or "attn_bwd" in name: return "fusion: attention core of the resampler (fwd, dQ, dK/dV)"
    if "adamw_kernel" in name: return "fusion: multi-tensor AdamW (ff_adamw_step)"
    if "shifted_ce" in name: return "fusion: shifted cross-entropy (fwd + bwd)"
    if "quick_gelu" in name: return "fusion: QuickGELU of the CLIP MLPs"
    if "ff::" in name or "_ZN2ff" in name: return "fusion: LayerNorm / reductions / gates"
    if "Cijk_" in name: return "stock: hipBLASLt GEMMs (CLIP, GPT-2, lm_head)"
    if "multi_tensor_apply" in name: return "stock: fused AdamW"
    if name in ("attn_fwd", "bwd_kernel_fuse", "bwd_kernel_dk_dv", "bwd_kernel_dq") or "attn" in name.lower(): return "stock: SDPA attention (CLIP, GPT-2)"
    if "elementwise" in name or "reduce_kernel" in name or "layer_norm" in name or "softmax" in name.lower() or "index" in name or "nll" in name: return "stock: elementwise / norm / loss"
    return "stock: other (copies, fills, conv, ...)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--steps-total", type=int, required=True, help="warmup + timed steps of the profiled bench.py run")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    init_ns = sum(int(r["TotalDurationNs"]) for r in rows if category(r["Name"]) == INIT)
    init_calls = sum(int(r["Calls"]) for r in rows if category(r["Name"]) == INIT)
    rows_all, rows = rows, [r for r in rows if category(r["Name"]) != INIT]
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    cats = {}
    for r in rows:
        c = cats.setdefault(category(r["Name"]), [0, 0])
        c[0] += int(r["TotalDurationNs"]); c[1] += int(r["Calls"])
    print(f"# rocprofv3 kernel-trace summary ({a.csv})\n")
    print(f"GPU-busy time {tot / 1e6:.1f} ms over {a.steps_total} steps = **{tot / 1e6 / a.steps_total:.2f} ms/step**, {sum(int(r['Calls']) for r in rows) // a.steps_total} kernel launches/step\n")
    print(f"(left out: {init_ns / 1e6:.1f} ms in {init_calls} launches of {INIT})\n")
    print("| category | ms/step | share | launches/step |\n|---|---:|---:|---:|")
    for k, (ns, n) in sorted(cats.items(), key=lambda kv: -kv[1][0]):
        print(f"| {k} | {ns / 1e6 / a.steps_total:.2f} | {100 * ns / tot:.1f}% | {n // a.steps_total} |")
    lib_ns = sum(ns for k, (ns, n) in cats.items() if k.startswith("fusion"))
    lib_n = sum(n for k, (ns, n) in cats.items() if k.startswith("fusion"))
    print(f"\nfusion library total: **{lib_ns / 1e6 / a.steps_total:.2f} ms/step in {lib_n // a.steps_total} launches/step**; stock backbones + loss glue: {(tot - lib_ns) / 1e6 / a.steps_total:.2f} ms/step")
    print(f"\n## top {a.top} kernels\n\n| kernel | calls | avg us | total ms/step | share |\n|---|---:|---:|---:|---:|")
    for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[: a.top]:
        name = re.sub(r"\(.*", "", r["Name"])[:110]
        print(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {int(r['TotalDurationNs']) / 1e6 / a.steps_total:.2f} | {r['Percentage']}% |")


if __name__ == "__main__":
    main()
