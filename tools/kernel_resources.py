#!/usr/bin/env python
"""Register / scratch / occupancy table of every kernel of the library, from the compiler's own resource remarks
(-Rpass-analysis=kernel-resource-usage; build.py keeps them per translation unit under csrc/_obj/*.resources.txt).
    python tools/kernel_resources.py [pattern]          e.g.  python tools/kernel_resources.py gemm_bf16_pc_kernel"""
import glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return out.strip().splitlines()
    except Exception:
        return list(names)


def parse(text):
    """[{name, vgprs, agprs, sgprs, scratch, occupancy, lds}] in file order"""
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            cur = {"mangled": v}
            rows.append(cur)
        elif cur is not None:
            cur[{"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs"}.get(k, "scratch" if k.startswith("Scratch") else "occupancy" if k.startswith("Occ") else "lds")] = int(v)
    for r, n in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = re.sub(r"\(.*", "", n.replace("void ", ""))
    return rows


def load():
    rows = []
    for f in sorted(glob.glob(os.path.join(ROOT, "flamingo-mini_amd", "csrc", "_obj", "*.resources.txt"))):
        for r in parse(open(f).read()):
            r["unit"] = os.path.basename(f).split(".")[0]
            rows.append(r)
    return rows


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    print(f"{'kernel':100s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'waves/SIMD':>10s}")
    for r in load():
        if pat in r["name"]:
            print(f"{r['name'][:100]:100s} {r.get('vgprs', 0):5d} {r.get('agprs', 0):5d} {r.get('sgprs', 0):5d} {r.get('scratch', 0):8d} {r.get('occupancy', 0):10d}")
