#!/usr/bin/env python
"""Only the caption leg of bench.py (cached greedy decoding of config B's model, batch 32, 32 new tokens), for rocprofv3:
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python tools/caption_profile.py [--eager]
`--eager`: model.decode_graph = False (every decode step launched kernel by kernel - what a kernel trace needs to see per-step launches);
`--tweaks`: with the op substitutions inside the backbones and the tuning file (bench.py --backbone-tweaks on)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
eager, tweaks = "--eager" in sys.argv, "--tweaks" in sys.argv
import torch
import bench

sys.argv = ["bench.py"] + (["--backbone-tweaks", "on"] if tweaks else [])
a = bench.parse()
dev = torch.device("cuda", 0)
if tweaks:
    from flamingo_mini_amd.backbones import load_stock_gemm_tuning
    load_stock_gemm_tuning()
model, cfg = bench.build_model(a, dev, torch.bfloat16)
model.decode_graph = not eager
batch = bench.synthetic_batch(a, cfg, dev, torch.bfloat16, 0)
model.eval()
ids, ml, am = batch["input_ids"][:, :4], batch["media_locations"][:, :4], batch["attention_mask"][:, :4]
with torch.no_grad():
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.greedy_generate(ids, ml, am, pixel_values=batch["pixel_values"], max_length=4 + 32)
        torch.cuda.synchronize()
        print(f"rep {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms for {out.shape[1] - 4} tokens x {ids.shape[0]}", flush=True)
