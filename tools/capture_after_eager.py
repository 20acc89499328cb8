#!/usr/bin/env python
"""Does building a GraphedTrainStep after an eager forward of the same model (whose autograd graph is still alive) break the capture?
(round 3, session 1: a segmentation fault in capture_end with exactly that sequence)   python tools/capture_after_eager.py keep|drop|nograd"""
import gc, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import torch
from test_hip_backbones import _batch, _build
from flamingo_mini_amd import FusedAdamW, GraphedTrainStep

mode = sys.argv[1]
dtype = torch.float32
model = _build(False, dtype)
batch = _batch(dtype)
if mode == "nograd":
    with torch.no_grad():
        out = model(**batch)
else:
    out = model(**batch)
print("eager forward done, loss", float(out.loss), flush=True)
if mode == "drop":
    del out
    gc.collect()
opt = FusedAdamW(list(model.parameters_trainable()), lr=1e-4, capturable=True)
step = GraphedTrainStep(model, opt, batch, warmup=1)
print("captured; replays:", [round(float(step()), 4) for _ in range(3)], flush=True)
