#!/usr/bin/env python
"""Building a GraphedTrainStep after an eager forward of the same model.   python tools/capture_after_eager.py keep|drop|nograd
Round 3, sessions 1 and 14: with the autograd graph of that forward still alive - through its outputs, or through the conditioning the
hooks used to keep - the capture ended in a segmentation fault (AccumulateGrad nodes bound to the default stream inside a stream capture).
Now: `drop` and `nograd` capture and replay; `keep` is refused by GraphedTrainStep with an explanation (exit code 3)."""
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
try:
    step = GraphedTrainStep(model, opt, batch, warmup=1)
except RuntimeError as e:
    print("refused:", e, flush=True)
    sys.exit(3)
print("captured; replays:", [round(float(step()), 4) for _ in range(3)], flush=True)
