#!/usr/bin/env python
"""Per-step training loss of bench.py's workload in its four launch modes (substituted / stock backbones x graph replay / eager launches),
same seed, same batch every step (what bench.py does): the evidence behind DESIGN.md's note on `config.loss` (13.84 vs 7.60).

    python tools/loss_trajectory.py --steps 24 [--config B] [--dropout default|0] > gpurun_out/loss_trajectory.json
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import torch
    import bench
    sys.argv = ["bench.py", "--config", args.config, "--backbone-tweaks", args.tweaks] + ([] if args.lm_dropout is None else ["--lm-dropout", str(args.lm_dropout)])
    a = bench.parse()
    device = torch.device("cuda", 0)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    if a.stock_tuning == "on" and a.dtype == "bf16":
        from flamingo_mini_amd.backbones import load_stock_gemm_tuning
        load_stock_gemm_tuning()
    model, cfg = bench.build_model(a, device, dtype)
    batch = bench.synthetic_batch(a, cfg, device, dtype, 0)
    params = list(model.parameters_trainable())
    from flamingo_mini_amd import FusedAdamW, GraphedTrainStep
    kw = dict(master_dtype=torch.float32) if args.optimizer == "master" else {}
    opt = FusedAdamW(params, lr=1e-4, capturable=args.graph == "on", **kw)
    losses = []

    def eager():
        for p in params:
            p.grad = None
        loss = model(**batch).loss
        loss.backward()
        opt.step()
        return loss

    if args.graph == "on":
        step = GraphedTrainStep(model, opt, batch, warmup=1)
        losses.append(None)                      # (the warm-up step inside the constructor is step 0)
        for _ in range(args.steps - 1):
            losses.append(round(float(step()), 4))
    else:
        for _ in range(args.steps):
            losses.append(round(float(eager()), 4))
    print(json.dumps(dict(tweaks=args.tweaks, graph=args.graph, dropout=args.lm_dropout if args.lm_dropout is not None else "default", optimizer=args.optimizer,
                          losses=losses)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--config", default="B")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--tweaks", default="on")
    ap.add_argument("--graph", default="on")
    ap.add_argument("--optimizer", default="bf16", choices=["bf16", "master"])
    ap.add_argument("--modes", default="on:on,on:off,off:on,off:off", help="comma-separated tweaks:graph pairs")
    ap.add_argument("--lm-dropout", type=float, default=None)
    ap.add_argument("--dropouts", default="default,0", help="LM dropout settings to run every mode with (default = the HF config's 0.1)")
    args = ap.parse_args()
    if args.child:
        return child(args)
    for drop in args.dropouts.split(","):
        for mode in args.modes.split(","):
            tw, gr = mode.split(":")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--steps", str(args.steps), "--config", args.config,
                                "--tweaks", tw, "--graph", gr, "--optimizer", args.optimizer] + ([] if drop == "default" else ["--lm-dropout", drop]),
                               capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else json.dumps(dict(tweaks=tw, graph=gr, dropout=drop, error=r.stderr[-400:])), flush=True)


if __name__ == "__main__":
    main()
