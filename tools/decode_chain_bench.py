#!/usr/bin/env python
"""The fusion library's share of one cached decode step in isolation: the 36 gated cross-attention blocks of flamingo-mini's geometry
(dim 1280, 8 x 64 heads, 64 cached keys, ff_mult 4, bf16), one token per sequence at batch 32, called back to back on persistent K / V -
what FlamingoModel's decode step issues between the stock LM blocks.  Every block has its own weights (1.04 GB in total: nothing stays in
the 256 MiB Infinity Cache from one step to the next).

    python tools/decode_chain_bench.py                 # HIP-graph replay: ms per chain, GB/s of weight + K / V bytes
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python tools/decode_chain_bench.py --eager     # per-kernel averages
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--dim", type=int, default=1280)
    ap.add_argument("--layers", type=int, default=36)
    ap.add_argument("--reps", type=int, default=40)
    a = ap.parse_args()
    from flamingo_mini_amd import GatedCrossAttentionBlock
    dt, dev = torch.bfloat16, "cuda"
    torch.manual_seed(0)
    blocks = [GatedCrossAttentionBlock(dim=a.dim, dim_visual=1024, heads=8, dim_head=64, ff_mult=4, n_visual=64).to(device=dev, dtype=dt) for _ in range(a.layers)]
    with torch.no_grad():
        for b in blocks:
            b.alpha_attn.fill_(0.5); b.alpha_ffw.fill_(0.5)
    past = [(torch.randn(a.batch, 8, 64, 64, device=dev, dtype=dt), torch.randn(a.batch, 8, 64, 64, device=dev, dtype=dt)) for _ in blocks]
    y0 = torch.randn(a.batch, 1, a.dim, device=dev, dtype=dt)
    tt = torch.ones(a.batch, 1, dtype=torch.int32, device=dev)
    nbytes = sum(p.numel() for b in blocks for n, p in b.named_parameters() if "to_kv" not in n) * 2 + sum(k.numel() + v.numel() for k, v in past) * 2

    def chain():
        h = y0
        for blk, kv in zip(blocks, past):
            h, _ = blk(h, None, None, previous_kv=kv, output_kv=False, text_time=tt)
        return h

    with torch.no_grad():
        if a.eager:
            for _ in range(a.reps):
                chain()
            torch.cuda.synchronize()
            print(f"eager: {a.reps} chains of {a.layers} blocks issued")
            return
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chain(); chain()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = chain()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        e0.record()
        for _ in range(a.reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
    print(f"decode chain: {a.layers} blocks, batch {a.batch}, dim {a.dim}: {ms:.3f} ms per token step ({ms / a.layers * 1e3:.1f} us per block), "
          f"{nbytes / 1e9:.3f} GB of weights + cached K / V -> {nbytes / (ms * 1e-3) / 1e12:.2f} TB/s ({nbytes / (ms * 1e-3) / 8e12:.3f} of the 8 TB/s peak); "
          f"finite: {bool(torch.isfinite(out.float()).all())}")


if __name__ == "__main__":
    main()
