#!/usr/bin/env python
"""Headline benchmark: images/sec (fwd+bwd) of flamingo-mini (gpt2-large + CLIP ViT-L/14), per-GPU batch 32, bf16, on
1..8 MI355X (BASELINE.json `metric`, config B of SURVEY.md section 8d).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 10 --warmup 3

One step = CLIP forward (frozen) -> perceiver resampler -> 36 x (gated xattn block + frozen GPT-2 block) -> lm_head ->
shifted cross-entropy -> backward -> mean all-reduce of the trainable gradients (N > 1) -> fused AdamW step.
Synthetic data (N(0,1) pixels, uniform token ids, media tag at position 0), random-init weights of the named
architectures (no network), gates alpha = 0.5 (zero gates would make the fusion path an identity with zero gradients).
The resampler / xattn blocks run in libflamingo_fusion (hand-written HIP); the run aborts if that library is missing.

The step (forward + backward + gradient exchange + optimizer) is captured once into a HIP graph and the timed region replays it
(`--graph off`: eager launches; `--graph piecewise`: one sub-graph per backward segment with the RCCL collectives issued eagerly
between the replays - the path that does not depend on collectives being capturable).
The headline `value` is the north star's configuration: the frozen CLIP / GPT-2 stay UNTOUCHED Hugging Face modules on stock
PyTorch-ROCm with hipBLASLt's default heuristic.  `--backbone-tweaks on` enables three result-identical op substitutions inside them
plus the pre-tuned hipBLASLt solution file for their GEMMs (flamingo-mini_amd/tuning/); the default run reports that configuration too,
as the companion `with_backbone_op_substitutions`.

Rank 0 prints ONE JSON line.  `roofline` is measured live: right after the timed region `--profile-steps` eager steps of
the same workload run with every GEMM / attention launch of the fusion library bracketed by HIP events on its stream
(ff_gemm_profile_*; events cannot be recorded inside a replayed graph); the dominant GEMM variant is reported against
the dense bf16 MFMA peak.  `cpu_baseline` times the all-core torch CPU restatement (oracle/torch_port.py) of the same hot path on
this box's host cores (and config A end to end).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16 (AMD's 5 PF headline is 2:1 sparse)
MFMA_F32_PEAK_TFLOPS = 157.3


# The five workloads of BASELINE.json `configs` as concrete synthetic inputs.  B is the headline (the config the
# metric is quoted on); the others are reached with --config and print the same JSON contract with their own `config.workload`.
CONFIGS = {
    "A": dict(lm="gpt2", clip="openai/clip-vit-base-patch32", batch=2, seq_len=32, xattn_every=1, images=1, frames=0, dtype="f32",
              what="flamingo-tiny (gpt2 124M + CLIP ViT-B/32), 1 image, seq_len 32, batch 2 (the reference's CPU-runnable plumbing case, here on the GPU in fp32)"),
    "B": dict(lm="gpt2-large", clip="openai/clip-vit-large-patch14", batch=32, seq_len=32, xattn_every=1, images=1, frames=0, dtype="bf16",
              what="flamingo-mini (gpt2-large + CLIP ViT-L/14), 1 image (224x224) + 32 tokens per sequence"),
    "C": dict(lm="facebook/opt-1.3b", clip="openai/clip-vit-large-patch14", batch=32, seq_len=32, xattn_every=2, images=1, frames=0, dtype="bf16",
              what="facebook/opt-1.3b + CLIP ViT-L/14, xattn_every=2 (12 gated blocks), 1 image + 32 tokens per sequence (global batch 256 = 32 per GPU x 8)"),
    "D": dict(lm="gpt2-large", clip="openai/clip-vit-large-patch14", batch=32, seq_len=32, xattn_every=1, images=1, frames=4, dtype="bf16",
              what="video path: gpt2-large + CLIP ViT-L/14, one 4-frame clip (4 x 224x224, resampler_num_time_embeds=4, 1092 resampler keys) + 32 tokens per sequence"),
    "E": dict(lm="facebook/opt-6.7b", clip="openai/clip-vit-large-patch14", batch=4, seq_len=1024, xattn_every=1, images=4, frames=0, dtype="bf16",
              what="facebook/opt-6.7b + CLIP ViT-L/14, few-shot interleaved: 4 images + 1024 tokens per sequence, media tags at 0/256/512/768"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="B", choices=sorted(CONFIGS), help="BASELINE.json workload (B = headline); the flags below override its fields")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling)")
    ap.add_argument("--seq-len", type=int, default=None)
    ap.add_argument("--lm", default=None)
    ap.add_argument("--clip", default=None)
    ap.add_argument("--xattn-every", type=int, default=None)
    ap.add_argument("--images", type=int, default=None, help="images per sequence (media tags spread evenly over the sequence)")
    ap.add_argument("--frames", type=int, default=None, help="frames per image (0 = still images, 5-D pixel tensor; > 0 = 6-D video tensor)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32"])
    ap.add_argument("--no-optimizer", action="store_true", help="time fwd+bwd(+all-reduce) only")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "fused-master", "sharded", "sharded-master", "torch"],
                    help="fused = ff_adamw_step on bf16 parameters with bf16 moments; fused-master = the same kernel with fp32 master weights and fp32 "
                         "moments (the reference's --fp16 recipe, training/train.sh:24); sharded(-master) = data_parallel.ShardedAdamW (reduce-scatter -> "
                         "update of this rank's 1/N slice -> all-gather per gradient bucket, replaces the gradient all-reduce); torch = torch.optim.AdamW(fused=True)")
    ap.add_argument("--reduce-dtype", default="native", choices=["native", "f32"],
                    help="N > 1 with the all-reduce path: exchange gradient buckets in their own dtype (bf16, ReduceOp.AVG) or widened to fp32")
    ap.add_argument("--companions", default="auto", choices=["auto", "on", "off"],
                    help="N=1 only: after the main measurement, re-run the same workload in child processes (a) with the op substitutions inside the "
                         "backbones + the tuning file, (b) with the fp32-master / fp32-moment optimizer, and add both to the JSON line (auto = on for the default config B run)")
    ap.add_argument("--segment-arena", default="on", choices=["on", "off"],
                    help="piecewise replay with a reducer: the gradient buckets of a backward segment live in one buffer and travel as ONE collective (11 per step instead of 47)")
    ap.add_argument("--overlap-optimizer", default="auto", choices=["auto", "on", "off"],
                    help="piecewise replay with host pacing: AdamW as one sub-graph per backward segment, launched on a side stream as soon as the segment "
                         "(and its collectives) are done, beside the backward of the layers below.  auto = on when the step exchanges gradients (43.0 -> 42.6 ms "
                         "with a 1-rank RCCL group; on one GPU without collectives the single whole-step graph is as fast: 40.9 ms either way)")
    ap.add_argument("--pace", default="host", choices=["host", "stream"],
                    help="piecewise replay: host = the host waits for a backward segment and then issues its collectives (default; no barrier packet waits in a "
                         "hardware queue beside the compute queue), stream = collectives enqueued right behind the segment's graph, ordered by an event wait")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off", "piecewise"],
                    help="replay the step from captured HIP graphs.  piecewise = one sub-graph per backward segment of 4 gated layers, the RCCL collectives "
                         "issued eagerly between the replays; on = the whole step incl. its collectives in ONE graph; auto = on at N = 1, piecewise -> on -> "
                         "eager launches at N > 1 (every rank takes the first mode that ALL ranks can capture)")
    ap.add_argument("--force-collectives", action="store_true", help="N=1: run the gradient exchange through a 1-rank RCCL group (what a single GPU can exercise of the N > 1 path)")
    ap.add_argument("--bucket-timeline", action="store_true", help="after the timed region, one eager step with HIP events around every gradient "
                                                                   "bucket's exchange: when it became ready, when its collective finished (rank 0, `bucket_timeline` in the JSON line)")
    ap.add_argument("--segment-layers", type=int, default=4,
                    help="piecewise replay: gated layers per backward segment (= per exchange point; with --wgrad-group / --kv-group set to the same number the "
                         "launch structure follows it)")
    ap.add_argument("--wgrad-group", type=int, default=0, help="gated layers per grouped weight-gradient launch (0 = default: 12, or 4 with collectives)")
    ap.add_argument("--kv-group", type=int, default=-1, help="gated layers per K / V projection call (-1 = default: all, or 4 with collectives)")
    ap.add_argument("--resampler-layerwise", default="auto", choices=["auto", "on", "off"],
                    help="the resampler as one library call per layer (ff_resampler_layer_*: one gradient bucket per layer) instead of the stack-level call; "
                         "auto = on with gradient collectives, off on one GPU")
    ap.add_argument("--sync-exchange", default="on", choices=["on", "off"],
                    help="off: the gated blocks keep to_out (+ gate + residual) and d LN(y) as launches of their own instead of running them inside the fused "
                         "attention launches through the in-launch exchange (ff_xattn_desc.sync; A/B timing)")
    ap.add_argument("--lm-dropout", type=float, default=None, help="debugging aid: dropout probability inside the stock LM (default: the architecture's)")
    ap.add_argument("--profile-steps", type=int, default=3, help="instrumented eager steps after the timed region (roofline objects)")
    ap.add_argument("--stock-tuning", default="auto", choices=["auto", "on", "off"],
                    help="load the pre-tuned hipBLASLt solution file for the stock CLIP / GPT-2 GEMMs (PyTorch TunableOp, tuning off); auto = with --backbone-tweaks on")
    ap.add_argument("--backbone-tweaks", default="off", choices=["on", "off"],
                    help="off (default, the north star's configuration) = untouched Hugging Face CLIP / LM on stock PyTorch-ROCm; on = fused QuickGELU, "
                         "patch-conv-as-matmul, fused tanh-GELU module inside them and the pre-tuned stock-GEMM file")
    ap.add_argument("--debug-phases", action="store_true", help="print host-side issue time of each phase of 5 eager steps and exit")
    ap.add_argument("--hoist-kv", default="on", choices=["on", "off"],
                    help="project K / V of all cross-attention layers up front in grouped launches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rccl-channels", type=int, default=0,
                    help="N > 1: cap RCCL's channels (NCCL_MAX_NCHANNELS; one workgroup per channel on the side stream).  0 = RCCL's choice.  The step's exchange "
                         "(1.37 GB per rank) needs ~8 ms of a ~25 ms backward at full xGMI speed, while every CU a collective occupies turns a 256-tile "
                         "one-tile-per-CU GEMM launch into two waves: the first knob to sweep on a multi-GPU node (DESIGN.md section 6)")
    ap.add_argument("--shared-gpu-rehearsal", action="store_true",
                    help="N > 1 ranks on a box with one GPU: all ranks on device 0, gloo instead of RCCL - a rehearsal of the multi-rank launch contract, not a measurement")
    ap.add_argument("--child", action="store_true", help="internal: a companion run started by the main process (timed region only)")
    ap.add_argument("--caption-tokens", type=int, default=32, help="N=1 only: also time cached greedy decoding of this many tokens per image (0 = skip)")
    ap.add_argument("--gemm-table", default="", help="write the per-shape GEMM timing table (measured inside the timed steps) to this file")
    args = ap.parse_args()
    if args.stock_tuning == "auto":
        args.stock_tuning = args.backbone_tweaks
    if args.child:
        args.no_cpu_baseline, args.caption_tokens, args.profile_steps, args.companions = True, 0, 0, "off"
    for k, v in CONFIGS[args.config].items():
        if k != "what" and getattr(args, k) is None:
            setattr(args, k, v)
    args.what = CONFIGS[args.config]["what"]
    return args


def hot_path_flops(args, dim, dim_visual, clip_tokens, n_blocks) -> float:
    """ALGORITHMIC fwd+bwd FLOPs of the fusion path per step and GPU, SURVEY.md section 8 row d3 (FLOP = 2 MAC, GEMM + attention contractions
    only, fwd+bwd = 3 x fwd; the flamingo defaults: 64 latents, 8 heads of 64, ff_mult 4, resampler depth 6): 1591.7 GF fwd = 4.78 TF at config B."""
    q, h, dh, inner, depth, ffm = 64, 8, 64, 512, 6, 4
    T = max(args.frames, 1)
    f = T * clip_tokens
    dv, d, L, N = dim_visual, dim, args.seq_len, args.images
    rs = depth * (2 * q * dv * inner + 4 * (f + q) * dv * inner + 4 * h * q * (f + q) * dh + 2 * q * inner * dv + 4 * q * dv * (ffm * dv))
    xa = 2 * L * d * inner + 4 * (N * q) * dv * inner + 4 * h * L * (N * q) * dh + 2 * L * inner * d + 4 * L * d * (ffm * d)
    return 3.0 * (args.batch * N * rs + args.batch * n_blocks * xa)


TOLERANCE = ("bf16 kernels: 8e-3 (outputs) / 1.2e-2 (gradients) rel-L2 per module vs the fp64 oracle, 1.5e-2 / 2.5e-2 whole model "
             "(tests/util.py; the reference's own bf16-vs-fp32 drift is 6.6e-3); fp32 kernels: logits <= 1e-4 rel of the reference "
             "(north star: 1e-3), tests/test_model_plumbing.py")


def _tile_name(code: int) -> str:
    return {128160: "128x160", 128002: "128x128 (producer/consumer)", 64002: "64x64 (producer/consumer)", 256128: "256x128", 256256: "256x256",
            3264: "32x64 (decode rows)", 3216: "32 rows x 16 columns (decode rows, weight-streaming)", 6412: "64x128"}.get(code, f"{code}x{code}" if code < 1000 else str(code))


def hot_path_summary(args, cfg, model, gemm_flops_step, gemm_ms_step, attn, prof_steps):
    """The whole fusion path against the MFMA peak: ALGORITHMIC flops per step over the library's GPU time per step.  The library
    time of a step is a rocprofv3 figure (every kernel of the library, incl. LayerNorm / reductions / AdamW / loss, which the in-process HIP-event
    log does not bracket): it is read from the latest committed summary of the same command and tagged with its source; what this run measures
    itself - the event-bracketed GEMM and attention launches - is reported next to it."""
    import glob, re
    fl = model.flamingo
    n_blocks = len(fl.get_modified_layers())
    from flamingo_mini_amd.backbones import CLIP_VISION
    patch, image = CLIP_VISION[args.clip][4:6]
    clip_tokens = (image // patch) ** 2 + 1                 # 257 for ViT-L/14 at 224, 50 for ViT-B/32
    flops = hot_path_flops(args, cfg.dim, cfg.dim_visual, clip_tokens, n_blocks)
    attn_ms = sum(v["ms"] for v in attn.values()) / max(prof_steps, 1)
    out = {"algorithmic_flops_per_step": flops, "event_bracketed_gemm_and_attention_ms": round(gemm_ms_step + attn_ms, 3),
           "frac_of_mfma_peak_over_event_bracketed_ms": round(flops / ((gemm_ms_step + attn_ms) * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
           "library_ms": None, "frac": None, "library_ms_source": "no profiles/r*_bench_b32_bf16_summary.md in the tree"}
    if args.config == "B":
        try:
            path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_b32_bf16_summary.md")))[-1]
            m = re.search(r"fusion library total: \*\*([0-9.]+) ms/step", open(path).read())
            if m:
                lib_ms = float(m.group(1))
                out.update(library_ms=lib_ms, frac=round(flops / (lib_ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                           library_ms_source=f"{os.path.relpath(path, ROOT)}: rocprofv3 --kernel-trace --stats of this command, all kernels of the fusion "
                                             "library incl. LayerNorm / reductions / AdamW / loss; not collected by this run")
        except Exception as e:
            out["library_ms_source"] = f"unreadable summary: {e!r}"[:160]
    else:
        out["library_ms_source"] = "rocprofv3 summaries are committed for config B only"
    return out


def build_model(args, device, dtype):
    from flamingo_mini_amd import FlamingoConfig, FlamingoModel
    from flamingo_mini_amd.backbones import CLIP_VISION, GPT2, OPT
    dim = GPT2[args.lm][0] if args.lm in GPT2 else OPT[args.lm][0]
    overrides = {}
    if args.lm_dropout is not None:
        d = float(args.lm_dropout)
        overrides["lm"] = dict(attn_pdrop=d, resid_pdrop=d, embd_pdrop=d) if args.lm.startswith("gpt2") else dict(dropout=d, attention_dropout=d)
    cfg = FlamingoConfig(lm=args.lm, clip_model_type=args.clip, dim=dim, dim_visual=CLIP_VISION[args.clip][0], xattn_every=args.xattn_every,
                         random_init_backbones=True, backbone_op_substitutions=args.backbone_tweaks == "on", backbone_overrides=overrides)
    torch.manual_seed(1234)                                   # same weights on every rank (DDP broadcasts; here: same seed)
    with torch.device(device):                                # random init straight on the GPU (opt-6.7b + 32 gated blocks are 11 G parameters)
        model = FlamingoModel(cfg)
    with torch.no_grad():
        for hook in model.flamingo.get_modified_layers():
            hook.xattn_block.alpha_attn.fill_(0.5)
            hook.xattn_block.alpha_ffw.fill_(0.5)
    return model.to(device=device, dtype=dtype).train(), cfg


def synthetic_batch(args, cfg, device, dtype, rank):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    shape = (args.batch, args.images) + ((args.frames,) if args.frames > 0 else ()) + (3, 224, 224)     # (b N [T] c h w)
    px = torch.randn(shape, generator=g).to(device=device, dtype=dtype)
    vocab = 50257 if args.lm.startswith("gpt2") else 50272
    ids = torch.randint(0, vocab, (args.batch, args.seq_len), generator=g).to(device)
    ml = torch.zeros((args.batch, args.seq_len), dtype=torch.long, device=device)
    ml[:, [i * (args.seq_len // args.images) for i in range(args.images)]] = 1          # one tag per image, evenly spread (0/256/512/768 in config E)
    return dict(pixel_values=px, input_ids=ids, media_locations=ml, attention_mask=torch.ones_like(ids), labels=ids)


def gemm_profile_summary(lib, ffi, max_records):
    recs = (ffi.GemmProfileRecord * max_records)()
    n = lib.ff_gemm_profile_read(recs, max_records)
    lib.ff_gemm_profile_enable(0)
    groups, shapes, attn, families = {}, {}, {}, {}
    for i in range(n):
        r = recs[i]
        if r.tile in (-4, -5, -6, -7, -8, -9):   # fused LayerNorm + projection + attention of a cross-attention block (forward / backward): every operand once
            es = 2 if r.dtype == ffi.DTYPE_BF16 else 4
            heads, dh = r.a_layout, r.split_k
            batch, inner = r.nz // heads, heads * dh
            rows_d, rows_i, kv_i, w_b = batch * r.M * r.K * es, batch * r.M * inner * es, batch * r.N * inner * es, inner * r.K * es
            # -6 / -7 (round 5): the same launches with to_out + gate + residual resp. d LN(y) = d q . Wq inside (in-launch exchange): the second
            # weight, the re-read of O / d Q by the sample's workgroups, and the outputs y1 + to_out(o) resp. d LN(y)
            nbytes = {-4: 2 * rows_d + w_b + 2 * kv_i + 2 * rows_i,            # y, yn | Wq | K, V | Qs, O
                      -5: rows_d + w_b + 3 * rows_i + 4 * kv_i,                # dy1 | Wo | Qs, O, dQ | K, V, dK, dV
                      -6: 4 * rows_d + 2 * w_b + 2 * kv_i + 3 * rows_i,        # y, yn, y1, to_out(o) | Wq, Wo | K, V | Qs, O, O again
                      -7: 2 * rows_d + 2 * w_b + 4 * rows_i + 4 * kv_i,        # dy1, d LN(y) | Wo, Wq | Qs, O, dQ, dQ again | K, V, dK, dV
                      # -8 / -9 (late in round 5): ... and the LayerNorm behind that output inside as well: LN(y1) written next to y1; d LN(y) never
                      # stored, y and d y1 read for the LayerNorm backward, d y written
                      -8: 5 * rows_d + 2 * w_b + 2 * kv_i + 3 * rows_i,
                      -9: 4 * rows_d + 2 * w_b + 4 * rows_i + 4 * kv_i}[r.tile]    # dy1, y, dy1 again, d y | ...
            a = attn.setdefault("xattn_fused_fwd" if r.tile in (-4, -6, -8) else "xattn_fused_bwd", dict(ms=0.0, bytes=0.0, launches=0))
            a["with_out_projection"] = r.tile in (-6, -7, -8, -9)
            a["with_layernorm"] = r.tile in (-8, -9)
            a["ms"] += r.ms; a["bytes"] += nbytes; a["launches"] += 1
            continue
        if r.tile < 0:      # attention core: algorithmic HBM bytes = each of Q, K, V, O (and their gradients) touched once
            es = 2 if r.dtype == ffi.DTYPE_BF16 else 4
            q_b, kv_b = r.nz * r.M * r.K * es, r.nz * r.N * r.K * es
            nbytes = {-1: 2 * q_b + 2 * kv_b, -2: 4 * q_b + 2 * kv_b, -3: 2 * q_b + 4 * kv_b}[r.tile]
            a = attn.setdefault({-1: "fwd", -2: "bwd_dq", -3: "bwd_dkv"}[r.tile], dict(ms=0.0, bytes=0.0, launches=0))
            a["ms"] += r.ms; a["bytes"] += nbytes; a["launches"] += 1
            continue
        # one group per kernel INSTANTIATION: split-K launches of the 128 x 160 tile run the 8-wave workgroup with a 3-deep ring, the others the
        # 12-wave one with a 4-deep ring (ff_gemm.hip run_bf16_dma) - two kernels in the rocprofv3 trace, two groups here
        key = (r.dtype, r.tile, r.a_layout, r.b_layout, 1 if (r.tile == 128160 and r.split_k > 1) else 0)
        for table, k in ((groups, key), (shapes, (r.M, r.N, r.K, r.nz, r.a_layout, r.b_layout, r.tile, r.split_k)), (families, (r.dtype, r.tile))):
            g = table.setdefault(k, dict(ms=0.0, flops=0.0, launches=0))
            g["ms"] += r.ms
            g["flops"] += 2.0 * r.M * r.N * r.K * r.nz
            g["launches"] += 1
    attn["_families"] = families
    return groups, shapes, attn


def caption_leg(args, model, batch, device):
    """Second half of the BASELINE metric: caption tokens/sec.  Prompt = 4 tokens with the media tag at position 0; step 1 runs
    CLIP + resampler and fills the xattn K/V + LM caches, every later step feeds one token through the cached path."""
    import torch
    model.eval()
    n_new = args.caption_tokens
    ids, ml, am = batch["input_ids"][:, :4], batch["media_locations"][:, :4], batch["attention_mask"][:, :4]
    with torch.no_grad():
        model.greedy_generate(ids, ml, am, pixel_values=batch["pixel_values"], max_length=4 + n_new)   # warm-up: builds the decode session
        # (fixed-shape caches + the captured HIP graph of one decode step, flamingo_mini_amd.modeling_flamingo._DecodeSession), which later
        # caption batches of the same shape reuse - the timed call below is such a batch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.greedy_generate(ids, ml, am, pixel_values=batch["pixel_values"], max_length=4 + n_new)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    model.train()
    new = out.shape[1] - 4
    sessions = list(getattr(model, "_decode_sessions", {}).values())
    graphed = bool(sessions) and all(s.replay is not None for s in sessions)
    # HBM roofline of one decode step: every weight a token step touches is streamed once (the LM incl. the tied lm_head, and per gated
    # block norm / to_q / to_out / ffw - to_kv is not needed, K / V are cached), plus the caches it reads: the cross-attention K / V of all
    # layers and the LM's self-attention K / V up to the current position (average over the decoded positions).
    es = next(model.parameters()).element_size()
    fl = model.flamingo
    lm_bytes = sum(p.numel() for n, p in fl.lm.named_parameters() if ".xattn_block." not in n) * es
    blocks = fl.get_modified_layers()
    blk_bytes = sum(p.numel() for h in blocks for n, p in h.xattn_block.named_parameters() if "to_kv" not in n) * es
    cfg = fl.config
    inner = cfg.xattn_heads * cfg.xattn_dim_head
    n_kv = cfg.resampler_num_latents * (batch["pixel_values"].shape[1] if batch["pixel_values"].ndim >= 5 else 1)
    xkv_bytes = len(blocks) * args.batch * n_kv * 2 * inner * es
    n_lm_layers = getattr(fl.lm.config, "n_layer", getattr(fl.lm.config, "num_hidden_layers", 0))
    avg_pos = 4 + new / 2.0
    lmkv_bytes = int(n_lm_layers * args.batch * 2 * avg_pos * cfg.dim * es)
    step_bytes = lm_bytes + blk_bytes + xkv_bytes + lmkv_bytes
    step_s = dt / new
    # The library's own share of a token step, measured directly: the 36 cached gated-block calls of one decode step (the same calls on the
    # same persistent K / V buffers, without the stock LM blocks between them) captured into a HIP graph and replayed.
    library = None
    try:
        sess = sessions[0]
        past, tt = sess.xattn_past, sess.tt_step
        y0 = torch.randn((args.batch, 1, cfg.dim), device=device, dtype=next(model.parameters()).dtype)

        def chain():
            h = y0
            for hook, kv in zip(blocks, past):
                h, _ = hook.xattn_block(h, None, None, previous_kv=kv, output_kv=False, text_time=tt)
            return h

        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                chain(); chain()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                chain()
            reps = 30
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g.replay(); torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            lib_ms = e0.elapsed_time(e1) / reps
        lib_bytes = blk_bytes + xkv_bytes
        library = {"library_ms_per_decode_step": round(lib_ms, 3), "launch_mode": "HIP graph replay of the 36 cached gated-block calls alone",
                   "library_roofline": {"bound": "hbm", "achieved": round(lib_bytes / (lib_ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                        "frac": round(lib_bytes / (lib_ms * 1e-3) / 8e12, 4), "bytes": lib_bytes}}
    except Exception as e:      # a reported extra, never a reason to lose the line
        library = {"error": repr(e)[:200]}
    return {"value": round(args.batch * new / dt, 1), "unit": "caption tokens/sec", "batch": args.batch, "new_tokens_per_image": new,
            "ms_per_decode_step": round(dt / new * 1e3, 2), "decode_step_hip_graph": graphed, "library": library,
            "roofline": {"bound": "hbm", "achieved": round(step_bytes / step_s / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(step_bytes / step_s / 8e12, 4),
                         "bytes_per_decode_step": {"lm_weights": lm_bytes, "xattn_block_weights": blk_bytes, "xattn_kv_cache": xkv_bytes, "lm_kv_cache_avg": lmkv_bytes},
                         "note": "algorithmic HBM bytes of one token step (weights streamed once + caches read) / measured step time, which includes "
                                 "the prompt / CLIP / resampler step amortised over the new tokens"},
            "note": "greedy, cached xattn K/V + static LM cache, decode steps replayed from a HIP graph when captured; includes the prompt/CLIP/resampler step"}


def run_companion(args, extra, timeout=420):
    """The same workload in a child process with other switches; returns the fields of its JSON line that matter next to the main one."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--child", "--config", args.config, "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--batch", str(args.batch), "--seq-len", str(args.seq_len), "--graph", args.graph, "--backbone-tweaks", args.backbone_tweaks] + extra
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-300:]}
        d = json.loads(line[-1])
        c = d["config"]
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "loss_first": c["loss_first"], "loss_last": c["loss_last"],
                "optimizer": c["optimizer"], "backbone_tweaks": c["backbone_tweaks"], "stock_gemm_tuning_file": c["stock_gemm_tuning_file"],
                "hip_graph": c["hip_graph"], "steps": d["steps"], "warmup": d["warmup"]}
    except Exception as e:
        return {"error": repr(e)[:300]}


def _usable_cores() -> int:
    """Host cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota (a container can see 64 CPUs and
    own 8 of them - asking torch for 64 threads then oversubscribes every core)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(args):
    """the stock-PyTorch CPU restatement of the hot path (oracle/torch_port.py - the reference itself cannot travel to
    this box), fp32, torch.set_num_threads(all host cores), 2 warm-up runs + the median of 5, on a bounded sample of config B
    (batch 8: resampler fwd+bwd, and 6 of the 36 gated blocks fwd+bwd, extrapolated to 36); plus config A end to end."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from detgen import det, resampler_params, xattn_params
    from oracle import torch_port as TP
    cores = _usable_cores()
    old_threads = torch.get_num_threads()
    torch.set_num_threads(cores)

    def median_ms(fn, warm=2, reps=5, budget_s=25.0):
        deadline = time.perf_counter() + budget_s   # every leg is bounded: a slow host shortens the sample instead of stalling the bench
        ts = []
        for i in range(warm + reps):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            if i >= warm:
                ts.append(dt)
            if time.perf_counter() + dt > deadline and (ts or i >= 1):      # no time for another run: keep what was measured
                ts = ts or [dt]
                break
        return sorted(ts)[len(ts) // 2]

    b, L, d, dv, n_timed, n_blocks = 8, 32, 1280, 1024, 6, 36
    rp = {k: torch.from_numpy(v).requires_grad_(True) for k, v in resampler_params(dv, 6, 8, 64, 64, 4, 4, tag="cpu").items()}
    xps = [{k: torch.from_numpy(v).requires_grad_(True) for k, v in xattn_params(d, dv, 8, 64, 4, tag=f"cpu{i}").items()} for i in range(n_timed)]
    x = torch.from_numpy(det((b, 1, 257, dv), "cpu-x"))
    y0 = torch.from_numpy(det((b, L, d), "cpu-y"))
    ml = torch.zeros((b, L), dtype=torch.long); ml[:, 0] = 1
    vf_const = TP.resampler(x, rp).detach().reshape(b, 1, 64, dv)

    def rs_step():
        TP.resampler(x, rp).sum().backward()

    def blocks_step():
        yy = y0.clone().requires_grad_(True)
        h = yy
        for p in xps:
            h, _ = TP.gated_xattn_block(h, vf_const, ml, p)
        h.sum().backward()

    t_rs = median_ms(rs_step)
    t_blk = median_ms(blocks_step) / n_timed
    hot = t_rs + n_blocks * t_blk
    out = {"value": round(b / hot, 3), "unit": "images/sec (hot path only: resampler + 36 xattn blocks, fwd+bwd, fp32)", "cores": int(cores),
           "kind": "port", "sample": f"torch CPU restatement (oracle/torch_port.py), {cores} threads (affinity / cgroup quota), warm-up 2 + median of up to 5 (time-bounded), batch {b} of config B: "
                                     f"resampler fwd+bwd {t_rs:.3f}s + {n_timed} of 36 gated blocks fwd+bwd ({t_blk:.4f}s each, extrapolated x36); "
                                     "excludes the frozen CLIP / GPT-2 backbones"}
    try:     # config A (BASELINE configs[0]) end to end on the host: the drop-in model with its fused entry points pointed at the torch port
        from flamingo_mini_amd import FlamingoConfig, FlamingoModel
        TP.install()
        cfg = FlamingoConfig(lm="gpt2", clip_model_type="openai/clip-vit-base-patch32", dim=768, dim_visual=768, random_init_backbones=True)
        torch.manual_seed(1)
        m = FlamingoModel(cfg).train()
        m.flamingo.hoist_kv = False
        with torch.no_grad():
            for hook in m.flamingo.get_modified_layers():
                hook.xattn_block.alpha_attn.fill_(0.5); hook.xattn_block.alpha_ffw.fill_(0.5)
        g = torch.Generator().manual_seed(5)
        px = torch.randn((2, 1, 3, 224, 224), generator=g)
        ids = torch.randint(0, 50257, (2, 32), generator=g)
        mla = torch.zeros((2, 32), dtype=torch.long); mla[:, 0] = 1

        def a_step():
            m.zero_grad(set_to_none=True)
            m(input_ids=ids, attention_mask=torch.ones_like(ids), media_locations=mla, pixel_values=px, labels=ids).loss.backward()

        t_a = median_ms(a_step)
        out["config_A_end_to_end"] = {"value": round(2 / t_a, 3), "unit": "images/sec (fwd+bwd, gpt2 124M + CLIP ViT-B/32, batch 2, seq 32, fp32, CPU)",
                                      "s_per_step": round(t_a, 4)}
    except Exception as e:      # never let the reported baseline take the benchmark down
        out["config_A_end_to_end"] = {"error": repr(e)[:200]}
    finally:
        TP.uninstall()
        torch.set_num_threads(old_threads)
    return out


def _dump_allocator_map(path):
    """Every allocator segment and block (address, size, state, the innermost repository / torch frames that allocated it), gzip JSON:
    a GPU memory-access fault prints an address, this says whose buffer it is next to."""
    import gzip
    snap = torch.cuda.memory_snapshot()
    segs = []
    for sg in snap:
        blocks = []
        for b in sg.get("blocks", []):
            fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in (b.get("frames") or [])[:40]
                  if "flamingo" in f["filename"] or "bench.py" in f["filename"] or "transformers" in f["filename"]][:5]
            blocks.append([b.get("address", 0), b["size"], b.get("requested_size", 0), b["state"], fr])
        segs.append(dict(address=sg["address"], size=sg["total_size"], pool=str(sg.get("segment_pool_id")), stream=sg.get("stream", 0), blocks=blocks))
    with gzip.open(path, "wt") as f:
        json.dump(segs, f)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (see the docstring)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the fusion path has no CPU fallback")
    # --shared-gpu-rehearsal: the N-rank launch contract on a box with ONE GPU - every rank uses device 0 and the exchange goes through gloo (RCCL
    # refuses two ranks on one device).  It exercises what the driver's `torch.distributed.run ... bench.py --gpus N` command exercises - environment,
    # rendezvous, reducer, agreement on the graph mode, max-over-ranks timing, one JSON line from rank 0 - and its throughput means nothing.
    dev_index = 0 if args.shared_gpu_rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1 or args.force_collectives:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.rccl_channels > 0:
            os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_channels)
        if args.shared_gpu_rehearsal:
            dist.init_process_group("gloo")
        elif world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        else:
            dist.init_process_group("nccl", device_id=device)
    collectives = world > 1 or args.force_collectives
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    memsnap = os.environ.get("FF_BENCH_MEMSNAP", "")            # debugging aid: allocator map (with allocation stacks) before the timed region
    if memsnap:
        torch.cuda.memory._record_memory_history(context="alloc", stacks="python", max_entries=400000)

    from flamingo_mini_amd import ffi
    from flamingo_mini_amd.data_parallel import GradientAllReducer
    lib = ffi.lib()                                            # aborts loudly if the HIP library is missing

    stock_tuned = False
    if args.stock_tuning == "on" and args.dtype == "bf16":
        from flamingo_mini_amd.backbones import load_stock_gemm_tuning
        stock_tuned = load_stock_gemm_tuning()
    model, cfg = build_model(args, device, dtype)
    batch = synthetic_batch(args, cfg, device, dtype, rank)
    model.set_launch_structure(hoist_kv=args.hoist_kv == "on")
    params = [p for p in model.parameters_trainable()]
    n_trainable = sum(p.numel() for p in params)
    # Config E (4 x 1024 tokens) is launched eagerly under `auto`: replaying a captured step faults inside PyTorch-ROCm's memory-efficient
    # SDPA kernels once the sequence reaches 1024 tokens and the stock LM has dropout active (bisected in round 2, DESIGN.md section 5: the same
    # fault with gpt2-large, with one gated block, without the optimizer; no fault with the math SDPA backend, with LM dropout 0, at 512
    # tokens, or when only this library's kernels are captured at the same shapes).  `--graph on` still forces the capture.
    use_graph = args.graph in ("on", "piecewise") or (args.graph == "auto" and args.config != "E")
    graph_mode = "piecewise" if args.graph == "piecewise" else ("full" if use_graph else "off")
    master = dict(master_dtype=torch.float32) if args.optimizer.endswith("-master") and dtype == torch.bfloat16 else {}

    def make_optimizer(capturable):
        if args.no_optimizer:
            return None
        if args.optimizer.startswith("fused"):
            from flamingo_mini_amd import FusedAdamW
            return FusedAdamW(params, lr=1e-4, capturable=capturable, **master)
        if args.optimizer.startswith("sharded"):
            from flamingo_mini_amd.data_parallel import ShardedAdamW
            return ShardedAdamW(model, lr=1e-4, capturable=capturable, **master)
        return torch.optim.AdamW(params, lr=1e-4, fused=True, capturable=capturable)

    opt = make_optimizer(use_graph)
    sharded = args.optimizer.startswith("sharded") and opt is not None
    # ShardedAdamW does the exchange itself (reduce-scatter / all-gather per bucket); otherwise the buckets are all-reduced
    reducer = None if sharded else GradientAllReducer(model, reduce_dtype=torch.float32 if args.reduce_dtype == "f32" else None,
                                                      force_collectives=args.force_collectives)
    # explicit launch structure (after the reducer chose its own): what tools/bucket_timeline.py traces on one GPU
    if args.sync_exchange == "off":
        from flamingo_mini_amd import functional as _F
        _F.use_sync_exchange = False
    if args.resampler_layerwise != "auto":
        model.flamingo.resampler.layerwise = args.resampler_layerwise == "on"
    if args.wgrad_group > 0:
        model.set_launch_structure(wgrad_group=args.wgrad_group)
    if args.kv_group >= 0:
        model.set_launch_structure(kv_project_group=args.kv_group)

    def eager_step():
        for p in params:                 # == model.zero_grad(set_to_none=True) without walking the ~1000 frozen parameters (4 ms of host time)
            p.grad = None
        out = model(**batch)
        out.loss.backward()
        if reducer is not None:
            reducer.finish()
        if opt is not None:
            opt.step()
        return out.loss

    if args.debug_phases:
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
        acc = [0.0] * 5
        for _ in range(5):
            t = [time.perf_counter()]
            for p in params:
                p.grad = None
            t.append(time.perf_counter())
            out = model(**batch)
            t.append(time.perf_counter())
            out.loss.backward()
            t.append(time.perf_counter())
            if reducer is not None:
                reducer.finish()
            if opt is not None:
                opt.step()
            t.append(time.perf_counter())
            torch.cuda.synchronize()
            t.append(time.perf_counter())
            acc = [a + (t[i + 1] - t[i]) * 1e3 / 5 for i, a in enumerate(acc)]
        print("host ms/step: zero_grad %.2f forward %.2f backward %.2f optimizer %.2f drain %.2f" % tuple(acc), flush=True)
        return
    graph_note = ""
    step = eager_step
    overlap_opt = False
    if use_graph:
        # full: forward + backward + gradient all-reduces + optimizer captured once, one graph launch per step; piecewise: one sub-graph per
        # backward segment, collectives issued eagerly between the replays (flamingo_mini_amd/graphs.py).  A rank on which `full` cannot be
        # captured (e.g. RCCL kernels refusing the capture) makes EVERY rank fall back to `piecewise`, then to eager launches: all ranks
        # must take the same path - a rank replaying captured collectives while another launches them eagerly would deadlock.
        from flamingo_mini_amd import GraphedTrainStep
        from flamingo_mini_amd.graphs import PiecewiseGraphedTrainStep
        live_reducer = reducer if (collectives and reducer is not None) else None
        # With collectives `auto` tries the piecewise replay first: on one GPU with the exchange going through a 1-rank RCCL group it is the
        # faster of the two (44.3 vs 47.6 ms per step, profiles/r04_launch_modes_one_rank_rccl.txt) and it does not depend on RCCL's kernels
        # accepting a stream capture; `--graph on` asks for the whole-step capture first.
        if graph_mode == "piecewise":
            attempts = ["piecewise"]
        elif collectives and not sharded:
            attempts = ["full", "piecewise"] if args.graph == "on" else ["piecewise", "full"]
            if args.shared_gpu_rehearsal:
                attempts = ["piecewise"]                    # (a gloo exchange happens on the host: it cannot be part of a captured graph)
        else:
            attempts = ["full"]
        overlap_opt = (args.overlap_optimizer == "on" or (args.overlap_optimizer == "auto" and collectives)) and args.pace == "host" and opt is not None and args.optimizer.startswith("fused") and not sharded
        graph_mode = "off"
        for mode in attempts:
            err = None
            try:
                if mode == "full":
                    graphed = GraphedTrainStep(model, opt, batch, warmup=max(args.warmup, 1), reducer=live_reducer)
                else:
                    graphed = PiecewiseGraphedTrainStep(model, opt, batch, warmup=max(args.warmup, 1), reducer=live_reducer, pace=args.pace, segment_layers=args.segment_layers,
                                                        overlap_optimizer=overlap_opt, segment_arena=args.segment_arena == "on")
            except Exception as e:
                if world == 1 and not collectives and args.graph in ("on", "piecewise"):
                    raise
                err = e
                torch.cuda.synchronize()
                graphed = None              # (a piecewise step that raised has already taken its cut points out and ended the reducer's recording)
            captured = err is None
            if world > 1:
                flag = torch.tensor([1 if captured else 0], device=device, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                captured = bool(flag.item())
            if captured:
                step, graph_mode = graphed, mode
                break
            if graphed is not None and hasattr(graphed, "close"):
                graphed.close()             # this rank captured but another did not: every rank leaves the mode, and leaves the model as it found it
            model.install_autograd_cuts(None)
            graph_note += f"; {mode} graph capture failed on a rank" + (f" ({type(err).__name__}: {str(err)[:120]})" if err is not None else "")
        if graph_mode == "off":
            graph_note += ", eager launches instead"
            use_graph = False
            if opt is not None and args.optimizer != "torch":
                if sharded:
                    opt.close()
                opt = make_optimizer(False)

    if memsnap:
        _dump_allocator_map(memsnap)
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss_first = None
    for i in range(args.steps):
        loss = step()
        if i == 0:
            loss_first = loss.detach().clone()      # (the replayed graph overwrites its static loss tensor every step)
    loss_last = loss.detach().clone()
    host_issue_ms = (time.perf_counter() - t0) * 1e3 / max(args.steps, 1)      # how long the host needed to ISSUE a step (a replayed step that is
    torch.cuda.synchronize()                                                    # as long as its GPU time is launch-bound)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # the error word of the fused cross-attention kernels' in-launch hand-offs (a timed-out arrival wait = a launch computed on garbage): read
    # after the timed region and again after the instrumented steps; a non-zero word is printed in `config` and fails the run
    from flamingo_mini_amd import functional as _Fs
    sync_timeouts = int(_Fs.sync_exchange_status()) if getattr(_Fs, "_sync_buffers", None) else 0
    piecewise_host = None
    if graph_mode == "piecewise" and hasattr(step, "host_timing"):      # host seconds by kind of call over 5 more steps (launch-bound or not, and by what)
        step.host_timing = {}
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        n = max(step.host_timing.pop("steps", 1), 1)
        piecewise_host = {k: round(v * 1e3 / n, 3) for k, v in step.host_timing.items()}
        piecewise_host["sub_graphs"] = len(step.graphs) + (1 if step._opt_graph is not None else 0) + sum(g is not None for g in step._opt_pieces)
        piecewise_host["collectives_issued"] = sum(len(b) for b in step.segment_buckets)
        step.host_timing = None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Roofline leg: HIP events cannot be recorded inside a replayed graph and cost ~2 ms/step when they bracket every launch, so
    # the per-launch durations come from `--profile-steps` eager steps of the same workload right after the timed region.
    prof_steps = max(args.profile_steps, 0)
    max_rec = 4096 * max(prof_steps, 1)
    eager_ms = None
    if graph_mode == "piecewise" and hasattr(step, "close"):
        step.close()                        # the eager steps below run one backward pass over the whole graph
    bucket_timeline = None
    if collectives and reducer is not None and (args.bucket_timeline or world > 1):
        # One eager step with HIP events around every bucket's exchange (rank 0 reports): when each bucket became final, when its collective
        # finished, both relative to the start of backward - the table DESIGN.md section 6 predicts, line by line.
        eager_step()
        torch.cuda.synchronize()
        for p in params:
            p.grad = None
        out = model(**batch)
        t_b0, t_b1, t_fin = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        reducer.record_timeline(True)
        t_b0.record()
        out.loss.backward()
        t_b1.record()
        reducer.finish()
        t_fin.record()
        if opt is not None:
            opt.step()
        torch.cuda.synchronize()
        rows = reducer.timeline_ms(t_b0)
        reducer.record_timeline(False)
        bwd_ms = t_b0.elapsed_time(t_b1)
        bucket_timeline = {"backward_ms": round(bwd_ms, 3), "exchange_finished_ms": round(t_b0.elapsed_time(t_fin), 3),
                           "exposed_communication_ms": round(max(0.0, t_b0.elapsed_time(t_fin) - bwd_ms), 3),
                           "launch_mode": "eager (events cannot be recorded inside a replayed graph)", "buckets": rows}
    if prof_steps:
        if use_graph:
            loss = eager_step()         # the eager allocator pool is cold after the graph's private pool: one untimed step
        if rank == 0:
            lib.ff_gemm_profile_enable(max_rec)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            loss = eager_step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t1) / prof_steps * 1e3
    sync_timeouts |= int(_Fs.sync_exchange_status()) if getattr(_Fs, "_sync_buffers", None) else 0
    if world > 1:
        flag = torch.tensor([sync_timeouts], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        sync_timeouts = int(flag.item())
    loss_val = float(loss.float().item())
    loss_first_val, loss_last_val = float(loss_first.float().item()), float(loss_last.float().item())

    if rank == 0:
        groups, shapes, attn = gemm_profile_summary(lib, ffi, max_rec) if prof_steps else ({}, {}, {})
        families = attn.pop("_families", {})
        if args.gemm_table:
            with open(args.gemm_table, "w") as f:
                f.write("M N K nz aL bL tile splitK launches/step us/launch TF/s ms/step\n")
                for k, g in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"]):
                    f.write(" ".join(map(str, k)) + f" {g['launches'] / prof_steps:.1f} {g['ms'] / g['launches'] * 1e3:.1f} "
                            f"{g['flops'] / (g['ms'] * 1e-3) / 1e12:.1f} {g['ms'] / prof_steps:.3f}\n")
        ms_per_step = elapsed / args.steps * 1e3
        images = args.batch * args.images * world * args.steps        # a video clip counts as one image (its frames share one set of 64 latents)
        roofline = None
        if groups:
            # The dominant kernel is chosen by tile FAMILY (all launches of one tile, whatever the operand layouts), not by (tile, layouts):
            # grouping by layout split the 128 x 160 feed-forward family four ways and made the weight-gradient instantiation - the
            # best-performing GEMM of the step - "dominant" (VERDICT r05).  Within the dominant family the layout group with the most time is
            # the kernel the line's achieved / peak / frac / traffic describe; `family` and `hot_path` put the honest totals next to it.
            fam_key, fam = max(families.items(), key=lambda kv: kv[1]["ms"])
            key, g = max(((k, v) for k, v in groups.items() if (k[0], k[1]) == fam_key), key=lambda kv: kv[1]["ms"])
            is_bf16 = key[0] == ffi.DTYPE_BF16
            peak = MFMA_BF16_DENSE_PEAK_TFLOPS if is_bf16 else MFMA_F32_PEAK_TFLOPS
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
            tm, tn = (64, 128) if key[1] == 6412 else (key[1], key[1])
            name = (f"ff::gemm_bf16_dma_kernel<{tm}, {tn}, {key[2]}, {key[3]}, 2>" if is_bf16 else f"ff::gemm_f32_kernel<{key[2]}, {key[3]}>")
            if key[1] == 128160:      # the producer / consumer kernels: <BM, BN, AL, BL, stages, workgroups per CU, DMA waves, MFMA waves>
                name = (f"ff::gemm_bf16_pc_kernel<128, 160, {key[2]}, {key[3]}, 3, 1, 4, 4>" if key[4]
                        else f"ff::gemm_bf16_pc_kernel<128, 160, {key[2]}, {key[3]}, 4, 1, 8, 4>")
            elif key[1] == 64002:
                name = f"ff::gemm_bf16_pc_kernel<64, 64, {key[2]}, {key[3]}, 3, 2, 4, 4>"
            elif key[1] == 3264:
                name = "ff::gemm_bf16_pc_kernel<32, 64, 0, 0, 4, 2, 4, 4>"
            elif key[1] == 3216:
                name = "ff::gemm_bf16_rows32_kernel"
            elif key[1] == 128002:
                name = f"ff::gemm_bf16_pc_kernel<128, 128, {key[2]}, {key[3]}, 2, 2, 4, 4>"
            tot_ms = sum(v["ms"] for v in groups.values())
            tot_fl = sum(v["flops"] for v in groups.values())
            # HBM bytes per launch of this kernel: PMC counters cannot be read from inside this process (rocprofv3 wraps the command), so the
            # figure comes from the latest committed counter passes of the SAME command (tools/gpu_final.sh -> profiles/rNN_pmc_traffic.json)
            # and is tagged with its source; a kernel that file does not list gives null plus the reason, never a number of another kernel.
            traffic, traffic_source = None, "no profiles/r*_pmc_traffic.json in the tree"
            try:
                import glob
                path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
                with open(path) as f:
                    kernels = json.load(f)["kernels"]
                rel_path = os.path.relpath(path, ROOT)
                if name in kernels:
                    traffic = kernels[name].get("hbm_bytes_per_launch")
                    traffic_source = (f"{rel_path}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, (2*FETCH_SIZE + WRITE_SIZE) KiB per launch "
                                      "(MI355X_MICROARCH.md HBM section), average over the kernel's launches; not collected by this run")
                else:
                    traffic_source = f"{rel_path} has no entry for {name} (kernel renamed since those passes): re-run tools/gpu_final.sh"
            except Exception as e:
                traffic_source = f"unreadable traffic file: {e!r}"[:160]
            roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                        "traffic": traffic, "traffic_source": traffic_source, "kernel": name, "launches": g["launches"],
                        "avg_launch_us": round(g["ms"] / g["launches"] * 1e3, 2),
                        "avg_launch_gflop": round(g["flops"] / g["launches"] / 1e9, 3),
                        "family": {"tile": _tile_name(fam_key[1]), "tflops": round(fam["flops"] / (fam["ms"] * 1e-3) / 1e12, 2),
                                   "frac": round(fam["flops"] / (fam["ms"] * 1e-3) / 1e12 / peak, 4), "launches_per_step": round(fam["launches"] / prof_steps, 1),
                                   "ms_per_step": round(fam["ms"] / prof_steps, 3)},
                        "by_family": [{"tile": _tile_name(k[1]), "dtype": "bf16" if k[0] == ffi.DTYPE_BF16 else "f32",
                                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                       "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / (MFMA_BF16_DENSE_PEAK_TFLOPS if k[0] == ffi.DTYPE_BF16 else MFMA_F32_PEAK_TFLOPS), 4),
                                       "launches_per_step": round(v["launches"] / prof_steps, 1), "ms_per_step": round(v["ms"] / prof_steps, 3)}
                                      for k, v in sorted(families.items(), key=lambda kv: -kv[1]["ms"])],
                        "hot_path": hot_path_summary(args, cfg, model, tot_fl / prof_steps, tot_ms / prof_steps, attn, prof_steps),
                        "all_fusion_gemms": {"tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                             "ms_per_step": round(tot_ms / prof_steps, 3),
                                             "share_of_step": round(tot_ms / prof_steps / ms_per_step, 3)},
                        "measured": f"HIP events around every launch in {prof_steps} eager steps of the same workload run right after the "
                                    f"timed region ({eager_ms:.2f} ms/step with the instrumentation)"}
        result = {
            "metric": "images/sec (fwd+bwd) flamingo-mini bs=32" if args.config == "B" else f"images/sec (fwd+bwd) config {args.config} bs={args.batch}",
            "value": round(images / elapsed, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "tolerance": TOLERANCE, "data": "synthetic",
            "config": {"name": args.config,
                       "workload": f"{args.what}; {args.lm} + {args.clip}, {args.images} image(s)" + (f" x {args.frames} frames" if args.frames else "")
                                   + f" + {args.seq_len} tokens per sequence, xattn_every {args.xattn_every}, "
                                   f"per-GPU batch {args.batch}; step = fwd + bwd + grad all-reduce"
                                   + ("" if args.no_optimizer else f" + AdamW ({args.optimizer})")
                                   + ({"full": "; step replayed from a captured HIP graph" + (" (RCCL all-reduces captured)" if collectives else ""),
                                       "piecewise": "; step replayed from one HIP graph per backward segment, RCCL collectives issued eagerly between the replays",
                                       "off": "; eager launches"}[graph_mode])
                                   + graph_note + "; random-init weights, gates alpha=0.5",
                       "global_batch": args.batch * world, "seq_len": args.seq_len, "parallelism": f"dp{world}",
                       "trainable_params": n_trainable,
                       # the same batch every step, AdamW at lr 1e-4 from random init: the loss of the timed steps is a trajectory, not a constant
                       # ("the loss printed by the bench"); `loss` = after the instrumented eager steps that follow
                       "loss_first": round(loss_first_val, 4), "loss_last": round(loss_last_val, 4), "loss": round(loss_val, 4),
                       "optimizer_steps_before_timed_region": args.warmup + (max(args.warmup, 1) if use_graph else 0),
                       "optimizer": "none" if args.no_optimizer else args.optimizer, "hip_graph": use_graph, "graph_mode": graph_mode, "collectives": bool(collectives), "rccl_channels": (args.rccl_channels or None), "segment_layers": (args.segment_layers if graph_mode == "piecewise" else None), **({"rehearsal": "all ranks share ONE GPU, gloo exchange: the value is not a measurement"} if args.shared_gpu_rehearsal else {}), "collective_pace": (args.pace if graph_mode == "piecewise" and collectives else None), "overlapped_optimizer": bool(graph_mode == "piecewise" and use_graph and overlap_opt), "host_issue_ms_per_step": round(host_issue_ms, 3), "piecewise_host_ms_per_step": piecewise_host,
                       "hoisted_kv": bool(model.flamingo.hoist_kv), "sync_exchange": args.sync_exchange == "on", "sync_exchange_timeouts": sync_timeouts, "resampler_layerwise": bool(model.flamingo.resampler.layerwise), "stock_gemm_tuning_file": stock_tuned,
                       "backbone_tweaks": args.backbone_tweaks == "on"},
            "roofline": roofline,
        }
        if attn:    # north star: throughput of the softmax(QK^T)V core as a fraction of the HBM roofline (8 TB/s spec peak)
            result["attention_roofline"] = {
                k: {"bound": "hbm", "achieved": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(v["bytes"] / (v["ms"] * 1e-3) / 8e12, 4), "launches": v["launches"],
                    "avg_launch_us": round(v["ms"] / v["launches"] * 1e3, 2), "avg_launch_mb": round(v["bytes"] / v["launches"] / 1e6, 2),
                    **({"with_out_projection": True} if v.get("with_out_projection") else {}), **({"with_layernorm": True} if v.get("with_layernorm") else {})}
                for k, v in attn.items()}
        if bucket_timeline is not None:
            result["bucket_timeline"] = bucket_timeline
        if world == 1 and args.caption_tokens > 0:
            result["caption"] = caption_leg(args, model, batch, device)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args)
        want_companions = args.companions == "on" or (args.companions == "auto" and args.config == "B" and args.backbone_tweaks == "off"
                                                      and args.optimizer == "fused" and not args.no_optimizer and not collectives)
        if world == 1 and want_companions:
            # free this process's model and graph pools first: the children build their own copies on the same GPU
            batch = model = opt = step = graphed = reducer = params = loss = loss_first = loss_last = None      # (closures keep the names alive)
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            result["with_backbone_op_substitutions"] = dict(
                run_companion(args, ["--backbone-tweaks", "on", "--optimizer", args.optimizer]),
                what="same workload and fusion path, plus three result-identical op substitutions INSIDE the frozen backbones (ViT patch convolution as a "
                     "matmul, CLIP's QuickGELU in one pass, HF's 8-kernel NewGELU -> torch's fused tanh GELU; tests/test_hip_backbones.py) and the pre-tuned "
                     "hipBLASLt solution file for their GEMMs - NOT the north star's configuration (backbones on stock PyTorch-ROCm), which is the headline")
            result["fp32_master_optimizer"] = dict(run_companion(args, ["--optimizer", "fused-master"]),
                                                   what="same as the headline run but AdamW keeps fp32 master weights and fp32 moments for the bf16 parameters "
                                                        "(ff_adamw_step_mixed) - the reference's --fp16 recipe, training/train.sh:24")
        try:        # RCCL prints its version banner through C stdio, which is block-buffered when stdout is a file: flush it so that the JSON
            import ctypes       # line below really is the LAST line of the output
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    if sync_timeouts:
        print("bench.py: an in-launch hand-off of the fused cross-attention kernels timed out (config.sync_exchange_timeouts = 1): the run's "
              "numbers are invalid", file=sys.stderr, flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()
