"""Two data-parallel ranks on the REAL kernels: two processes share the one GPU of the test box (RCCL refuses two ranks on a device, so
the exchange goes through gloo, which accepts device tensors), each trains on its half of the h64 fixture's batch, and the result must
equal one process training on the whole batch - DDP's mean semantics (/root/reference/training/train.sh:26,36) on the
fused bf16 / fp32 kernels with hoisted K / V, deferred grouped weight gradients and the reducer's buckets, for eager steps and for the
piecewise replay (captured sub-graphs, collectives issued between them, AdamW as per-segment sub-graphs)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_STEPS = 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _paths():
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _train(model, batch, mode, reducer, adamw):
    from flamingo_mini_amd import FusedAdamW
    from flamingo_mini_amd.graphs import PiecewiseGraphedTrainStep
    opt = FusedAdamW([p for p in model.parameters_trainable()], capturable=mode not in ("eager", "eager-autocast"), **adamw)
    losses = []
    if mode in ("eager", "eager-autocast"):
        for _ in range(N_STEPS):
            model.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "eager-autocast"):
                loss = model(**batch).loss
            loss.backward()
            if reducer is not None:
                reducer.finish()
            opt.step()
            losses.append(float(loss.detach()))
    else:
        step = PiecewiseGraphedTrainStep(model, opt, batch, warmup=1, reducer=reducer, segment_layers=1, pace="host" if mode == "overlapped" else "stream",
                                         overlap_optimizer=mode == "overlapped")       # (the constructor's warm-up is training step 1)
        losses = [float("nan")] + [float(step()) for _ in range(N_STEPS - 1)]
    torch.cuda.synchronize()
    return losses


def _worker(rank, world, port, out_dir, mode, dtype_name):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _paths()
    from test_model_plumbing import H64, build_h64
    from flamingo_mini_amd.data_parallel import GradientAllReducer
    model, z, batch = build_h64(getattr(torch, dtype_name), "cuda")
    per = batch["input_ids"].shape[0] // world
    mine = {k: v[rank * per:(rank + 1) * per].contiguous() for k, v in batch.items()}
    reducer = GradientAllReducer(model)
    assert reducer.active and not reducer.cuda
    losses = _train(model, mine, mode, reducer, H64["adamw"])
    reducer.close()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), losses=np.array(losses),
             **{k: p.detach().float().cpu().numpy() for k, p in model.named_parameters() if p.requires_grad})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,dtype_name", [(m, d) for d in ("float32", "bfloat16") for m in ("eager", "piecewise", "overlapped")] + [("eager-autocast", "float32")],
                         ids=lambda v: str(v))
def test_two_ranks_on_the_real_kernels_equal_one_process_on_the_whole_batch(tmp_path, mode, dtype_name):
    """(eager-autocast, round 6: fp32 parameters under torch.autocast(bf16) - the fused modules run on casts of the parameters, no gradient bucket
    of theirs arrives, and the reducer exchanges those parameters' gradients in finish().)"""
    _paths()
    from test_model_plumbing import H64, build_h64
    from util import rel
    mp.start_processes(_worker, args=(2, _free_port(), str(tmp_path), mode, dtype_name), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    model, z, batch = build_h64(getattr(torch, dtype_name), "cuda")
    ref_losses = _train(model, batch, "eager-autocast" if mode == "eager-autocast" else "eager", None, H64["adamw"])
    f32 = dtype_name == "float32" and mode != "eager-autocast"
    for i in range(1 if mode not in ("eager", "eager-autocast") else 0, N_STEPS):      # the whole-batch loss is the mean of the two ranks' losses (equal token counts)
        both = 0.5 * (float(r0["losses"][i]) + float(r1["losses"][i]))
        assert abs(both - ref_losses[i]) <= (2e-5 if f32 else 3e-2) * max(1.0, abs(ref_losses[i])), (i, both, ref_losses)
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        assert np.array_equal(r0[k], r1[k]), k                  # the ranks hold the same parameters after every exchange
        assert rel(torch.from_numpy(r0[k]), p.detach().float().cpu()) < (1e-4 if f32 else 3e-2), k
