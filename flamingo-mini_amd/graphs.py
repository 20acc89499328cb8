"""HIP-graph replay of a whole training step.

At flamingo-mini's benchmark size one step is ~3100 kernel launches of 5-40 us each; launched one by one from Python the
GPU idles ~10 % of the time between them.  Every launch of this library only enqueues work on the caller's stream (no
allocation, no host synchronisation), so forward + backward + optimizer can be captured once into a HIP graph
(torch.cuda.CUDAGraph on ROCm) and replayed with a single launch per step.

    step = GraphedTrainStep(model, optimizer, example_batch)         # warms up, captures
    loss = step(batch)                                                # copies the batch into the static inputs, replays

Requirements: fixed shapes, an optimizer whose step is capture-safe (FusedAdamW(capturable=True) or
torch.optim.AdamW(capturable=True)), and no data-dependent host control flow in the model (true for FlamingoModel's
training forward).  With `reducer=` (data_parallel.GradientAllReducer) the gradient all-reduces are part of the capture: they are
issued on the reducer's side stream, which forks from and rejoins the capturing stream through events, so the replayed graph
contains the RCCL kernels and their overlap with backward (PyTorch captures NCCL / RCCL collectives; the communicator must already
exist, which the eager warm-up steps guarantee).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch


def _refuse_live_autograd_graphs(model: torch.nn.Module) -> None:
    """A parameter's AccumulateGrad node lives as long as some autograd graph refers to it and remembers the stream it was created on.  If
    outputs of an earlier eager forward (built on the default stream) are still alive, the captured backward would hand its gradients to
    those nodes: the engine then synchronises the capturing stream with the default stream in the middle of the capture - on ROCm 7 a
    segmentation fault in the runtime at the end of the capture (tools/capture_after_eager.py), not an error message.  Refuse up front.
    The probe: a mark left in the node's metadata survives only if somebody else keeps the node alive."""
    def accumulator(p):
        return p.view_as(p).grad_fn.next_functions[0][0]

    stale = []
    try:
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            accumulator(p).metadata["ff_capture_probe"] = True
            node = accumulator(p)
            if node.metadata.pop("ff_capture_probe", False):
                stale.append(name)
    except (AttributeError, IndexError, TypeError):      # a torch build whose autograd nodes cannot be probed this way: capture unguarded
        return
    if stale:
        raise RuntimeError(
            f"GraphedTrainStep: an autograd graph from an earlier forward still refers to {len(stale)} parameter(s) of the model (e.g. "
            f"{stale[0]}). Delete the outputs / losses of earlier forward passes (or run them under torch.no_grad()) before capturing: "
            "their gradient-accumulation nodes are bound to the stream of that forward and cannot take part in a stream capture.")


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer], example_batch: Dict[str, torch.Tensor],
                 warmup: int = 3, loss_fn: Optional[Callable] = None, reducer=None):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedTrainStep needs a GPU")
        self.model, self.optimizer, self.reducer = model, optimizer, reducer
        _refuse_live_autograd_graphs(model)
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        self._loss_fn = loss_fn or (lambda out: out.loss)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up off the default stream: lazy init, autotuning, allocator pools
            for _ in range(max(warmup, 1)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        model.zero_grad(set_to_none=True)
        # With collectives in the step the process group's watchdog THREAD is alive and polls the events of the warm-up steps' collectives
        # (hipEventQuery) whenever it wakes up; under the default "global" capture mode such a call from another thread during the capture
        # aborts the process (seen once in ~10 runs of the 1-rank RCCL test).  "thread_local" restricts the check to this thread's own calls.
        import torch.distributed as dist
        mode = "thread_local" if reducer is not None or (dist.is_available() and dist.is_initialized()) else "global"
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.loss = self._eager().detach()
        torch.cuda.synchronize()

    def _eager(self) -> torch.Tensor:
        self.model.zero_grad(set_to_none=True)              # gradients are re-created (not accumulated) by every backward
        loss = self._loss_fn(self.model(**self.static))
        loss.backward()
        if self.reducer is not None:
            self.reducer.finish()                           # the capturing stream waits for the collectives before the optimizer reads .grad
        if self.optimizer is not None:
            self.optimizer.step()
        return loss

    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        """Replay one step.  `batch` (same keys / shapes / dtypes as the example) is copied into the static inputs; None reuses them.
        Returns the static loss tensor (overwritten by the next replay)."""
        if batch is not None:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self.static[k].copy_(v, non_blocking=True)
        if self.optimizer is not None and hasattr(self.optimizer, "sync_device_hyperparams"):
            self.optimizer.sync_device_hyperparams()        # an LR scheduler may have changed group["lr"] since the capture
        self.graph.replay()
        return self.loss
