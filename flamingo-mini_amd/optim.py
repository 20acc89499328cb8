"""FusedAdamW: torch.optim.Optimizer whose step() is one multi-tensor HIP kernel sweep (ff_adamw_step) over all parameters
of a dtype — same update rule, defaults and state_dict layout (`step`, `exp_avg`, `exp_avg_sq`) as torch.optim.AdamW, so
optimizer checkpoints interchange.  The reference trains with `--optim adamw_torch` (training/train.sh:10-13) on
`model.parameters_trainable()`.

`capturable=True` keeps the step count in a device scalar per parameter group (bias corrections are computed in the kernel),
so `step()` can be captured into a HIP graph and replayed (graphs.GraphedTrainStep); `state_dict()` reads the count back."""
from __future__ import annotations

import ctypes as C
from typing import Iterable

import torch

from . import ffi


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 grad_scale: float = 1.0, capturable: bool = False):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale, capturable=capturable))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = ffi.lib()
        for group in self.param_groups:
            buckets = {}
            capturable = group.get("capturable", False)
            if capturable:
                self._advance_device_steps(group)
            for p in group["params"]:
                if p.grad is None:
                    continue
                ffi.require_cuda(p, p.grad)
                if p.grad.dtype != p.dtype or not p.is_contiguous():
                    raise ffi.FusionLibraryError("FusedAdamW needs contiguous parameters with gradients of the same dtype")
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not capturable:
                    st["step"] += 1
                buckets.setdefault((p.dtype, p.device, 0 if capturable else int(st["step"])), []).append((p, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"]))
            for (dtype, device, step), items in buckets.items():
                n = len(items)
                desc = ffi.AdamWDesc(ffi.dtype_code(dtype), n, step, group["lr"], group["betas"][0], group["betas"][1], group["eps"],
                                     group["weight_decay"], group["grad_scale"],
                                     group["_step_dev"][device].data_ptr() if capturable else None)
                cols = [ffi.ptr_array([it[k] for it in items]) for k in range(4)]
                numels = (C.c_longlong * n)(*[it[0].numel() for it in items])
                ffi.check(lib.ff_adamw_step(desc, cols[0], cols[1], cols[2], cols[3], numels, ffi.stream_handle(device)), "ff_adamw_step")
        return loss

    # ------------------------------------------------------------------ capturable mode
    def _advance_device_steps(self, group):
        """One float32 step counter per (group, device), advanced by a device-side add (captured along with the update)."""
        counters = group.setdefault("_step_dev", {})
        for p in group["params"]:
            if p.grad is not None and p.device not in counters:
                host_steps = [int(self.state[q]["step"]) for q in group["params"] if q in self.state and "step" in self.state[q]]
                counters[p.device] = torch.full((), float(max(host_steps, default=0)), dtype=torch.float32, device=p.device)
        for counter in counters.values():
            counter += 1

    def state_dict(self):
        for group in self.param_groups:      # capturable: bring the host-side `step` entries up to date before serialising
            for device, counter in group.get("_step_dev", {}).items():
                step = float(counter)
                for p in group["params"]:
                    if p in self.state and p.device == device:
                        self.state[p]["step"] = torch.tensor(step, dtype=torch.float32)
        out = super().state_dict()
        for g in out["param_groups"]:
            g.pop("_step_dev", None)
        return out
