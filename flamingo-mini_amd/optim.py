"""FusedAdamW: torch.optim.Optimizer whose step() is one multi-tensor HIP kernel sweep (ff_adamw_step) over all parameters
of a dtype — same update rule, defaults and state_dict layout (`step`, `exp_avg`, `exp_avg_sq`) as torch.optim.AdamW, so
optimizer checkpoints interchange.  The reference trains with `--optim adamw_torch` (training/train.sh:10-13) on
`model.parameters_trainable()`.

`capturable=True` keeps the step count AND the learning rate in device scalars per parameter group (bias corrections are computed
in the kernel), so `step()` can be captured into a HIP graph and replayed (graphs.GraphedTrainStep) while an LR scheduler keeps
changing `group["lr"]`: call `sync_device_hyperparams()` (GraphedTrainStep does) before a replay.  `state_dict()` reads the count back.

Mixed precision (the reference trains with `--fp16` autocast = fp32 master weights and fp32 moments, training/train.sh:24):
`master_dtype=torch.float32` keeps an fp32 master copy of every bf16 parameter in the optimizer state (`state["master"]`); the kernel
updates the master copy and writes its bf16 rounding into the parameter in the same pass, so steps far below the bf16 resolution of a
weight (lr 1e-4) accumulate instead of vanishing.  `state_dtype=torch.float32` alone keeps only the two moments in fp32."""
from __future__ import annotations

import ctypes as C
from typing import Iterable

import torch

from . import ffi


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 grad_scale: float = 1.0, capturable: bool = False, master_dtype=None, state_dtype=None):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        if master_dtype not in (None, torch.float32) or state_dtype not in (None, torch.float32):
            raise ValueError("master_dtype / state_dtype: None (the parameter's dtype) or torch.float32")
        if master_dtype is not None:
            state_dtype = torch.float32                   # fp32 masters go with fp32 moments
        self.master_dtype, self.state_dtype = master_dtype, state_dtype
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale, capturable=capturable))

    @torch.no_grad()
    def step(self, closure=None, *, only=None, advance: bool = True):
        """only / advance (capturable mode): update just the parameters whose id() is in `only` - one of several calls that together cover every
        parameter with a gradient exactly once per training step (graphs.PiecewiseGraphedTrainStep(overlap_optimizer=True) updates the
        parameters of a backward segment while the next segments still run).  The FIRST partial call of a training step passes advance=True
        (the device step counters count training steps, not calls) and must execute before the others."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = ffi.lib()
        if advance and not torch.cuda.is_current_stream_capturing():
            # an eager training loop cannot run through a timed-out in-launch hand-off of the fused cross-attention kernels unnoticed:
            # looks at the status word the previous step copied to pinned memory, enqueues the next copy; never synchronises
            from . import functional as _F
            _F.poll_sync_exchange("FusedAdamW.step")
        if only is not None:
            only = frozenset(only)
            if not all(g.get("capturable", False) for g in self.param_groups):
                raise ValueError("FusedAdamW.step(only=...) needs capturable=True (step counts live in device scalars shared by the partial calls)")
        for gi, group in enumerate(self.param_groups):
            capturable = group.get("capturable", False)
            if capturable and advance:
                self._advance_device_steps(group)
            for bucket in self._buckets(gi, group, only):
                params, grad_ptrs, n = bucket["params"], bucket["grad_ptrs"], len(bucket["params"])
                for i, p in enumerate(params):           # only the gradient addresses change from step to step
                    g = p.grad
                    if g.dtype != p.dtype or not g.is_contiguous():
                        raise ffi.FusionLibraryError("FusedAdamW needs contiguous gradients of the parameter's dtype")
                    grad_ptrs[i] = g.data_ptr()
                lr_dev = None
                if capturable:
                    step, step_dev = 0, group["_step_dev"][bucket["device"]].data_ptr()
                    lr_dev = group["_lr_dev"][bucket["device"]].data_ptr()
                else:
                    bucket["step"] += 1                   # the per-parameter `step` entries are refreshed lazily (_sync_host_steps)
                    step, step_dev = bucket["step"], None
                desc = ffi.AdamWDesc(bucket["dtype_code"], n, step, group["lr"], group["betas"][0], group["betas"][1], group["eps"],
                                     group["weight_decay"], group["grad_scale"], step_dev)
                ffi.check(lib.ff_adamw_step_mixed(desc, bucket["state_code"], bucket["param_ptrs"], grad_ptrs, bucket["m_ptrs"], bucket["v_ptrs"],
                                                  bucket["w_ptrs"], lr_dev, bucket["numels"], ffi.stream_handle(bucket["device"])),
                          "ff_adamw_step_mixed")
        return loss

    def sync_device_hyperparams(self) -> None:
        """capturable mode: copy `group["lr"]` into the device scalar the captured kernels read (call before replaying a graph)."""
        for group in self.param_groups:
            for dev, t in group.get("_lr_dev", {}).items():
                if group.get("_lr_on_dev", {}).get(dev) != group["lr"]:
                    t.fill_(float(group["lr"]))
                    group.setdefault("_lr_on_dev", {})[dev] = group["lr"]

    def _buckets(self, gi, group, only=None):
        """Parameters with a gradient, grouped by (dtype, device, step count); the pointer tables of everything that does not
        change between steps (parameters, moments, sizes) are built once and reused while the same parameters have gradients."""
        active = [p for p in group["params"] if p.grad is not None and (only is None or id(p) in only)]
        cache = self.__dict__.setdefault("_bucket_cache", {})
        key = tuple(id(p) for p in active)
        slot = gi if only is None else (gi, only)
        hit = cache.get(slot)
        if hit is not None and hit[0] == key and all(b["param_ptrs"][0] == b["params"][0].data_ptr() for b in hit[1]):
            return hit[1]
        self._sync_host_steps()
        table = {}
        for p in active:
            ffi.require_cuda(p, p.grad)
            if not p.is_contiguous():
                raise ffi.FusionLibraryError("FusedAdamW needs contiguous parameters")
            st = self.state[p]
            mixed = p.dtype == torch.bfloat16          # fp32 parameters are their own master copy and already have fp32 moments
            sdt = self.state_dtype if (mixed and self.state_dtype is not None) else p.dtype
            if not st:
                st["step"] = torch.zeros((), dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, dtype=sdt, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, dtype=sdt, memory_format=torch.preserve_format)
            if mixed and self.master_dtype is not None and "master" not in st:
                st["master"] = p.detach().to(torch.float32)
            has_master = "master" in st
            table.setdefault((p.dtype, p.device, int(st["step"]), st["exp_avg"].dtype, has_master), []).append(p)
        buckets = []
        for (dtype, device, step, sdt, has_master), params in table.items():
            n = len(params)
            buckets.append(dict(params=params, device=device, dtype_code=ffi.dtype_code(dtype), step=step, state_code=ffi.dtype_code(sdt),
                                w_ptrs=ffi.ptr_array([self.state[p]["master"] for p in params]) if has_master else None,
                                param_ptrs=ffi.ptr_array(params), grad_ptrs=(C.c_void_p * n)(),
                                m_ptrs=ffi.ptr_array([self.state[p]["exp_avg"] for p in params]),
                                v_ptrs=ffi.ptr_array([self.state[p]["exp_avg_sq"] for p in params]),
                                numels=(C.c_longlong * n)(*[p.numel() for p in params])))
        cache[slot] = (key, buckets)
        return buckets

    # ------------------------------------------------------------------ capturable mode
    def _advance_device_steps(self, group):
        """One float32 step counter per (group, device), advanced by a device-side add (captured along with the update)."""
        counters = group.setdefault("_step_dev", {})
        lrs = group.setdefault("_lr_dev", {})
        for p in group["params"]:
            if p.grad is not None and p.device not in counters:
                host_steps = [int(self.state[q]["step"]) for q in group["params"] if q in self.state and "step" in self.state[q]]
                counters[p.device] = torch.full((), float(max(host_steps, default=0)), dtype=torch.float32, device=p.device)
                lrs[p.device] = torch.full((), float(group["lr"]), dtype=torch.float32, device=p.device)
                group.setdefault("_lr_on_dev", {})[p.device] = group["lr"]
        if not torch.cuda.is_current_stream_capturing():
            self.sync_device_hyperparams()
        for counter in counters.values():
            counter += 1

    def _sync_host_steps(self):
        """Write the step counts kept per bucket (host mode) back into the per-parameter state entries."""
        for gi, group in enumerate(self.param_groups):
            if group.get("capturable", False):
                continue
            hit = self.__dict__.get("_bucket_cache", {}).get(gi)
            for bucket in (hit[1] if hit else ()):
                for p in bucket["params"]:
                    self.state[p]["step"] = torch.tensor(float(bucket["step"]), dtype=torch.float32)

    def load_state_dict(self, state_dict):
        """torch's Optimizer.load_state_dict casts every floating-point state tensor to the PARAMETER's dtype, which would turn the fp32
        moments / master copies of a bf16 parameter into bf16 (and the next step would then either raise or silently continue with bf16
        moments).  The tensors of the incoming state_dict are therefore put back in their own precision afterwards: `master` always in
        fp32, the moments in `state_dtype` (the precision this optimizer was built for) or, without one, the checkpoint's."""
        import copy
        incoming = state_dict["state"]
        saved_groups = state_dict["param_groups"]
        super().load_state_dict(copy.copy(state_dict))
        # map the checkpoint's integer ids to this optimizer's parameters (same order: torch's own rule)
        ids = [i for g in saved_groups for i in g["params"]]
        params = [p for g in self.param_groups for p in g["params"]]
        for i, p in zip(ids, params):
            src = incoming.get(i)
            st = self.state.get(p)
            if src is None or st is None:
                continue
            mixed = p.dtype == torch.bfloat16
            for key in ("exp_avg", "exp_avg_sq", "master"):
                if key not in src or not torch.is_tensor(src[key]):
                    continue
                want = torch.float32 if key == "master" else (self.state_dtype if (mixed and self.state_dtype is not None) else
                                                              (src[key].dtype if mixed else p.dtype))
                if st[key].dtype != want or st[key].dtype != src[key].dtype:
                    st[key] = src[key].detach().to(device=p.device, dtype=want).clone(memory_format=torch.preserve_format)
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].detach().to("cpu", torch.float32)
            if mixed and self.master_dtype is None and "master" in st:
                del st["master"]                       # this optimizer keeps no master copies: the parameter itself is the weight
        for group in self.param_groups:                # device-side counters of the capturable mode restart from the loaded step counts
            for k in ("_step_dev", "_lr_dev", "_lr_on_dev"):
                group.pop(k, None)
        self.__dict__.pop("_bucket_cache", None)       # moments were replaced: rebuild the pointer tables

    def state_dict(self):
        self._sync_host_steps()
        for group in self.param_groups:      # capturable: bring the host-side `step` entries up to date before serialising
            for device, counter in group.get("_step_dev", {}).items():
                step = float(counter)
                for p in group["params"]:
                    if p in self.state and p.device == device:
                        self.state[p]["step"] = torch.tensor(step, dtype=torch.float32)
        out = super().state_dict()
        for g in out["param_groups"]:
            for k in ("_step_dev", "_lr_dev", "_lr_on_dev"):
                g.pop(k, None)
        return out
