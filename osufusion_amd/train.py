"""Training engine for the hot loop of trainer.py:293-309: forward, backward, gradient all-reduce, clip, AdamW.

MI355X-first choices (nothing here mirrors accelerate/DDP's object model):
  * all parameters live in ONE flat fp32 buffer (and their grads / Adam moments in three more): the optimizer is a single
    HIP kernel launch over 343 M elements, the grad-norm is one reduction, and the data-parallel exchange is a handful of
    large contiguous RCCL all-reduces instead of ~1.2 k per-tensor ops;
  * one process per GPU; gradients are reduced bucket-by-bucket (reverse-autograd order) as soon as a bucket's last
    gradient has been accumulated, on RCCL's own stream, overlapping the rest of backward; xGMI is point-to-point so buckets
    are large (64 MiB default) -- a few big rings, not many small ones;
  * no host sync in the step: the clip coefficient and the 1/world averaging are a device scalar read by the AdamW kernel.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import functional as Fn
from . import ops


class FlatParameters:
    """Re-homes every parameter of `module` into one flat fp32 buffer (reverse registration order, so the gradients that
    autograd produces first sit at the front) and gives each a persistent `.grad` view into a flat gradient buffer."""

    def __init__(self, module: nn.Module, align: int = 64) -> None:
        params = [p for p in module.parameters() if p.requires_grad]
        assert all(p.dtype == torch.float32 for p in params), "master parameters are fp32"
        self.params: List[nn.Parameter] = list(reversed(params))
        device = self.params[0].device
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + align - 1) // align * align
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.data[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def zero_grad(self) -> None:
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):          # a foreign zero_grad(set_to_none=True) may have dropped the views
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)


class GradReducer:
    """Bucketed, overlapped gradient all-reduce over the flat gradient buffer (the DDP semantics of trainer.py:264-269,301)."""

    def __init__(self, flat: FlatParameters, bucket_mib: float = 64.0, group=None) -> None:
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        cap = int(bucket_mib * (1 << 20) / 4)
        self.bounds: List[List[int]] = []                  # [start, end) element ranges
        self.bucket_of: List[int] = []
        start, cur = 0, 0
        for i, (p, o) in enumerate(zip(flat.params, flat.offsets)):
            end = flat.offsets[i + 1] if i + 1 < len(flat.offsets) else flat.numel
            self.bucket_of.append(len(self.bounds))
            cur = end
            if cur - start >= cap:
                self.bounds.append([start, cur])
                start = cur
        if cur > start:
            self.bounds.append([start, cur])
        self.bucket_of = [min(b, len(self.bounds) - 1) for b in self.bucket_of]
        self.expected = [0] * len(self.bounds)
        for b in self.bucket_of:
            self.expected[b] += 1
        self.pending = list(self.expected)
        self.handles = []
        self.enabled = self.world > 1
        self._index = {id(p): i for i, p in enumerate(flat.params)}
        self._seen = set()
        if self.enabled:
            for idx, p in enumerate(flat.params):
                p.register_post_accumulate_grad_hook(self._make_hook(idx))

    def param_ready(self, p) -> None:
        """Called by the kernels' direct-accumulation path (functional.grad_done): the gradient of `p` is complete."""
        if self.enabled:
            idx = self._index.get(id(p))
            if idx is not None:
                self._ready(idx)

    def _ready(self, idx: int) -> None:
        if idx in self._seen:                              # a parameter counts once per step, however its gradient arrived
            return
        self._seen.add(idx)
        b = self.bucket_of[idx]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            s, e = self.bounds[b]
            self.handles.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _make_hook(self, idx: int):
        def hook(_param):
            self._ready(idx)
        return hook

    def finish(self) -> None:
        """Wait for the in-flight buckets (and reduce any bucket whose hooks did not all fire, e.g. unused parameters)."""
        if not self.enabled:
            return
        for b, left in enumerate(self.pending):
            if left > 0:
                s, e = self.bounds[b]
                self.handles.append(dist.all_reduce(self.flat.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in self.handles:
            h.wait()
        self.handles = []
        self.pending = list(self.expected)
        self._seen = set()


class FusedAdamW:
    """torch.optim.AdamW semantics (trainer.py:230,307) as one kernel over the flat buffers; lr may be changed per step."""

    def __init__(self, flat: FlatParameters, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2) -> None:
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.step_count = 0
        dev = flat.data.device
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.coef = torch.ones(1, dtype=torch.float32, device=dev)
        self.total_norm = torch.zeros(1, dtype=torch.float32, device=dev)

    def step(self, grad_scale: float = 1.0, clip_grad_norm: float = 0.0) -> torch.Tensor:
        """grad_scale: e.g. 1/world after a SUM all-reduce.  Returns the (device) total gradient norm of the scaled gradient."""
        self.step_count += 1
        self.sumsq.zero_()
        ops.sqnorm(self.flat.grad, self.sumsq)
        # total_norm of the averaged gradient = grad_scale * sqrt(sumsq); clip coefficient computed on device
        ops.clip_coef(self.sumsq, clip_grad_norm / grad_scale if clip_grad_norm > 0 else 0.0, grad_scale, self.coef, self.total_norm)
        ops.adamw(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0], self.betas[1], self.eps,
                  self.weight_decay, self.step_count, self.coef)
        Fn.bump_weight_epoch()
        return self.total_norm * grad_scale


    # -- torch.optim.AdamW-format state (checkpoint.pt["optimizer_state_dict"], trainer.py:163-171,189-199) ----------------
    def _model_order(self, module: nn.Module) -> List[int]:
        """flat index of each trainable parameter, in module.parameters() order (the order torch.optim indexes them in)."""
        pos = {id(p): i for i, p in enumerate(self.flat.params)}
        return [pos[id(p)] for p in module.parameters() if p.requires_grad]

    def state_dict(self, module: nn.Module) -> dict:
        order = self._model_order(module)
        state = {}
        if self.step_count > 0:
            for k, i in enumerate(order):
                p, o = self.flat.params[i], self.flat.offsets[i]
                state[k] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + p.numel()].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view_as(p).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, module: nn.Module, sd: dict) -> None:
        order = self._model_order(module)
        group = sd["param_groups"][0]
        if len(group["params"]) != len(order):
            raise ValueError(f"optimizer state has {len(group['params'])} parameters, the model has {len(order)} trainable ones")
        self.lr, self.betas, self.eps, self.weight_decay = group["lr"], tuple(group["betas"]), group["eps"], group["weight_decay"]
        steps = set()
        with torch.no_grad():
            for k, i in enumerate(order):
                st = sd["state"].get(k)
                if st is None:
                    continue
                p, o = self.flat.params[i], self.flat.offsets[i]
                self.exp_avg[o:o + p.numel()].view_as(p).copy_(st["exp_avg"])
                self.exp_avg_sq[o:o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ; the fused kernel keeps one step count for all parameters")
        self.step_count = steps.pop() if steps else 0


def cosine_warmup_lr(step: int, base_lr: float, warmup: int, total: int, num_cycles: float = 0.5) -> float:
    """diffusers.get_cosine_schedule_with_warmup as called at trainer.py:231-236."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    progress = (step - warmup) / max(1, total - warmup)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * progress)))


class Trainer:
    """One object per rank: flat parameters, (optional) overlapped RCCL gradient reduction, fused AdamW."""

    def __init__(self, model: nn.Module, lr: float = 1e-4, weight_decay: float = 1e-2, clip_grad_norm: float = 0.0,
                 bucket_mib: float = 64.0, compute_dtype: Optional[torch.dtype] = torch.bfloat16) -> None:
        self.model = model
        self.flat = FlatParameters(model)
        self.reducer = GradReducer(self.flat, bucket_mib)
        self.opt = FusedAdamW(self.flat, lr=lr, weight_decay=weight_decay)
        self.clip = clip_grad_norm
        self.compute_dtype = compute_dtype
        if self.reducer.enabled:                            # identical replicas (guard; inits are already deterministic)
            dist.broadcast(self.flat.data, src=0)
        Fn.bump_weight_epoch()
        Fn.enable_direct_grads(True, self.reducer.param_ready)

    def step(self, x, a, c, noise=None, timesteps=None, orig_len=None):
        from .runtime import forced_compute_dtype
        self.flat.zero_grad()
        with forced_compute_dtype(self.compute_dtype):
            if noise is None:
                loss = self.model(x, a, c, orig_len)
            else:
                loss = self.model.loss_with(x, a, c, noise, timesteps, orig_len)
            loss.backward()
        self.reducer.finish()
        total_norm = self.opt.step(grad_scale=1.0 / self.reducer.world, clip_grad_norm=self.clip)
        return loss.detach(), total_norm

    # -- checkpoint.pt in the reference's layout (trainer.py:148-203): resumable by either trainer ------------------------
    def state_dict(self, scheduler_state: Optional[dict] = None) -> dict:
        return {"model_state_dict": self.model.state_dict(), "optimizer_state_dict": self.opt.state_dict(self.model),
                "scheduler_state_dict": scheduler_state if scheduler_state is not None else {}, "rng_state": torch.get_rng_state()}

    def load_state_dict(self, checkpoint: dict, strict: bool = True) -> None:
        self.model.load_state_dict(checkpoint["model_state_dict"], strict=strict)      # copies into the flat buffer's views
        self.opt.load_state_dict(self.model, checkpoint["optimizer_state_dict"])
        if "rng_state" in checkpoint and checkpoint["rng_state"] is not None:
            torch.set_rng_state(checkpoint["rng_state"].cpu())
        Fn.bump_weight_epoch()
