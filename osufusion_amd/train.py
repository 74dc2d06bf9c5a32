"""Training engine for the hot loop of trainer.py:293-309: forward, backward, gradient all-reduce, clip, AdamW.

MI355X-first choices (nothing here mirrors accelerate/DDP's object model):
  * all parameters live in ONE flat fp32 buffer (and their grads / Adam moments in three more): the optimizer is a single
    HIP kernel launch over 343 M elements, the grad-norm is one reduction, and the data-parallel exchange is a handful of
    large contiguous RCCL all-reduces instead of ~1.2 k per-tensor ops;
  * one process per GPU; gradients are reduced bucket-by-bucket as soon as a bucket's last gradient has been accumulated,
    asynchronously, overlapping the rest of backward; the flat layout follows the OBSERVED completion order of the first
    backward, so buckets complete front to back; xGMI is point-to-point so buckets are large (64 MiB default) -- a few big
    rings, not many small ones.  (Not yet run on more than one physical GPU: the round's box has one; covered by 2-rank
    gloo tests on CPU and on one GPU.)  The layout is agreed across ranks: rank 0's observed order is broadcast and a
    fingerprint of (name, offset, size) is compared on every rank before the first positional all-reduce over a new layout.
  * no host sync in the step: the clip coefficient and the 1/world averaging are a device scalar read by the AdamW kernel.
"""
from __future__ import annotations

import math
import os
import time
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import functional as Fn
from . import ops


class FlatParameters:
    """Re-homes every parameter of `module` into one flat fp32 buffer and gives each a persistent `.grad` view into a flat
    gradient buffer.  Initial layout: reverse registration order (a first guess at the order in which backward completes the
    gradients); `reorder` re-lays the buffers out in an observed completion order."""

    def __init__(self, module: nn.Module, align: int = 64) -> None:
        params = [p for p in module.parameters() if p.requires_grad]
        assert all(p.dtype == torch.float32 for p in params), "master parameters are fp32"
        self.align = align
        self.params: List[nn.Parameter] = list(reversed(params))
        device = self.params[0].device
        self.offsets, self.numel = self._layout(self.params)
        self.data = torch.zeros(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.numel, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.data[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def _layout(self, params):
        offsets, off = [], 0
        for p in params:
            offsets.append(off)
            off += (p.numel() + self.align - 1) // self.align * self.align
        return offsets, off

    def zero_grad(self) -> None:
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):          # a foreign zero_grad(set_to_none=True) may have dropped the views
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)

    def reorder(self, order: List[int], companions=()) -> list:
        """Lay the buffers out as params[order[0]], params[order[1]], ... (a permutation of the current indices).  The parameter
        values, the gradients and every companion buffer (Adam moments: same layout as `data`) move with their parameter.
        Returns the re-laid-out companions."""
        assert sorted(order) == list(range(len(self.params))), "order must be a permutation of the parameter indices"
        new_params = [self.params[i] for i in order]
        new_offsets, numel = self._layout(new_params)
        assert numel == self.numel
        moved = []
        with torch.no_grad():
            for buf in (self.data, self.grad, *companions):
                nb = torch.zeros_like(buf)
                for i, no in zip(order, new_offsets):
                    n = self.params[i].numel()
                    nb[no:no + n].copy_(buf[self.offsets[i]:self.offsets[i] + n])
                moved.append(nb)
            self.data, self.grad = moved[0], moved[1]
            for p, o in zip(new_params, new_offsets):
                p.data = self.data[o:o + p.numel()].view_as(p)
                p.grad = self.grad[o:o + p.numel()].view_as(p)
        self.params, self.offsets = new_params, new_offsets
        return moved[2:]


class GradReducer:
    """Bucketed, overlapped gradient all-reduce over the flat gradient buffer (the DDP semantics of trainer.py:264-269,301).

    A bucket is a contiguous slice of the flat gradient buffer; it is all-reduced (async, SUM) the moment its last parameter's
    gradient is complete.  "Complete" is autograd's post-accumulate-grad hook and nothing else: measured on the GPU
    (tools/check_grad_reports.py, torch 2.10) the hook fires for EVERY parameter of the graph -- also for those whose gradient the
    HIP kernels added straight into the flat buffer while the autograd Function returned None -- and it fires once, after the LAST
    use of the parameter has run its backward (a module applied twice, tied weights: still one hook, after both).  The kernels'
    own notification (functional.grad_done -> param_ready) always comes earlier and says only "one use is done", so it is recorded
    for diagnostics and never launches a bucket (round 1 launched on whichever came first: premature for a re-used parameter).
    `begin(sync=False)` makes a backward a no-sync micro-step of gradient accumulation (`accelerator.accumulate`,
    trainer.py:293-295): nothing is sent.  overlap=False reduces every bucket in finish()."""

    def __init__(self, flat: FlatParameters, bucket_mib: float = 64.0, group=None, overlap: bool = True,
                 comm_dtype: Optional[torch.dtype] = None) -> None:
        self.flat, self.group, self.bucket_mib, self.overlap = flat, group, bucket_mib, overlap
        # comm_dtype = torch.bfloat16: a bucket travels as bf16 (half the xGMI bytes: 0.69 GB per step instead of 1.37) -- cast into a
        # persistent staging buffer when it is launched, summed by the collective in bf16, copied back into the fp32 gradient in finish().
        # Off by default: the sum then carries one bf16 rounding per rank contribution (tests: within 1e-2 of the fp32 reduction).
        self.comm_dtype = comm_dtype if comm_dtype not in (None, torch.float32) else None
        self.staging = torch.empty(flat.numel, dtype=self.comm_dtype, device=flat.grad.device) if self.comm_dtype is not None else None
        self._staged: List[int] = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # OSUF_DIST_REHEARSE=1: run the whole reduction path on a ONE-rank group too (a 1-GPU box's rehearsal of the RCCL leg)
        self.enabled = self.world > 1 or (dist.is_initialized() and os.environ.get("OSUF_DIST_REHEARSE") == "1")
        self.sync = True                                   # False during no-sync accumulation micro-steps
        self.handles = []
        self._seen = set()
        self.order_log: List[int] = []                     # parameter indices in the order they completed (last backward)
        self.fired_early: List[int] = []                   # buckets launched from a completion report (before finish())
        self.next_bucket = 0                               # collectives are matched by ISSUE order: bucket b goes out only after b-1
        self.launched_bytes = 0                            # bytes handed to all_reduce in the last begin()..finish()
        self.launches = 0                                  # all_reduce calls of the last begin()..finish()
        self.wait_ms = 0.0                                 # host time finish() spent in handle.wait() (not hidden behind backward)
        self.out_of_order_completions = 0                  # buckets that completed before a lower-indexed one (held back)
        self._hooked = set()
        self.names = {}                                    # id(param) -> name, filled by Trainer for error messages
        self.direct_reports = 0
        self._done = set()
        self._fresh = True                                 # the next completion report opens a new backward
        self._explicit = False                             # begin() was called for the backward in progress
        self.duplicate_reports = 0                         # second completion reports of one parameter inside one begin()..finish()
        self.rebuild()

    def rebuild(self) -> None:
        """(Re)compute the buckets from the flat layout (after FlatParameters.reorder)."""
        flat = self.flat
        cap = int(self.bucket_mib * (1 << 20) / 4)
        self.bounds: List[List[int]] = []                  # [start, end) element ranges
        self.bucket_of: List[int] = []
        start, cur = 0, 0
        for i in range(len(flat.params)):
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
        self._index = {id(p): i for i, p in enumerate(flat.params)}
        for p in flat.params:                              # hooks look the index up at fire time: they survive a reorder
            if id(p) not in self._hooked:
                p.register_post_accumulate_grad_hook(self._hook)
                self._hooked.add(id(p))

    def begin(self, sync: bool = True) -> None:
        """Start of a backward: sync=False = gradient-accumulation micro-step without communication."""
        if self.handles:                                   # a previous backward died between launch and finish(): drain it first
            self.abort()
        self.sync = sync
        self._open()
        self._explicit = True

    def abort(self) -> None:
        """A forward / backward raised after buckets were launched: wait for what is on the wire and forget the half-done backward,
        so the next begin() starts clean (the caller zeroes the gradients: they hold a mix of reduced and local sums)."""
        for h in self.handles:
            try:
                h.wait()
            except Exception:                              # noqa: BLE001 -- the collective itself failed: nothing left to wait for
                pass
        self.handles = []
        self._staged = []
        self.pending = list(self.expected)
        self.next_bucket = 0
        self._seen, self._fresh, self._explicit = set(), True, False
        Fn.reset_grad_uses()

    def _open(self) -> None:
        self._seen = set()
        self.order_log = []
        self.fired_early = []
        self.pending = list(self.expected)
        self.next_bucket = 0
        self.launched_bytes, self.launches, self.wait_ms, self.out_of_order_completions = 0, 0, 0.0, 0
        self._fresh = False

    def param_ready(self, p) -> None:
        """functional.grad_done: one direct-accumulation kernel for `p` has been launched.  Diagnostic only (see the class note)."""
        self.direct_reports += 1

    def _hook(self, param) -> None:
        idx = self._index.get(id(param))
        if idx is not None:
            self._ready(idx)

    def param_complete(self, p) -> None:
        """functional.grad_used: the gradient of a parameter that is kept out of the autograd graph (so has no post-accumulate hook) is
        complete -- every forward use has written its share.  Same effect as the hook."""
        self._hook(p)

    def _ready(self, idx: int) -> None:
        if self._fresh:                                    # first report after a finish(): a new backward (begin() is optional)
            self._open()
        if idx in self._seen and self._explicit:
            # Between begin() and finish() a second report is a DUPLICATE (a parameter that reaches the reducer both through its
            # AccumulateGrad hook and through functional.grad_used), never a new backward: do not reset the bookkeeping.  It is
            # harmless while the parameter's bucket is still waiting for others and an error once that bucket is on the wire.
            self.duplicate_reports += 1
            if self.enabled and self.sync and self.overlap and self.bucket_of[idx] in self.fired_early:
                who = self.names.get(id(self.flat.params[idx]), f"#{idx}")
                raise RuntimeError(f"parameter {who} reported its gradient complete twice in one backward and its bucket was already "
                                   "all-reduced after the first report: the second contribution would be lost")
            return
        if idx in self._seen:                              # one AccumulateGrad node per parameter: a second hook = a new backward
            if self.enabled and self.sync and self.overlap and self.handles:
                who = self.names.get(id(self.flat.params[idx]), f"#{idx}")
                raise RuntimeError(f"parameter {who}: a new backward started while buckets of the previous one are in flight -- "
                                   "call finish() (or begin()) between backwards")
            self._open()
        self._seen.add(idx)
        self.order_log.append(idx)
        if not (self.enabled and self.sync and self.overlap):
            return
        b = self.bucket_of[idx]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            # RCCL / gloo pair the collectives of a communicator by the order in which each rank ISSUES them, so every rank must issue
            # the buckets in one fixed order whatever order its own backward completed them in: index order, as torch DDP does.  A
            # bucket that completes ahead of a lower-indexed one is held until that one has gone out.
            if b != self.next_bucket:
                self.out_of_order_completions += 1
            while self.next_bucket < len(self.bounds) and self.pending[self.next_bucket] == 0:
                self.fired_early.append(self.next_bucket)
                self._launch(self.next_bucket)
                self.next_bucket += 1

    def _launch(self, b: int) -> None:
        s, e = self.bounds[b]
        buf = self.flat.grad[s:e]
        if self.staging is not None:
            buf = self.staging[s:e]
            buf.copy_(self.flat.grad[s:e])                 # fp32 -> comm dtype on the current stream, ahead of the collective
            self._staged.append(b)
        self.launched_bytes += (e - s) * buf.element_size()
        self.launches += 1
        self.handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        """Wait for the in-flight buckets and reduce, still in index order, every bucket that has not gone out yet (a bucket with an
        unused parameter never completes, and it holds back every bucket after it)."""
        if self.enabled and self.sync:
            for b in range(self.next_bucket, len(self.bounds)):
                self._launch(b)
            self.next_bucket = len(self.bounds)
            t0 = time.perf_counter()
            for h in self.handles:
                h.wait()
            self.wait_ms = (time.perf_counter() - t0) * 1e3
            for b in self._staged:                         # reduced sums back into the fp32 gradient
                s, e = self.bounds[b]
                self.flat.grad[s:e].copy_(self.staging[s:e])
        self._staged = []
        self.handles = []
        self.pending = list(self.expected)
        self._done, self._fresh, self._explicit = set(self._seen), True, False
        Fn.reset_grad_uses()

    def observed_order(self) -> List[int]:
        """Parameter indices in the order the last finished backward completed them; never-reported ones follow, in place."""
        rest = [i for i in range(len(self.flat.params)) if i not in self._done]
        return list(self.order_log) + rest


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
    """One object per rank: flat parameters, (optional) overlapped RCCL gradient reduction, fused AdamW.

    gradient_accumulation_steps = k reproduces the reference loop `for _ in range(k): with accelerator.accumulate(model): ...`
    (trainer.py:293-309): each step() call is one micro-batch; the first k-1 add their gradients into the flat buffer without
    any communication (no-sync), the k-th also reduces, clips and applies AdamW on the mean over all k * world micro-batches.
    After the first optimizer step the flat buffers are re-laid out in the order in which that backward completed the
    gradients (reorder_buckets), so that bucket i is complete -- and on the wire -- before bucket i+1."""

    def __init__(self, model: nn.Module, lr: float = 1e-4, weight_decay: float = 1e-2, clip_grad_norm: float = 0.0,
                 bucket_mib: float = 64.0, compute_dtype: Optional[torch.dtype] = torch.bfloat16,
                 gradient_accumulation_steps: int = 1, overlap: bool = True, reorder_buckets: bool = True,
                 comm_dtype: Optional[torch.dtype] = None) -> None:
        assert gradient_accumulation_steps >= 1
        self.model = model
        self.flat = FlatParameters(model)
        self.reducer = GradReducer(self.flat, bucket_mib, overlap=overlap, comm_dtype=comm_dtype)
        self.reducer.names = {id(p): n for n, p in model.named_parameters()}
        self.opt = FusedAdamW(self.flat, lr=lr, weight_decay=weight_decay)
        self.clip = clip_grad_norm
        self.compute_dtype = compute_dtype
        self.accum = gradient_accumulation_steps
        self.micro = 0                                      # micro-batches accumulated since the last optimizer step
        self._reorder_pending = reorder_buckets
        self.arena = ops.ZeroArena(self.flat.data.device) if self.flat.data.is_cuda else None
        self.order_disagreements = 0
        if self.reducer.enabled:                            # identical replicas (guard; inits are already deterministic)
            dist.broadcast(self.flat.data, src=0)
            self.check_layout_agreement()
        Fn.bump_weight_epoch()
        Fn.enable_direct_grads(True, self.reducer.param_ready, self.reducer.param_complete)

    def step(self, x, a, c, noise=None, timesteps=None, orig_len=None):
        """One micro-batch.  Returns (loss, total_norm): total_norm is the device scalar of the clipped step's gradient norm on
        the micro-batch that applied the optimizer, None on accumulation-only micro-batches."""
        from .runtime import forced_compute_dtype
        if self.micro == 0:
            self.flat.zero_grad()
        last = self.micro + 1 == self.accum
        self.reducer.begin(sync=last)
        if self.arena is not None:
            self.arena.begin()                              # one fill clears what the previous step's accumulators dirtied
        ops.set_zero_arena(self.arena)
        try:
            with forced_compute_dtype(self.compute_dtype):
                if noise is None:
                    loss = self.model(x, a, c, orig_len)
                else:
                    loss = self.model.loss_with(x, a, c, noise, timesteps, orig_len)
                loss.backward()
        except BaseException:
            # buckets of the dead backward may be on the wire and the flat gradient holds a partial sum: drain, and restart the
            # accumulation window (trainer.py:296-299 skips the batch on an AssertionError and goes on with the next one)
            self.reducer.abort()
            self.micro = 0
            raise
        finally:
            ops.set_zero_arena(None)
        self.reducer.finish()
        self.micro += 1
        if not last:
            return loss.detach(), None
        self.micro = 0
        total_norm = self.opt.step(grad_scale=1.0 / (self.reducer.world * self.accum), clip_grad_norm=self.clip)
        if self._reorder_pending:
            self._reorder_pending = False
            self.apply_observed_order()
        Fn.refresh_packs()                                  # every packed GEMM operand went stale with that step: one grouped launch
        return loss.detach(), total_norm

    def apply_observed_order(self) -> None:
        """Re-lay the flat buffers (parameters, gradients, Adam moments) out in the gradient-completion order of the last
        backward and rebuild the buckets over the new layout."""
        order = self.reducer.observed_order()
        if self.reducer.enabled:
            order = self._agree_on_order(order)
        if order != list(range(len(order))):
            self.opt.exp_avg, self.opt.exp_avg_sq = self.flat.reorder(order, (self.opt.exp_avg, self.opt.exp_avg_sq))
            self.reducer.rebuild()
            Fn.bump_weight_epoch()                          # parameter storage moved: cached operand packs key on data_ptr
        self.check_layout_agreement()

    def _comm_device(self) -> torch.device:
        """Where small control tensors of a collective must live: the GPU for RCCL, the host for gloo."""
        backend = str(dist.get_backend(self.reducer.group)).lower()
        return self.flat.data.device if "nccl" in backend else torch.device("cpu")

    def _agree_on_order(self, order: List[int]) -> List[int]:
        """Every rank must lay its flat buffers out identically: the bucketed all-reduce is positional.  Rank 0's observed order is
        broadcast and used everywhere (what torch DDP does with its rebuilt bucket order); a rank whose own backward completed the
        gradients in another order is counted in `order_disagreements`.  Such a rank is still correct -- GradReducer issues the buckets
        in index order on every rank, never in completion order -- it only sends a bucket later than it could have."""
        dev = self._comm_device()
        t = torch.tensor(order, dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0, group=self.reducer.group)
        agreed = [int(v) for v in t.cpu().tolist()]
        if sorted(agreed) != list(range(len(order))):
            raise RuntimeError("rank 0 broadcast an order that is not a permutation of this rank's parameters: the ranks hold different models")
        self.order_disagreements = getattr(self, "order_disagreements", 0) + int(agreed != list(order))
        return agreed

    def comm_stats(self) -> dict:
        """What the last step's gradient reduction did on this rank -- enough for a first multi-GPU run to diagnose itself."""
        r = self.reducer
        return {"backend": str(dist.get_backend(r.group)) if r.enabled else None, "world": r.world, "buckets": len(r.bounds),
                "bucket_mib": r.bucket_mib, "comm_dtype": str(r.comm_dtype or torch.float32).replace("torch.", ""), "allreduce_calls": r.launches, "allreduce_bytes": r.launched_bytes,
                "buckets_fired_before_finish": len(r.fired_early), "out_of_order_completions": r.out_of_order_completions,
                "finish_wait_ms": round(r.wait_ms, 3), "order_disagreements": self.order_disagreements,
                "layout_fingerprint": self.layout_fingerprint()}

    def layout_fingerprint(self) -> int:
        """63-bit hash of (parameter name, offset, numel) in flat order."""
        import hashlib
        h = hashlib.sha256()
        for p, o in zip(self.flat.params, self.flat.offsets):
            h.update(f"{self.reducer.names.get(id(p), '?')}:{o}:{p.numel()};".encode())
        return int.from_bytes(h.digest()[:8], "little") >> 1

    def check_layout_agreement(self) -> None:
        """Raise unless every rank's flat layout (names, offsets, sizes) is the same.  Called after construction and after every
        re-layout: a silent mismatch would make ranks sum different parameters into each other."""
        if not self.reducer.enabled:
            return
        dev = self._comm_device()
        mine = torch.tensor([self.layout_fingerprint()], dtype=torch.int64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(self.reducer.world)]
        dist.all_gather(every, mine, group=self.reducer.group)
        vals = [int(v.item()) for v in every]
        if len(set(vals)) != 1:
            raise RuntimeError(f"flat parameter layouts differ across ranks (fingerprints {vals}): refusing to all-reduce positionally")

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
