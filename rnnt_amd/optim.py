"""Gradient-norm clip + AdamW on the HIP engine: the two statements that
follow loss.backward() in the reference's loop,

    total_norm = torch.nn.utils.clip_grad_norm_(params, cfg.training.clip_grad_norm)   # rnnt/train.py:136
    optimizer.step()      # torch.optim.AdamW(lr, betas, eps, weight_decay)             # rnnt/train.py:164

as multi-tensor kernels reached through the C ABI (rnnt_engine_grad_norm, rnnt_engine_adamw_step):
one launch per 40 parameter tensors instead of one per tensor, and the clip coefficient is read on
the device, so nothing synchronises with the host between backward and step.

    optimizer:                       # rnnt/config/*.yaml training.optimizer
      _target_: rnnt_amd.optim.AdamW
      lr: 3e-4 ...

Reference behaviour kept: `clip_grad_norm_` given an exhausted generator (rnnt/train.py:95 makes
`params` a generator that the optimizer's constructor consumes, train.py:104) clips nothing and
returns 0, exactly like torch's.  fp32 CUDA/HIP parameters only; anything else raises.
"""
import ctypes
from typing import Iterable, List

import torch

from . import engine

__all__ = ["AdamW", "clip_grad_norm_", "BucketGradNorm"]


def _ptr_array(tensors: List[torch.Tensor]):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _numel_array(tensors: List[torch.Tensor]):
    return (ctypes.c_int64 * len(tensors))(*[t.numel() for t in tensors])


def _check_group(tensors: List[torch.Tensor], what: str):
    dev = tensors[0].device
    for t in tensors:
        if t.device.type != "cuda" or t.device != dev:
            raise RuntimeError(f"rnnt_amd.optim: {what} must all live on one HIP device (got {t.device})")
        if t.dtype != torch.float32:
            raise RuntimeError(f"rnnt_amd.optim: {what} must be float32 (got {t.dtype})")
        if not t.is_contiguous():
            raise RuntimeError(f"rnnt_amd.optim: {what} must be contiguous")
    return dev


def grad_norm(grads: List[torch.Tensor]) -> torch.Tensor:
    """2-norm of all `grads` as a 0-dim device tensor (no host sync)."""
    dev = _check_group(grads, "gradients")
    lib = engine.lib()
    with torch.cuda.device(dev):
        out = torch.empty(1, dtype=torch.float32, device=dev)
        n = ctypes.c_size_t(0)
        numels = _numel_array(grads)
        engine._check(lib.rnnt_engine_grad_norm_workspace_bytes(len(grads), numels, ctypes.byref(n)))
        ws = torch.empty(int(n.value), dtype=torch.uint8, device=dev)  # caching allocator: no sync
        engine._check(lib.rnnt_engine_grad_norm(len(grads), _ptr_array(grads), numels, engine._p(out),
                                                engine._p(ws), ctypes.c_size_t(ws.numel()),
                                                engine._stream(dev)))
    return out[0]


def clip_grad_norm_(parameters: Iterable[torch.Tensor], max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) (norm_type 2) on the engine: returns the
    total norm (0-dim device tensor) and scales every .grad in place by
    min(1, max_norm / (total_norm + 1e-6))."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad for p in parameters if p.grad is not None]
    if len(grads) == 0:  # also the reference's exhausted generator: nothing to clip, norm 0
        return torch.tensor(0.0)
    total = grad_norm(grads)
    coef = torch.clamp(float(max_norm) / (total + 1e-6), max=1.0)
    torch._foreach_mul_(grads, coef)
    return total


class BucketGradNorm:
    """The gradient norm of a DDP model, accumulated bucket by bucket as the all-reduces complete — the pass over
    the gradients that `clip_grad_norm_` (rnnt/train.py:136) makes after backward then overlaps the remaining
    all-reduces and the rest of backward (the reference wraps the model in DDP, rnnt/train.py:68, and steps at :164).

        norm = rnnt_amd.optim.BucketGradNorm(ddp_model)                  # registers a DDP communication hook
        optimizer = rnnt_amd.optim.AdamW(params, ..., max_grad_norm=clip, norm_source=norm)
        loss.backward(); optimizer.step()                                # no separate norm pass

    The hook does what DDP's default one does (divide the bucket by the world size, all-reduce it) and, chained to
    that, one engine norm kernel on the reduced bucket on the stream the all-reduce completed on.  `total()` returns
    sqrt(sum of the buckets' squared norms) as a 0-dim device tensor (no host sync) and starts the next iteration.
    Every parameter with a gradient lives in exactly one bucket, so this is the norm `clip_grad_norm_` computes."""

    def __init__(self, ddp_model, process_group=None, norm_fn=None):
        import torch.distributed as dist
        self._dist = dist
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group)
        self._parts: List[torch.Tensor] = []
        # the norm of one reduced bucket: the engine's kernel (HIP tensors only — it raises on anything else).
        # `norm_fn` exists for the world-2 gloo test of the hook's arithmetic (tests/test_dist_cpu.py), where
        # the buckets are CPU tensors; nothing in the package passes it.
        self._norm = norm_fn if norm_fn is not None else (lambda t: grad_norm([t]))
        ddp_model.register_comm_hook(self, BucketGradNorm._hook)

    @staticmethod
    def _hook(state, bucket):
        buf = bucket.buffer()
        buf.div_(state.world_size)
        fut = state._dist.all_reduce(buf, group=state.process_group, async_op=True).get_future()

        def reduced(f):
            t = f.value()[0]
            n = state._norm(t)
            state._parts.append(n * n)
            return t

        return fut.then(reduced)

    def total(self) -> torch.Tensor:
        """Norm over the buckets reduced since the last call (call once per iteration, after backward)."""
        if not self._parts:
            return torch.tensor(0.0)
        parts, self._parts = self._parts, []
        return torch.sqrt(torch.stack(parts).sum())


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW(params, lr, betas, eps, weight_decay) — decoupled weight decay, no amsgrad,
    no maximize — whose step() is one multi-tensor engine call per parameter group.

    `max_grad_norm` (optional, > 0) fuses clip_grad_norm_ into step(): the gradient norm of ALL
    groups is reduced on the device and every gradient is scaled by min(1, max/(norm+1e-6)) on its
    way into the moments — the clip the reference meant at rnnt/train.py:136 — without touching the
    host.  `last_grad_norm` then holds the device scalar."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 max_grad_norm=None, capturable=False, norm_source=None):
        if float(lr) < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("rnnt_amd.optim.AdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm = None
        # optional BucketGradNorm: the norm was accumulated during backward, step() makes no pass of its own
        self.norm_source = norm_source
        # capturable=True (as torch.optim.AdamW's flag): the step count and the learning rate live on the
        # device (rnnt_engine_adamw_step_dev) — nothing of an update is baked into the launches, so step()
        # can be captured into a HIP graph together with forward and backward.  group["lr"] becomes a
        # device tensor that LR schedulers fill in place; every parameter must have a gradient each step.
        self.capturable = bool(capturable)
        self._dev = {}  # group index -> (step int64[1], hyper float32[4])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = engine.lib()
        total = None
        if self.max_grad_norm is not None and self.max_grad_norm > 0:
            allg = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
            if allg:
                total = self.norm_source.total().reshape(()) if self.norm_source is not None else grad_norm(allg)
                if total.device != allg[0].device:  # (no bucket was reduced: nothing to clip)
                    total = total.to(allg[0].device)
                total = total.contiguous()
                self.last_grad_norm = total
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            gs = [p.grad for p in ps]
            if any(g.is_sparse for g in gs):
                raise RuntimeError("rnnt_amd.optim.AdamW does not support sparse gradients")
            dev = _check_group(ps, "parameters")
            _check_group(gs, "gradients")
            # the kernels write parameters and moments through raw pointers: tell autograd (tensors saved for a pending backward, and
            # anything else keyed on `_version`) that they changed in place, as torch's own in-place optimizer ops do.  (A REPLAY of a
            # captured step cannot do this: consumers must not cache on `_version` alone — RNNTModel's decode tables do not.)
            torch.autograd.graph.increment_version(ps)
            if self.capturable:
                self._step_capturable(lib, gi, group, ps, gs, dev, total)
                continue
            ms, vs = [], []
            for p in ps:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if torch.is_tensor(st["step"]):  # state loaded from a torch.optim.AdamW checkpoint (util.py:7-12)
                    st["step"] = int(st["step"].item())
                st["step"] += 1
                ms.append(st["exp_avg"])
                vs.append(st["exp_avg_sq"])
            steps = {self.state[p]["step"] for p in ps}
            # parameters that joined later (or skipped steps without a gradient) carry their own
            # bias corrections: one engine call per distinct step count
            for step in sorted(steps):
                idx = [i for i, p in enumerate(ps) if self.state[p]["step"] == step]
                sel = lambda xs: [xs[i] for i in idx]
                b1, b2 = group["betas"]
                with torch.cuda.device(dev):
                    engine._check(lib.rnnt_engine_adamw_step(
                        len(idx), _ptr_array(sel(ps)), _ptr_array(sel(gs)), _ptr_array(sel(ms)),
                        _ptr_array(sel(vs)), _numel_array(sel(ps)), ctypes.c_double(group["lr"]),
                        ctypes.c_double(b1), ctypes.c_double(b2), ctypes.c_double(group["eps"]),
                        ctypes.c_double(group["weight_decay"]), ctypes.c_int64(int(step)),
                        engine._p(total) if total is not None else ctypes.c_void_p(0),
                        ctypes.c_float(self.max_grad_norm if total is not None else -1.0), 1,
                        engine._stream(dev)))
        return loss

    def _state_step(self, ps):
        """Step count the (loaded or running) state of `ps` carries: int, or tensor from torch's own AdamW."""
        step0 = 0
        for p in ps:
            st = self.state.get(p, {})
            if "step" in st:
                step0 = max(step0, int(st["step"].item()) if torch.is_tensor(st["step"]) else int(st["step"]))
        return step0

    def _seed_dev(self, gi, ps, dev):
        """The group's device step counter and scratch, created once (a captured graph points at them)
        from whatever count the state holds."""
        if gi not in self._dev:
            self._dev[gi] = (torch.full((1,), self._state_step(ps), dtype=torch.int64, device=dev),
                             torch.zeros(4, dtype=torch.float32, device=dev))
        return self._dev[gi]

    def load_state_dict(self, state_dict):
        """A resume (the reference's checkpoints, rnnt/util.py:7-24) after the first step: the device
        counter of every group continues from the LOADED count and the loaded learning rate lands in the
        device lr tensor — both updated IN PLACE, so a captured graph that points at them stays valid and
        the loaded per-parameter step values are not overwritten by the pre-load counter."""
        old_lr = [g["lr"] for g in self.param_groups]
        super().load_state_dict(state_dict)
        for gi, group in enumerate(self.param_groups):
            if torch.is_tensor(old_lr[gi]) and self.capturable:
                new = group["lr"]
                old_lr[gi].fill_(float(new.item()) if torch.is_tensor(new) else float(new))
                group["lr"] = old_lr[gi]
            if gi in self._dev:
                self._dev[gi][0].fill_(self._state_step(group["params"]))

    def _step_capturable(self, lib, gi, group, ps, gs, dev, total):
        if len(ps) != len(group["params"]):
            raise RuntimeError("rnnt_amd.optim.AdamW(capturable=True): every parameter of a group needs a gradient "
                               "each step (one device step counter per group)")
        step_dev, hyper = self._seed_dev(gi, ps, dev)
        lr = group["lr"]
        if not torch.is_tensor(lr):
            lr = group["lr"] = torch.tensor(float(lr), dtype=torch.float32, device=dev)
        if lr.device != dev or lr.dtype != torch.float32:
            raise RuntimeError("rnnt_amd.optim.AdamW(capturable=True): lr must be a float32 tensor on the parameters' device")
        ms, vs = [], []
        for p in ps:
            st = self.state[p]
            if "exp_avg" not in st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["step"] = step_dev  # shared device counter (state_dict() stores its value per parameter)
            ms.append(st["exp_avg"])
            vs.append(st["exp_avg_sq"])
        b1, b2 = group["betas"]
        with torch.cuda.device(dev):
            engine._check(lib.rnnt_engine_adamw_step_dev(
                len(ps), _ptr_array(ps), _ptr_array(gs), _ptr_array(ms), _ptr_array(vs), _numel_array(ps),
                engine._p(lr), ctypes.c_double(b1), ctypes.c_double(b2), ctypes.c_double(group["eps"]),
                ctypes.c_double(group["weight_decay"]), engine._p(step_dev), engine._p(hyper),
                engine._p(total) if total is not None else ctypes.c_void_p(0),
                ctypes.c_float(self.max_grad_norm if total is not None else -1.0), 1, engine._stream(dev)))
