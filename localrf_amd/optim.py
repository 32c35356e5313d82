"""FusedAdam -- torch.optim.Adam as the reference configures it (local_tensorfs.py:88-97,146,245:
betas (0.9, 0.99), eps 1e-8, no weight decay, no amsgrad), stepped by ONE HIP launch
(lrf_adam_step) for all tensors of all parameter groups -- and, through `step_many`, for the
tensors of many optimiser objects at once: the reference steps one tiny Adam per frame for the
rotation, translation and exposure parameters (local_tensorfs.py:229-243), dozens of launches
per iteration that batch into the same table here.

Same constructor arguments, param_groups (lr decay by `group["lr"] *= f` works as in
local_tensorfs.py:224-247) and per-parameter state keys (`step`, `exp_avg`, `exp_avg_sq`) as
torch.optim.Adam, so state dicts interchange.  No CPU fallback: parameters must live on the GPU.
Cited lines are relative to /root/reference/localTensoRF."""
import ctypes as C
import math

import torch
from torch.autograd.graph import increment_version

from . import _native as N


def _pack_target(fields):
    """The one field whose layout cache a launch over these optimisers' tensors can rewrite (FusedAdam.pack_field), with what
    lrf_adam_step_pack needs, or None: no such field, more than one, or no cache of the current grid's size yet."""
    fs = {id(f): f for f in fields if f is not None}
    if len(fs) != 1:
        return None
    (field,) = fs.values()
    tgt = field._fused_step_target()
    return None if tgt is None else (field, tgt)


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, pack_field=None):
        """pack_field: the localrf_amd TensorVMSplit whose parameters this optimiser steps -- its layout cache is then rewritten
        by the step itself (lrf_adam_step_pack: the plane / line tensors are stepped and written channel-last by one kernel)
        instead of by the next forward; an attribute, may be set or changed later (append_rf moves on to a new field)."""
        self.pack_field = pack_field
        if weight_decay != 0 or amsgrad:
            raise ValueError("FusedAdam implements the reference's configuration: weight_decay=0, amsgrad=False")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False))

    def zero_grad(self, set_to_none=True):
        """Same effect as Optimizer.zero_grad without its per-call profiler / compile wrappers: the
        scene calls this on dozens of one-tensor optimisers per iteration (local_tensorfs.py:224-243)."""
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_()
                        p.grad.zero_()

    # ------------------------------------------------------------------ table building
    def _entries(self):
        """[(param, grad, exp_avg, exp_avg_sq, step_size, bc2_sqrt, betas, eps)] for this step;
        advances the per-parameter step counters (torch/optim/adam.py::_init_group)."""
        out = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                if not p.is_cuda:
                    raise N.NativeError("localrf_amd: FusedAdam steps parameters on an AMD GPU only "
                                        f"(got {p.device}); there is no CPU fallback.")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise N.NativeError("localrf_amd: FusedAdam needs contiguous fp32 parameters")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0          # plain number (torch.optim.Adam.__setstate__ accepts either form)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                step = int(st["step"]) + 1      # int() also reads the tensor form torch.optim.Adam stores
                st["step"] = step
                bc1 = 1.0 - b1 ** step
                bc2_sqrt = math.sqrt(1.0 - b2 ** step)
                g = p.grad if p.grad.is_contiguous() and p.grad.dtype == torch.float32 else p.grad.contiguous().float()
                out.append((p, g, st["exp_avg"], st["exp_avg_sq"], group["lr"] / bc1, bc2_sqrt, (b1, b2), group["eps"], self.pack_field))
        return out

    @staticmethod
    def _launch(entries):
        """One lrf_adam_step per (betas, eps, device) class, LRF_ADAM_MAX tensors at a time."""
        lib = N.lib()
        classes = {}
        for e in entries:
            classes.setdefault((e[6], e[7], e[0].device), []).append(e)
        for ((b1, b2), eps, dev), es in classes.items():
            st = torch.cuda.current_stream(dev).cuda_stream
            for lo in range(0, len(es), N.LRF_ADAM_MAX):
                part = es[lo:lo + N.LRF_ADAM_MAX]
                tab = (N.LrfAdamTensor * len(part))()
                for t, (p, g, m, v, step_size, bc2_sqrt, _, _, _) in zip(tab, part):
                    t.p, t.g, t.m, t.v = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                    t.n, t.step_size, t.bc2_sqrt = p.numel(), step_size, bc2_sqrt
                fused = _pack_target(e[8] for e in part) if len(es) <= N.LRF_ADAM_MAX else None
                if fused is not None:                         # the step writes the field's layout cache too
                    field, (cp, keep, cache) = fused
                    N.check(lib.lrf_adam_step_pack(tab, len(part), None, b1, b2, eps, C.byref(cp), cache.data_ptr(), st), "lrf_adam_step_pack")
                    increment_version([e[0] for e in part])
                    field._mark_cache_fresh()
                    continue
                N.check(lib.lrf_adam_step(tab, len(part), b1, b2, eps, st), "lrf_adam_step")
                # the kernel rewrote the parameters behind autograd's back: bump their versions so
                # layout caches keyed on (data_ptr, _version) (TensorVMSplit._ensure_cache) and
                # autograd's saved-tensor checks see the change
                increment_version([e[0] for e in part])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._launch(self._entries())
        return loss

    @staticmethod
    @torch.no_grad()
    def step_many(optimizers):
        """Step several FusedAdam objects with one launch (per hyper-parameter class)."""
        entries = []
        for opt in optimizers:
            if not isinstance(opt, FusedAdam):
                raise TypeError("step_many takes FusedAdam optimisers")
            entries.extend(opt._entries())
        FusedAdam._launch(entries)


class StaticAdamPlan:
    """The Adam launches of one training iteration in capturable form: a fixed list of (optimiser, parameter) pairs whose
    per-tensor step_size / bc2_sqrt are read by the kernel from a DEVICE table at execution time (lrf_adam_step_dev), so
    that the launches can be captured once in a hipGraph and replayed with the learning rates, bias corrections and
    "this view was not sampled: skip" flags the host writes before each replay.  Same arithmetic and per-parameter
    state (`step`, `exp_avg`, `exp_avg_sq`) as FusedAdam.step / torch.optim.Adam; state dicts stay interchangeable."""

    def __init__(self, pairs):
        self.pairs = list(pairs)                                  # [(FusedAdam, parameter)], launch order
        for opt, p in self.pairs:
            if not isinstance(opt, FusedAdam):
                raise TypeError("StaticAdamPlan takes FusedAdam optimisers")
            st = opt.state[p]
            if len(st) == 0:                                      # as FusedAdam._entries / torch.optim.Adam._init_group
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        self._group = {}
        for opt, _ in self.pairs:
            for grp in opt.param_groups:
                for q in grp["params"]:
                    self._group[(id(opt), id(q))] = grp

    def __len__(self):
        return len(self.pairs)

    def _classes(self):
        classes = {}
        for i, (opt, p) in enumerate(self.pairs):
            grp = self._group[(id(opt), id(p))]
            classes.setdefault((tuple(grp["betas"]), grp["eps"], p.device), []).append(i)
        return classes

    def packs(self, field):
        """Whether launch() would step this field through lrf_adam_step_pack right now, i.e. leave its layout cache holding
        the stepped values (the captured iteration then needs no refresh node in front of its forward)."""
        for idx in self._classes().values():
            if len(idx) <= N.LRF_ADAM_MAX:
                t = _pack_target(self.pairs[i][0].pack_field for i in idx)
                if t is not None and t[0] is field:
                    return True
        return False

    def launch(self, scalars_dev):
        """Enqueue (or capture) the launches.  scalars_dev: float32 [len(self), 2] on the device.  Every parameter must
        hold its gradient (.grad) at this point: the pointers are baked into the launch."""
        lib = N.lib()
        classes = {}
        for i, (opt, p) in enumerate(self.pairs):
            grp = self._group[(id(opt), id(p))]
            if p.grad is None:
                raise RuntimeError("StaticAdamPlan.launch: a parameter of the plan has no gradient")
            if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or not p.is_contiguous() or p.dtype != torch.float32:
                raise N.NativeError("localrf_amd: StaticAdamPlan needs contiguous fp32 parameters and gradients")
            classes.setdefault((tuple(grp["betas"]), grp["eps"], p.device), []).append(i)
        order = [i for idx in classes.values() for i in idx]
        if order != list(range(len(self.pairs))):
            raise ValueError("StaticAdamPlan: pairs must be grouped by (betas, eps, device)")
        for ((b1, b2), eps, dev), idx in classes.items():
            st = torch.cuda.current_stream(dev).cuda_stream
            for lo in range(0, len(idx), N.LRF_ADAM_MAX):
                part = idx[lo:lo + N.LRF_ADAM_MAX]
                tab = (N.LrfAdamTensor * len(part))()
                for t, i in zip(tab, part):
                    opt, p = self.pairs[i]
                    s = opt.state[p]
                    t.p, t.g, t.m, t.v = p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr()
                    t.n, t.step_size, t.bc2_sqrt = p.numel(), 0.0, 0.0
                fused = _pack_target(self.pairs[i][0].pack_field for i in part) if len(idx) <= N.LRF_ADAM_MAX else None
                if fused is not None:                         # the step leaves the field's layout cache holding the stepped values
                    field, (cp, keep, cache) = fused
                    N.check(lib.lrf_adam_step_pack(tab, len(part), scalars_dev[part[0]:].data_ptr(), b1, b2, eps, C.byref(cp), cache.data_ptr(), st),
                            "lrf_adam_step_pack")
                    self._packed_field = field
                    continue
                N.check(lib.lrf_adam_step_dev(tab, len(part), scalars_dev[part[0]:].data_ptr(), b1, b2, eps, st), "lrf_adam_step_dev")

    def host_scalars(self, out, active=None):
        """Advance the step counters of the parameters stepped this iteration and write their (step_size, bc2_sqrt) rows into
        `out` (a float32 [len(self), 2] numpy view of pinned memory); rows of parameters outside `active` (ids; None = all)
        are zero: the kernel leaves those tensors alone, as torch.optim.Adam leaves a parameter whose .grad is None."""
        for i, (opt, p) in enumerate(self.pairs):
            if active is not None and id(p) not in active:
                out[i, 0] = 0.0
                out[i, 1] = 0.0
                continue
            grp = self._group[(id(opt), id(p))]
            b1, b2 = grp["betas"]
            st = opt.state[p]
            step = int(st["step"]) + 1
            st["step"] = step
            out[i, 0] = grp["lr"] / (1.0 - b1 ** step)
            out[i, 1] = math.sqrt(1.0 - b2 ** step)

    def bump_versions(self):
        """The replayed kernels rewrote the parameters behind autograd's back (see FusedAdam._launch) -- and, when the launch was
        lrf_adam_step_pack, the field's layout cache with them."""
        increment_version([p for _, p in self.pairs])
        f = getattr(self, "_packed_field", None)
        if f is not None:
            f._mark_cache_fresh()
