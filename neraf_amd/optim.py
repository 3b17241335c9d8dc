"""Optimizer step of the joint training loop: ``FusedAdam`` is torch.optim.Adam (the optimizer nerfstudio's ``Optimizers``
builds for the reference's ``fields`` and ``audio_fields`` groups, NeRAF_config.py:116-127) with every parameter tensor of all
groups updated by ONE HIP launch (``neraf_fused_adam``, include/neraf_hip.h).  It plugs into ``torch.amp.GradScaler`` the way
torch's own fused Adam does: the scaler hands over its device-side ``grad_scale`` / ``found_inf`` and the kernel un-scales on the
fly and skips the update when a gradient was non-finite, so no pass ever writes un-scaled gradients.

fp32 parameters and gradients only; no weight decay, no amsgrad (what the reference configures)."""
from __future__ import annotations

import ctypes as C
from typing import List

import numpy as np
import torch

from . import _lib
from .field import _dev_index, _stream_ptr

# Bumped by every FusedAdam.step: the kernel updates parameters in place without touching tensor version counters, so caches of
# derived data (packed fp16 weights used in eval mode) key on this as well.
UPDATE_EPOCH = 0
_ARG_PTRS = 256          # csrc/optim.hip kArgPtrs: gradient pointers that fit the kernel-argument table

_REC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("numel", "<i8"), ("group", "<i4"), ("slot", "<i4")])
assert _REC.itemsize == 48
_DUAL = np.dtype([("m", "<u8"), ("v", "<u8"), ("group", "<i4"), ("slot", "<i4")])      # AdamDual, csrc/optim.hip
assert _DUAL.itemsize == 24


class FusedAdam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True        # GradScaler.step passes grad_scale / found_inf instead of synchronising

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if len(self.param_groups) > 8:
            raise ValueError("FusedAdam supports up to 8 parameter groups")
        b, e = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        if any(g["betas"] != b or g["eps"] != e for g in self.param_groups):
            raise ValueError("FusedAdam: betas and eps must be the same in every group (the learning rate may differ)")
        self._plans = {}                       # ids of the parameters holding a gradient -> device tables (the proposal networks
                                               # only receive gradients every few steps: two plans alternate)
        self._step_t = None                    # device float [n_slots][4]: per-TENSOR step counter + bias corrections (slot = position
                                               # of the tensor in param-group order), torch.optim.Adam's per-parameter `step`
        self._found = None
        self._fresh_plan = None
        self._defer_to = None                  # see fuse_shared_updates_into
        self._dual_from = None
        self._pending = None
        self._reindex()

    def fuse_shared_updates_into(self, later: "FusedAdam") -> None:
        """The reference steps the radiance-field parameters with "fields" and then again with "audio_fields" (NeRAF_pipeline.py:487):
        two full passes over p and g.  After ``first.fuse_shared_updates_into(second)`` the FIRST optimizer's ``step`` only advances
        the counters of the tensors both optimizers hold, and the SECOND optimizer's ``step`` applies both updates to them in one
        pass (``neraf_fused_adam_dual``: same fp32 operations in the same order, bit-identical parameters and moments).  Contract:
        the two are stepped alternately, first then second, every iteration -- what ``Optimizers.optimizer_scaler_step_all`` does; a
        second ``first.step()`` before ``second.step()`` raises (the deferred update's gradient would be gone)."""
        if later is self or not isinstance(later, FusedAdam):
            raise ValueError("fuse_shared_updates_into needs another FusedAdam")
        if self._dual_from is not None or self._defer_to is not None or later._dual_from is not None or later._defer_to is not None:
            # chains (a -> b -> c) are not supported: b would both apply a's deferred update and defer its own, and a tensor held by
            # all three would have a's pending update consumed by a launch that does not contain the tensor
            raise ValueError("fuse_shared_updates_into links PAIRS: one of the two optimizers is already part of a fused pair")
        if self.param_groups[0]["betas"] != later.param_groups[0]["betas"] or self.param_groups[0]["eps"] != later.param_groups[0]["eps"]:
            raise ValueError("fused double updates need equal betas / eps in both optimizers")
        mine = {id(p) for p, _ in self._flat}
        self._shared = {id(p) for p, _ in later._flat if id(p) in mine}
        if not self._shared:
            return
        self._defer_to, later._dual_from = later, self
        self._plans, later._plans = {}, {}
        self._fresh_plan = later._fresh_plan = None

    def flush_pending(self) -> bool:
        """Apply a deferred update of the shared tensors that the later optimizer never consumed (it was not stepped this iteration: an
        exception, a caller stepping this optimizer alone, a state load in between) with a plain launch over those tensors -- their
        counters were already advanced by this optimizer's ``step``, and their gradients are still the ones of that step as long as
        this is called before they are dropped (``zero_grad`` does).  Returns whether there was anything to flush."""
        if self._pending is None:
            return False
        lrs, fi = self._pending
        plan, gs, b1, b2 = self._pending_flush
        self._pending = None
        first = self._defer_to
        bt, bc = plan["sh_blk_tensor"], plan["sh_blk_chunk"]
        if int(bt.numel()) == 0:
            return False
        if any(p.grad is None for p, _ in self._flat if id(p) in self._shared):
            # the counters of these tensors advanced in step(), their update cannot be applied any more: say so
            import warnings
            warnings.warn("FusedAdam.flush_pending: the gradients of a deferred update are gone (set to None before the later "
                          "optimizer stepped or zero_grad ran): that update is dropped, its step counters have advanced")
            return False
        global UPDATE_EPOCH
        UPDATE_EPOCH += 1
        lib = _lib.load()
        gargs = plan.get("gargs") if plan.get("gdev") is None else None
        if gargs is not None:
            # the pointers recorded at step() time are stale if a .grad tensor was replaced since: rebuild them from the current ones
            cur = tuple(e[0].grad.data_ptr() for e in self._flat if e[0].grad is not None)
            if len(cur) == len(gargs) and cur != plan.get("gp"):
                gargs = (C.c_void_p * len(cur))(*cur)
        _lib.check(lib.neraf_fused_adam_dual(_lib.ctx(plan["dev"]), plan["table"].data_ptr(),
                                             plan["gdev"].data_ptr() if plan.get("gdev") is not None else None, bt.data_ptr(), bc.data_ptr(),
                                             int(bt.numel()), lrs, len(self.param_groups), 0, b1, b2, float(self.param_groups[0]["eps"]),
                                             self._step_t.data_ptr(), gs.data_ptr() if gs is not None else None,
                                             fi.data_ptr() if fi is not None else None, None, None, None, None, 0,
                                             gargs, len(gargs) if gargs is not None else 0, _stream_ptr()), plan["dev"])
        del first
        return True

    def zero_grad(self, set_to_none: bool = True):
        self.flush_pending()             # the gradients about to be dropped are the deferred update's
        return super().zero_grad(set_to_none=set_to_none)

    def state_dict(self):
        self.flush_pending()
        return super().state_dict()

    def _reindex(self):
        self._flat = [(p, gi) for gi, group in enumerate(self.param_groups) for p in group["params"]]
        self._all_sig = tuple(id(e[0]) for e in self._flat)
        self._slot = {id(p): i for i, (p, _) in enumerate(self._flat)}

    def group_steps(self, gi: int) -> torch.Tensor:
        """Step counters (device float [n]) of the tensors of parameter group ``gi``, in group order."""
        idx = [self._slot[id(p)] for p in self.param_groups[gi]["params"]]
        return self._steps(self.param_groups[gi]["params"][0].device)[idx, 0]

    def _steps(self, device) -> torch.Tensor:
        n = len(self._flat)
        if self._step_t is None or self._step_t.device != device or self._step_t.shape[0] < n:
            old = self._step_t
            self._step_t = torch.zeros((max(n, 8), 4), dtype=torch.float32, device=device)   # per tensor {t, 1/(1-b1^t), 1/sqrt(1-b2^t), -}
            if old is not None:
                self._step_t[:old.shape[0]].copy_(old)
                for q, _ in self._flat:                     # `step` entries are views of the table: re-point them
                    stq = self.state.get(q)
                    if stq and "step" in stq:
                        stq["step"] = self._step_t[self._slot[id(q)], 0]
                self._plans, self._fresh_plan = {}, None
        return self._step_t

    def _state_for(self, p: torch.Tensor, gi: int):
        st = self.state[p]
        if not st:
            st["step"] = self._steps(p.device)[self._slot[id(p)], 0]     # one device counter per tensor (see csrc/optim.hip)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _adopt_loaded_state(self):
        """After ``load_state_dict`` / unpickling: the loaded per-parameter ``step`` values become the device counters again (the
        kernel advances those, and every parameter's ``state['step']`` is a view of its slot), moments move to the parameter's
        device, and every cached launch plan -- which holds raw moment pointers -- is dropped."""
        if getattr(self, "_pending", None) is not None and getattr(self, "_step_t", None) is not None:
            self.flush_pending()                # a deferred update is applied, not discarded (the loaded state then replaces it, as torch's would)
        self._plans, self._fresh_plan = {}, None
        for other in (getattr(self, "_defer_to", None), getattr(self, "_dual_from", None)):
            if other is not None:               # the partner's plans hold raw pointers to this optimizer's moments / expect them
                other._plans, other._fresh_plan = {}, None
        self._pending = None
        self._step_t = None
        self._reindex()
        loaded = {}
        for p, gi in self._flat:
            st = self.state.get(p)
            if st:
                loaded[id(p)] = float(st["step"]) if "step" in st else 0.0
        if not loaded:
            return
        dev = next(p for p, _ in self._flat).device
        steps = self._steps(dev)
        b1, b2 = self.param_groups[0]["betas"]
        host = np.zeros((steps.shape[0], 4), np.float32)
        for p, gi in self._flat:
            t = loaded.get(id(p))
            if t is None:
                continue
            i = self._slot[id(p)]
            host[i, 0] = t
            if t > 0:
                host[i, 1] = 1.0 / (1.0 - b1 ** t)
                host[i, 2] = 1.0 / (1.0 - b2 ** t) ** 0.5
        steps.copy_(torch.from_numpy(host))
        for p, gi in self._flat:
            st = self.state.get(p)
            if not st:
                continue
            st["step"] = steps[self._slot[id(p)], 0]
            for k in ("exp_avg", "exp_avg_sq"):
                st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._adopt_loaded_state()

    def __setstate__(self, state):
        super().__setstate__(state)
        if not hasattr(self, "_plans"):
            self._step_t = self._found = None
        self._adopt_loaded_state()

    def _build(self, entries: List[tuple], device) -> dict:
        chunk = _lib.load().neraf_fused_adam_chunk()
        rec = np.zeros(len(entries), dtype=_REC)
        bt, bc = [], []
        for i, (p, gi) in enumerate(entries):
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise TypeError("FusedAdam: contiguous fp32 parameters and gradients only")
            if p.device != device:
                raise ValueError("FusedAdam: all parameters on one device")
            st = self._state_for(p, gi)
            rec[i] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), gi, self._slot[id(p)])
            n = (p.numel() + chunk - 1) // chunk
            bt.append(np.full(n, i, np.int32)); bc.append(np.arange(n, dtype=np.int32))
        n = len(entries)
        extra = {}
        if self._defer_to is not None:         # update map without the tensors the later optimizer updates for this one
            keep = [i for i, (p, _) in enumerate(entries) if id(p) not in self._shared]
            ubt = [bt[i] for i in keep] or [np.zeros(0, np.int32)]
            ubc = [bc[i] for i in keep] or [np.zeros(0, np.int32)]
            shared = [i for i, (p, _) in enumerate(entries) if id(p) in self._shared]
            sbt = [bt[i] for i in shared] or [np.zeros(0, np.int32)]
            sbc = [bc[i] for i in shared] or [np.zeros(0, np.int32)]
            extra.update(upd_blk_tensor=torch.from_numpy(np.concatenate(ubt)).to(device),
                         upd_blk_chunk=torch.from_numpy(np.concatenate(ubc)).to(device),
                         # the shared tensors alone: launched by flush_pending() when the later optimizer never stepped
                         sh_blk_tensor=torch.from_numpy(np.concatenate(sbt)).to(device),
                         sh_blk_chunk=torch.from_numpy(np.concatenate(sbc)).to(device),
                         deferred=bool(shared))
        if self._dual_from is not None:        # the earlier optimizer's moments / group / counter slot of the shared tensors
            first = self._dual_from
            dual = np.zeros(n, dtype=_DUAL)
            gi0 = {id(p): g for p, g in first._flat}
            for i, (p, _) in enumerate(entries):
                if id(p) in first._shared:
                    st0 = first._state_for(p, gi0[id(p)])
                    dual[i] = (st0["exp_avg"].data_ptr(), st0["exp_avg_sq"].data_ptr(), gi0[id(p)], first._slot[id(p)])
            extra["dual"] = torch.from_numpy(dual.view(np.uint8).copy()).to(device)
        return dict(n_tensors=n, table=torch.from_numpy(rec.view(np.uint8).copy()).to(device),
                    blk_tensor=torch.from_numpy(np.concatenate(bt)).to(device),
                    blk_chunk=torch.from_numpy(np.concatenate(bc)).to(device), **extra,
                    # gradient-pointer column, refreshed asynchronously every step: (pinned, device, event, used) x 4
                    ring=[[torch.empty(n, dtype=torch.int64).pin_memory(), torch.empty(n, dtype=torch.int64, device=device),
                           torch.cuda.Event(), False] for _ in range(4)], ring_i=0, gp=None, gdev=None)

    def _plan(self):
        """Device tables for the parameters that currently hold a gradient, with the gradient-pointer column refreshed (through
        pinned memory, without synchronising).  Cached while the gradient tensors are the same objects' storages."""
        if len(self._flat) != sum(len(g["params"]) for g in self.param_groups):       # add_param_group since construction
            self._reindex()
        entries, gp = [], []
        for e in self._flat:
            g = e[0].grad
            if g is not None:
                entries.append(e)
                gp.append(g.data_ptr())
        if not entries:
            return None
        sig = [id(e[0]) for e in entries] if len(entries) != len(self._flat) else self._all_sig
        p0 = entries[0][0]
        sig = tuple(sig)
        plan = self._plans.get(sig)
        if plan is None:                       # a new set of parameters with gradients: build its device table (synchronises, once)
            plan = self._plans[sig] = self._build(entries, p0.device)
        gp = tuple(gp)
        if len(gp) <= _ARG_PTRS:
            # the gradient pointers travel in the kernel arguments (include/neraf_hip.h, neraf_grads_nonfinite): no device column,
            # no host-to-device copy per step
            if plan.get("gp") != gp:
                plan["gp"], plan["gargs"] = gp, (C.c_void_p * len(gp))(*gp)
            plan["gdev"] = None
        elif plan.get("gp") != gp:             # gradient tensors are new every step: refresh the pointer column
            k = plan["ring_i"]
            plan["ring_i"] = (k + 1) % 4
            slot = plan["ring"][k]
            pinned, gdev, ev = slot[0], slot[1], slot[2]
            if slot[3] and not ev.query():
                ev.synchronize()               # four uses old: complete unless the host ran that far ahead
            pinned.numpy()[:] = gp
            gdev.copy_(pinned, non_blocking=True)
            ev.record()
            slot[3] = True
            plan["gp"], plan["gdev"] = gp, gdev
        plan["dev"] = _dev_index(p0)
        return plan

    def check_finite(self, _keep_plan_for_step: bool = False) -> torch.Tensor:
        """found_inf (device float, 0 or 1) over every gradient of this optimizer: GradScaler's check in one launch."""
        plan = self._plan()
        if _keep_plan_for_step:               # GradScaler.step calls step() right after: no second walk over the parameters
            self._fresh_plan = plan
        if self._found is None:
            dev = next(p for g in self.param_groups for p in g["params"]).device
            self._found = torch.zeros(1, dtype=torch.float32, device=dev)
            self._found_clean = True
        if plan is None:
            self._found_clean = True
            return self._found.zero_()
        lib = _lib.load()
        gargs = plan.get("gargs") if plan.get("gdev") is None else None
        # `_found_clean`: GradScaler.update's launch reset the flag after reading it (neraf_amp_update_scale, clear_flags): no
        # clearing launch here; any other consumer of the flag leaves it to be cleared by this call
        _lib.check(lib.neraf_grads_nonfinite(_lib.ctx(plan["dev"]), plan["table"].data_ptr(),
                                             plan["gdev"].data_ptr() if plan.get("gdev") is not None else None,
                                             plan["blk_tensor"].data_ptr(), plan["blk_chunk"].data_ptr(), int(plan["blk_tensor"].numel()),
                                             self._found.data_ptr(), gargs, len(gargs) if gargs is not None else 0,
                                             int(bool(getattr(self, "_found_clean", False))), _stream_ptr()), plan["dev"])
        self._found_clean = False
        return self._found

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        global UPDATE_EPOCH
        UPDATE_EPOCH += 1
        plan = self._fresh_plan if self._fresh_plan is not None else self._plan()
        self._fresh_plan = None
        if plan is None:
            return loss
        lib = _lib.load()
        dev = plan["dev"]
        lrs = (C.c_float * len(self.param_groups))(*[float(g["lr"]) for g in self.param_groups])
        b1, b2 = self.param_groups[0]["betas"]
        gs = getattr(self, "grad_scale", None)
        fi = getattr(self, "found_inf", None)
        bt, bc = plan["blk_tensor"], plan["blk_chunk"]
        if self._defer_to is not None and plan.get("deferred"):
            if self._pending is not None:
                raise RuntimeError("FusedAdam: this optimizer's shared tensors are updated by the optimizer it was fused into "
                                   "(fuse_shared_updates_into): step that one (or call zero_grad / flush_pending) before stepping "
                                   "this one again")
            bt, bc = plan["upd_blk_tensor"], plan["upd_blk_chunk"]
            # what the later optimizer needs to apply this update: the learning rates as of NOW and this step's non-finite flag
            # (+ what flush_pending needs should that optimizer never step: this plan, the scale, the betas)
            self._pending = (lrs, fi)
            self._pending_flush = (plan, gs, float(b1), float(b2))
        dual = step0 = fi0 = lrs0 = None
        n0 = 0
        first = self._dual_from
        if first is not None and first._pending is not None and "dual" in plan:
            lrs0, fi0 = first._pending
            first._pending = None
            dual, step0, n0 = plan["dual"], first._step_t, len(first.param_groups)
        gargs = plan.get("gargs") if plan.get("gdev") is None else None
        _lib.check(lib.neraf_fused_adam_dual(_lib.ctx(dev), plan["table"].data_ptr(),
                                             plan["gdev"].data_ptr() if plan.get("gdev") is not None else None, bt.data_ptr(), bc.data_ptr(),
                                             int(bt.numel()), lrs, len(self.param_groups), plan["n_tensors"], float(b1), float(b2),
                                             float(self.param_groups[0]["eps"]), self._step_t.data_ptr(),
                                             gs.data_ptr() if gs is not None else None, fi.data_ptr() if fi is not None else None,
                                             dual.data_ptr() if dual is not None else None,
                                             step0.data_ptr() if step0 is not None else None,
                                             fi0.data_ptr() if fi0 is not None else None, lrs0, n0,
                                             gargs, len(gargs) if gargs is not None else 0, _stream_ptr()), dev)
        return loss


class GradScaler(torch.amp.GradScaler):
    """torch.amp.GradScaler whose non-finite check of a ``FusedAdam``'s gradients is ONE launch over the optimizer's tensor table
    (torch's check is a foreach pass per 100-odd tensors, after a Python loop over every parameter).  Everything else -- scale
    growth / back-off, ``step`` / ``update`` -- is inherited; other optimizers take the inherited path."""

    def _check_inf_per_device(self, optimizer):
        if not isinstance(optimizer, FusedAdam):
            return super()._check_inf_per_device(optimizer)
        found = optimizer.check_finite(_keep_plan_for_step=True)
        if not hasattr(self, "_fused_optimizers"):
            self._fused_optimizers = {}
        self._fused_optimizers[id(optimizer)] = optimizer
        state = self._per_optimizer_states[id(optimizer)]
        state["found_inf_per_device"] = {found.device: found}
        return state["found_inf_per_device"]

    def step(self, optimizer, *args, **kwargs):
        """``torch.amp.GradScaler.step`` for a ``FusedAdam``: the inherited path builds ``found_inf`` with ``sum([...])`` and the scale
        with ``scaler * 1`` -- two scalar launches per optimizer; here the optimizer receives the check's flag and the scale tensor
        themselves.  States, errors and every other optimizer follow the inherited code."""
        from torch.amp.grad_scaler import OptState
        if not self._enabled or not isinstance(optimizer, FusedAdam) or args or kwargs:
            return super().step(optimizer, *args, **kwargs)
        self._check_scale_growth_tracker("step")
        state = self._per_optimizer_states[id(optimizer)]
        if state["stage"] is OptState.STEPPED:
            raise RuntimeError("step() has already been called since the last update().")
        if state["stage"] is OptState.READY:
            self._check_inf_per_device(optimizer)
            optimizer.grad_scale = self._scale                     # gradients are still scaled: the kernel un-scales them
        else:
            optimizer.grad_scale = None                            # unscale_() already divided them
        founds = list(state["found_inf_per_device"].values())
        if len(founds) != 1:
            raise RuntimeError("FusedAdam holds the parameters of one device")
        optimizer.found_inf = founds[0]
        try:
            retval = optimizer.step()
        finally:
            del optimizer.grad_scale
            del optimizer.found_inf
        state["stage"] = OptState.STEPPED
        return retval

    def update(self, new_scale=None):
        """``GradScaler.update`` in one launch when every recorded flag lives on the scale's device (the inherited code adds the
        flags pairwise, then calls ``torch._amp_update_scale_``)."""
        if not self._enabled:
            return
        if new_scale is None and self._scale is not None and self._scale.is_cuda:
            founds = [f for st in self._per_optimizer_states.values() for f in st["found_inf_per_device"].values()]
            if 1 <= len(founds) <= 8 and all(f.device == self._scale.device and f.dtype == torch.float32 for f in founds):
                from collections import defaultdict
                from torch.amp.grad_scaler import _refresh_per_optimizer_state
                _scale, _growth_tracker = self._check_scale_growth_tracker("update")
                dev = _dev_index(_scale)
                # flags that belong to FusedAdam optimizers are reset by the same launch: their next check needs no clearing launch
                owners = [o for o in getattr(self, "_fused_optimizers", {}).values() if any(o._found is f for f in founds)]
                # a flag with a deferred consumer still outstanding (flush_pending replays the update against it) is left standing:
                # cleared here, a step that was skipped for non-finite gradients would be replayed as finite
                clear = len(owners) == len(founds) and all(o._pending is None for o in owners)
                _lib.check(_lib.load().neraf_amp_update_scale(_lib.ctx(dev), _scale.data_ptr(), _growth_tracker.data_ptr(),
                                                              _lib.ptr_array(founds), len(founds), float(self._growth_factor),
                                                              float(self._backoff_factor), int(self._growth_interval), int(clear),
                                                              _stream_ptr()), dev)
                if clear:
                    for o in owners:
                        o._found_clean = True
                self._per_optimizer_states = defaultdict(_refresh_per_optimizer_state)
                return
        return super().update(new_scale)
