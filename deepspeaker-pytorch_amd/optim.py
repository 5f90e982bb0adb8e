"""Fused multi-tensor optimizers.

Drop-in for the three optimizers `create_optimizer` builds in the reference (train_triplet.py:369-383):
`optim.Adagrad(lr, lr_decay, weight_decay)`, `optim.SGD(lr, momentum=0.9, dampening=0.9, weight_decay)`,
`optim.Adam(lr, weight_decay)`.  Same constructor arguments, same per-parameter state keys (`step`, `sum`,
`momentum_buffer`, `exp_avg`, `exp_avg_sq`), so `optimizer.state_dict()` checkpoints
(train_triplet.py:325-327) interchange with torch's.  One HIP launch per `step()` for all tensors.
"""
from __future__ import annotations

import ctypes
import math
from typing import List

import torch

from .engine import trace_range
from .model import _require_cuda, get_engine


class _FusedBase(torch.optim.Optimizer):
    _engine = None          # tests may inject an Engine bound to the host emulator
    # Gradient-overflow flag of the loss-scaled fp16 training step: an int32 device tensor [1]; while it is non-zero a
    # step() leaves parameters and state untouched (decided on the device, no host round trip), and step() CONSUMES it --
    # the backward passes only ever raise it (a step may be several passes), the optimizer clears it after its last launch.
    # Either a tensor assigned to `skip_flag` directly, or -- what create_optimizer wires -- the model itself
    # (`skip_source`, a weak reference): the flag is then resolved at every step() on the device the parameters are on
    # NOW (a model moved with .cuda() after the optimizer was built gets a flag there, never a stale or host pointer), and
    # the model latches the consumed value for `grad_overflow` / `update_loss_scale()`.
    # The host-side step counters (Adagrad's decayed lr, Adam's bias correction) still advance on a skipped step -- unlike
    # torch.amp.GradScaler, which does not call optimizer.step() at all: undoing them would need the flag on the host.
    _skip_tensor = None
    skip_source = None
    # Device-side step count (enable_device_step): Adagrad's decayed learning rate and Adam's bias corrections are computed
    # inside the kernels from an int32 counter on the device that every step() bumps there -- a step captured into a HIP
    # graph (train_graph.GraphedTripletStep) then replays with the count of the replay, not of the capture.
    _dev_step = None
    _dev_step_on = False

    def enable_device_step(self):
        """From now on the step count that enters the update lives on the device (initialised from the host-side counts;
        all parameters must share one count).  Host-side `state[p]["step"]` keeps advancing in eager steps; after graph
        replays `sync_host_steps()` brings it up to date (one host synchronisation).  A step skipped by the overflow flag
        does not count (as with torch.amp.GradScaler, which does not call step() at all)."""
        self._dev_step_on = True                # (the counter itself is created at the first step, where the parameters are)
        return self

    def _device_step(self, params):
        """the counter, on the device the parameters are on NOW (created at first use; follows a model that was moved)"""
        dev = params[0].device
        if self._dev_step is None:
            allp = [p for g in self.param_groups for p in g["params"]]
            counts = {float(self.state[p]["step"]) for p in allp if "step" in self.state.get(p, {})}
            if len(counts) > 1:
                raise RuntimeError(f"device-side step count needs one count for all parameters, found {sorted(counts)}")
            start = int(counts.pop()) if counts else 0
            if start == 0 and any("momentum_buffer" in self.state.get(p, {}) for p in allp):
                start = 1                       # SGD keeps no step count: buffers that exist are past their first step
            self._dev_step = torch.full((1,), start, dtype=torch.int32, device=dev)
        elif self._dev_step.device != dev:
            self._dev_step = self._dev_step.to(dev)
        return self._dev_step

    def sync_host_steps(self):
        if self._dev_step is None:
            return
        n = float(self._dev_step.item())
        for g in self.param_groups:
            for p in g["params"]:
                st = self.state.get(p)
                if st is not None and "step" in st:
                    st["step"].fill_(n)

    def _use_dev_step(self, parts) -> bool:
        """The device-side count serves ONE step count for all parameters.  Parameters that joined later (the classifier
        head when the regime switches: its own count / a fresh momentum buffer) make a step fall back to the host-side
        counts -- except inside a stream capture, where host-side counts would be frozen into the graph."""
        if not self._dev_step_on:
            return False
        if len(parts) <= 1 and len(self.param_groups) == 1:
            if self._dev_step is None and parts:        # created from the host-side counts BEFORE this step bumps them
                self._device_step(next(iter(parts.values())))
            return True
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("device-side step count: parameters with different step counts (or several parameter groups) "
                               "cannot share the one counter a captured step needs")
        return False

    def _bump_dev_step(self, eng, params):
        eng.lib.call("ds_optim_step_inc", eng._p(self._device_step(params)), self._skip(eng, params), eng._stream(params[0]))

    @property
    def skip_flag(self):
        model = self.skip_source() if self.skip_source is not None else None
        if model is not None:
            for group in self.param_groups:
                for p in group["params"]:
                    return model.grad_overflow_flag(p.device)
        return self._skip_tensor

    @skip_flag.setter
    def skip_flag(self, flag):
        self._skip_tensor, self.skip_source = flag, None

    def _skip(self, eng, params):
        flag = self.skip_flag
        if flag is None:
            return None
        if flag.device != params[0].device:
            raise RuntimeError(f"optimizer skip_flag lives on {flag.device}, the parameters on {params[0].device}")
        return eng._p(flag)

    def _consume_skip(self):
        """after the last launch of a step(): the flag has been read by every launch; latch + clear it"""
        model = self.skip_source() if self.skip_source is not None else None
        if model is not None:
            model._consume_overflow()
        elif self._skip_tensor is not None:
            self._skip_tensor.zero_()

    def _eng(self):
        return self._engine if self._engine is not None else get_engine()

    @staticmethod
    def _bump_versions(params):
        """The kernels write parameters through raw pointers; autograd's version counters are what
        DeepSpeakerModel's packed-filter / folded-BatchNorm caches (and autograd's own saved-tensor checks) watch."""
        for p in params:
            torch.autograd.graph.increment_version(p)

    def _partitions(self, group, fresh_key=None):
        """Parameters with gradients, split into sets that share the per-step scalars: the step count (Adagrad's
        decayed lr, Adam's bias correction) or, for SGD, whether the momentum buffer exists yet -- a parameter
        that starts receiving gradients later (the classifier head when the regime switches) must not inherit
        the others' count, nor reset their buffers."""
        parts = {}
        for p in group["params"]:
            if p.grad is None:
                continue
            st = self.state[p]
            k = (fresh_key not in st) if fresh_key is not None else float(st["step"]) if "step" in st else 0.0
            parts.setdefault(k, []).append(p)
        return parts

    def _tables(self, group, state_keys: List[str], need_state2: bool, with_step: bool = True, params=None):
        """Device pointer tables + chunk table for `params` (default: the parameters of `group` that have
        gradients), cached until the set of tensors / their storage changes."""
        if params is None:
            params = [p for p in group["params"] if p.grad is not None]
        if not params:
            return None
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise ValueError("fused optimizers need contiguous float32 parameters and gradients")
        states = []
        for p in params:
            st = self.state[p]
            if with_step and "step" not in st:
                st["step"] = torch.tensor(0.0)
            for k in state_keys:
                if k not in st:
                    st[k] = torch.zeros_like(p, memory_format=torch.preserve_format)
            states.append(st)
        # Two caches.  STATIC: pointer tables of the parameters and their state, element counts, the chunk table -- rebuilt
        # only when the set of tensors / their storage changes.  PER STEP: the gradients' pointers (fresh tensors after every
        # zero_grad(set_to_none=True); the caching allocator usually hands the same blocks back, then nothing is written).
        # Tables are written by ds_fill_bytes -- values as kernel arguments, no pinned staging buffer: nothing blocks the
        # host, and the step can be captured into a HIP graph (a pinned asynchronous copy leaves an event behind that
        # torch's host allocator later queries: illegal for an event recorded in a capturing stream).
        eng = self._eng()
        dev = params[0].device
        skey = tuple((p.data_ptr(),) + tuple(st[k].data_ptr() for k in state_keys) for p, st in zip(params, states))
        caches = group.setdefault("_ds_cache", {})
        cache = caches.get(len(params))

        def fill(values, dtype, out=None):
            import numpy as np
            host = np.ascontiguousarray(np.asarray(values, dtype=np.int64 if dtype == torch.int64 else np.int32))
            if out is None:
                out = torch.empty(host.size, dtype=dtype, device=dev)
            if dev.type != "cuda" and not getattr(eng.lib, "host_memory", False):
                raise RuntimeError("fused optimizers need parameters on a ROCm device")
            raw, nbytes, st_ = host.ctypes.data, host.nbytes, eng._stream(out)
            for off in range(0, nbytes, 2048):
                eng.lib.call("ds_fill_bytes", ctypes.c_void_p(out.data_ptr() + off), ctypes.c_void_p(raw + off),
                             min(2048, nbytes - off), st_)
            return out

        if cache is None or cache["key"] != skey:
            chunk = eng.lib.raw("ds_optim_chunk_elems")()
            ct, ci = [], []
            for i, p in enumerate(params):
                for c in range((p.numel() + chunk - 1) // chunk):
                    ct.append(i)
                    ci.append(c)
            cache = {
                "key": skey,
                "params": fill([p.data_ptr() for p in params], torch.int64),
                "s1": fill([st[state_keys[0]].data_ptr() for st in states], torch.int64) if state_keys else None,
                "s2": fill([st[state_keys[1]].data_ptr() for st in states], torch.int64) if need_state2 else None,
                "numel": fill([p.numel() for p in params], torch.int64),
                "ct": fill(ct, torch.int32), "ci": fill(ci, torch.int32),
                "n_chunks": len(ct), "grads": None, "gkey": None,
            }
            caches[len(params)] = cache
        gkey = tuple(p.grad.data_ptr() for p in params)
        if cache["gkey"] != gkey or (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            # (inside a capture the write is always enqueued: the graph must own a node that sets the table it replays with)
            cache["grads"] = fill(list(gkey), torch.int64, out=cache["grads"])
            cache["gkey"] = gkey
        return params, states, cache

    @staticmethod
    def _args(eng, c):
        return (eng._p(c["params"]), eng._p(c["grads"]), eng._p(c["s1"]), eng._p(c["s2"]), eng._p(c["numel"]),
                eng._p(c["ct"]), eng._p(c["ci"]), c["n_chunks"])

    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:
            g.pop("_ds_cache", None)
        return sd


class FusedAdagrad(_FusedBase):
    """torch.optim.Adagrad(params, lr, lr_decay, weight_decay, initial_accumulator_value=0, eps=1e-10)."""

    def __init__(self, params, lr=1e-2, lr_decay=0.0, weight_decay=0.0, eps=1e-10):
        super().__init__(params, dict(lr=lr, lr_decay=lr_decay, weight_decay=weight_decay, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        eng = self._eng()
        with trace_range("ds.optimizer.step"):
            self._step_groups(eng)
        self._consume_skip()
        return loss

    def _step_groups(self, eng):
        for group in self.param_groups:
            parts = self._partitions(group)
            dev_step = self._use_dev_step(parts)
            for part in parts.values():
                params, states, c = self._tables(group, ["sum"], False, params=part)
                for st in states:
                    st["step"] += 1
                if dev_step:
                    self._bump_dev_step(eng, params)
                    eng.lib.call("ds_adagrad_step_dev_f32", *self._args(eng, c), float(group["lr"]), float(group["lr_decay"]),
                                 group["weight_decay"], group["eps"], eng._p(self._dev_step), self._skip(eng, params),
                                 eng._stream(params[0]))
                else:
                    step = float(states[0]["step"])
                    clr = group["lr"] / (1 + (step - 1) * group["lr_decay"])
                    eng.lib.call("ds_adagrad_step_f32", *self._args(eng, c), clr, group["weight_decay"], group["eps"],
                                 self._skip(eng, params), eng._stream(params[0]))
                self._bump_versions(params)


class FusedSGD(_FusedBase):
    """torch.optim.SGD(params, lr, momentum, dampening, weight_decay) (no nesterov)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        eng = self._eng()
        with trace_range("ds.optimizer.step"):
            self._step_groups(eng)
        self._consume_skip()
        return loss

    def _step_groups(self, eng):
        for group in self.param_groups:
            parts = self._partitions(group, fresh_key="momentum_buffer")
            dev_step = self._use_dev_step(parts)
            for fresh, part in parts.items():
                params, states, c = self._tables(group, ["momentum_buffer"] if group["momentum"] != 0 else [], False,
                                                 with_step=False, params=part)
                if dev_step:                            # "first step" = the step that counts 1 on the device
                    self._bump_dev_step(eng, params)
                    eng.lib.call("ds_sgd_step_dev_f32", *self._args(eng, c), group["lr"], group["momentum"], group["dampening"],
                                 group["weight_decay"], eng._p(self._dev_step), self._skip(eng, params), eng._stream(params[0]))
                else:
                    eng.lib.call("ds_sgd_step_f32", *self._args(eng, c), group["lr"], group["momentum"], group["dampening"],
                                 group["weight_decay"], int(fresh), self._skip(eng, params), eng._stream(params[0]))
                self._bump_versions(params)


class FusedAdam(_FusedBase):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) (no amsgrad)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        eng = self._eng()
        with trace_range("ds.optimizer.step"):
            self._step_groups(eng)
        self._consume_skip()
        return loss

    def _step_groups(self, eng):
        for group in self.param_groups:
            parts = self._partitions(group)
            dev_step = self._use_dev_step(parts)
            for part in parts.values():
                params, states, c = self._tables(group, ["exp_avg", "exp_avg_sq"], True, params=part)
                for st in states:
                    st["step"] += 1
                step = float(states[0]["step"])
                b1, b2 = group["betas"]
                if dev_step:
                    self._bump_dev_step(eng, params)
                    eng.lib.call("ds_adam_step_dev_f32", *self._args(eng, c), group["lr"], float(b1), float(b2), group["eps"],
                                 group["weight_decay"], eng._p(self._dev_step), self._skip(eng, params), eng._stream(params[0]))
                else:
                    eng.lib.call("ds_adam_step_f32", *self._args(eng, c), group["lr"], b1, b2, group["eps"],
                                 group["weight_decay"], 1 - b1 ** step, math.sqrt(1 - b2 ** step), self._skip(eng, params),
                                 eng._stream(params[0]))
                self._bump_versions(params)


def create_optimizer(model, new_lr, optimizer="adagrad", lr_decay=1e-4, wd=0.0):
    """train_triplet.py:369-383 with the fused implementations (same hyper-parameters).  A model whose training step is the
    loss-scaled fp16 one (train_precision="f16") hands the optimizer its gradient-overflow flag: a step whose scaled
    gradients left fp16's range updates nothing."""
    if optimizer == "sgd":
        opt = FusedSGD(model.parameters(), lr=new_lr, momentum=0.9, dampening=0.9, weight_decay=wd)
    elif optimizer == "adam":
        opt = FusedAdam(model.parameters(), lr=new_lr, weight_decay=wd)
    elif optimizer == "adagrad":
        opt = FusedAdagrad(model.parameters(), lr=new_lr, lr_decay=lr_decay, weight_decay=wd)
    else:
        raise ValueError(optimizer)
    if getattr(model, "train_precision", None) == "f16" and hasattr(model, "grad_overflow_flag"):
        import weakref
        opt.skip_source = weakref.ref(model)
        # ... and its step count lives on the device, so that a step the flag skips does not advance Adagrad's decayed
        # learning rate / Adam's bias corrections either (the counter is created on the parameters' device at
        # the first step)
        opt.enable_device_step()
    return opt
