"""``torch.optim.Adam`` with the update of all parameters in one HIP launch.

The reference builds ``torch.optim.Adam(g_params, lr=..., betas=(beta1, beta2), eps=...)`` (cpc/train.py:335-337) and
calls ``optimizer.step()`` once per batch (:88-89).  This subclass keeps that interface and the state layout
(``state[p] = {"step", "exp_avg", "exp_avg_sq"}``), so ``state_dict()`` / ``load_state_dict()`` interoperate with
torch's own Adam and with the reference's checkpoints (``"optimizer"`` entry, train.py:139, :343-346); only ``step()``
differs: when every parameter of a group is a dense fp32 tensor on the GPU and the group uses none of weight decay /
amsgrad / maximize (the reference uses none), the update runs through ``cpc_adam_step`` (csrc/adam.hip).  Any other
group is handled by torch's own implementation.
"""
import ctypes
import math

import torch

from . import _lib


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        self._device_step = None          # graph-capturable mode (device_step_counter()): the step count lives on the GPU
        self._fast = None                 # step(): pointer tables of the last one-launch update, reused while nothing changed
        self._dev_tables = None

    def add_param_group(self, param_group):
        if getattr(self, "_fast", None) is not None:
            self._flush_fast()
        self._dev_tables = None
        return super().add_param_group(param_group)

    def device_step_counter(self, enable=True):
        """Graph-capturable mode: ``step()`` makes no per-step host decision (no ``.item()``, no host-side bias correction),
        the step count lives in one device double advanced by the kernel (cpc_adam_step_capturable), so a whole train step
        can be captured once and replayed as a HIP graph (train.Trainer(graph=True)).  ``state[p]["step"]`` is brought up to
        date by ``sync_step_state()`` (state_dict() calls it).  Requires every group to satisfy ``_hip_ok`` and one shared
        step count, which is how the reference's single-group optimiser behaves."""
        self._flush_fast()
        if not enable:
            if self._device_step is not None:
                self.sync_step_state()
            self._device_step = None
            return self
        if self._device_step is not None:
            # already on: the host-side copies of the count are stale (they are only refreshed on demand) -- bring them up to
            # date before the count is re-seeded from them (a re-capture after a learning-rate change comes through here)
            self.sync_step_state()
        if len(self.param_groups) != 1:
            # one device counter, advanced once per launch: with G groups it would advance G times per step()
            raise ValueError("device_step_counter() needs a single parameter group (the reference's optimiser has one)")
        params = [p for g in self.param_groups for p in g["params"]]
        counts = {int(self.state[p]["step"].item()) for p in params if len(self.state[p])}
        if len(counts) > 1:
            raise ValueError("device_step_counter() needs one shared step count")
        dev = params[0].device
        self._device_step = torch.tensor([float(counts.pop()) if counts else 0.0], dtype=torch.float64, device=dev)
        self._device_coef = torch.zeros(8, dtype=torch.float32, device=dev)
        for p in params:                                   # state tensors must exist before a capture
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return self

    def _flush_fast(self):
        """The fast path of step() counts in a Python int; bring ``state[p]["step"]`` up to date and leave the fast path."""
        fast, self._fast = self._fast, None
        if fast is not None:
            for p in fast["params"]:
                self.state[p]["step"] = torch.tensor(float(fast["k"]), dtype=torch.float32)

    def sync_step_state(self):
        self._flush_fast()
        if self._device_step is not None:
            k = float(self._device_step.item())
            for g in self.param_groups:
                for p in g["params"]:
                    if len(self.state[p]):
                        self.state[p]["step"] = torch.tensor(k, dtype=torch.float32)

    def state_dict(self):
        self.sync_step_state()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """In graph-capturable mode a captured step points at the moment tensors and at the device step count: the loaded
        state is copied INTO them (same storage, new values) and the device count re-seeded from the loaded one."""
        self._flush_fast()
        self._dev_tables = None           # the moment tensors may be replaced below
        if self._device_step is None:
            return super().load_state_dict(state_dict)
        old = {p: (st["exp_avg"], st["exp_avg_sq"]) for p, st in self.state.items() if len(st)}
        super().load_state_dict(state_dict)
        counts = set()
        with torch.no_grad():
            for p, st in self.state.items():
                if not len(st):
                    continue
                counts.add(int(torch.as_tensor(st["step"]).item()))
                if p in old:
                    m, v = old[p]
                    m.copy_(st["exp_avg"])
                    v.copy_(st["exp_avg_sq"])
                    st["exp_avg"], st["exp_avg_sq"] = m, v
            if len(counts) > 1:
                raise ValueError("device_step_counter(): the loaded state has more than one step count")
            k = float(counts.pop()) if counts else 0.0
            # parameters the loaded state does not mention (a step-0 checkpoint has an empty state; a partial one lacks some):
            # torch leaves self.state[p] empty for them, but a captured step still points at the OLD moment tensors -- they
            # restart from zero in place, at the loaded count
            for p, (m, v) in old.items():
                if not len(self.state[p]):
                    m.zero_()
                    v.zero_()
                    self.state[p]["step"] = torch.tensor(k, dtype=torch.float32)
                    self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"] = m, v
            self._device_step.fill_(k)

    def _tables(self, group, params):
        """ctypes pointer tables of one launch, cached while the same (parameter, gradient) OBJECTS come back -- the train
        loops keep their gradients in one persistent flat buffer (train.Trainer), so this is every step after the first."""
        t = getattr(self, "_dev_tables", None)
        if (t is not None and t["group"] is group and len(t["params"]) == len(params)
                and self._group_ok(group)              # weight_decay / amsgrad / maximize / a tensor lr set since: not this kernel's
                and all(p is q and p.grad is g for p, q, g in zip(params, t["params"], t["grads"]))
                and all(p.data_ptr() == a for p, a in zip(params, t["pptr"]))
                and all(g.data_ptr() == a for g, a in zip(t["grads"], t["gptr"]))):
            return t
        self._dev_tables = None
        if not self._hip_ok(group, params):
            return None
        n = len(params)
        arr = ctypes.c_void_p * n
        pptr = [p.data_ptr() for p in params]
        t = self._dev_tables = {
            "group": group, "params": list(params), "grads": [p.grad for p in params], "pptr": pptr, "n": n,
            "gptr": [p.grad.data_ptr() for p in params],
            "ps": arr(*pptr), "gs": arr(*[p.grad.data_ptr() for p in params]),
            "ms": arr(*[self.state[p]["exp_avg"].data_ptr() for p in params]),
            "vs": arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in params]),
            "ns": (ctypes.c_long * n)(*[p.numel() for p in params]), "dev": params[0].device}
        return t

    def _step_on_device(self):
        lib = _lib.get()
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            t = self._tables(group, params)
            if t is None:
                raise RuntimeError("device_step_counter(): a parameter group cannot run on the one-launch HIP update")
            beta1, beta2 = group["betas"]
            dev = t["dev"]
            with torch.cuda.device(dev):
                lib.check(lib.cpc_adam_step_capturable(t["ps"], t["gs"], t["ms"], t["vs"], t["ns"], t["n"], float(group["lr"]),
                                                       beta1, beta2, float(group["eps"]), self._device_step.data_ptr(),
                                                       self._device_coef.data_ptr(),
                                                       torch.cuda.current_stream(dev).cuda_stream), "adam_step_capturable")

    @staticmethod
    def _group_ok(group):
        """The group-level half of _hip_ok: options the one-launch kernel does not implement (the reference uses none)."""
        if group["weight_decay"] != 0 or group["amsgrad"] or group.get("maximize", False):
            return False
        if group.get("capturable", False) or group.get("differentiable", False):
            return False
        return not isinstance(group["lr"], torch.Tensor)

    @staticmethod
    def _hip_ok(group, params):
        if not Adam._group_ok(group):
            return False
        for p in params:
            g = p.grad
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                    and g.is_cuda and g.dtype == torch.float32 and not g.is_sparse and g.is_contiguous()):
                return False
        return len({p.device for p in params}) == 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._device_step is not None:
            self._step_on_device()
            return loss
        # fast path: ONE group whose parameters all carry a gradient and share the step count, same tensors as last time --
        # the count lives in a Python int (materialised into state[p]["step"] by sync_step_state / state_dict), the pointer
        # tables are reused: ~20 us of host time instead of ~0.25 ms of per-parameter bookkeeping
        fast = self._fast
        if fast is not None and len(self.param_groups) == 1:
            group = self.param_groups[0]
            params = group["params"]
            t = self._tables(group, params) if all(p.grad is not None for p in params) else None
            if t is not None and t is fast["tables"]:
                fast["k"] += 1
                beta1, beta2 = group["betas"]
                self._launch_tables(t, float(group["lr"]), beta1, beta2, float(group["eps"]), fast["k"])
                return loss
            self._flush_fast()
        elif fast is not None:
            self._flush_fast()
        rest = []
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if not self._hip_ok(group, params):
                rest.append(group)
                continue
            beta1, beta2 = group["betas"]
            steps = set()
            for p in params:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st["step"].is_cuda:                      # a state loaded from a fused / capturable torch Adam
                    st["step"] = st["step"].cpu()
                st["step"] += 1
                steps.add(int(st["step"].item()))
            # parameters of one group normally share the step count; if not (some were frozen for a while), one launch each
            for k in sorted(steps):
                sel = [p for p in params if int(self.state[p]["step"].item()) == k] if len(steps) > 1 else params
                self._launch(sel, float(group["lr"]), beta1, beta2, float(group["eps"]), k)
            if len(self.param_groups) == 1 and len(steps) == 1 and len(params) == len(group["params"]):
                t = self._tables(group, params)              # the next call with the same tensors takes the fast path
                if t is not None:
                    self._fast = {"tables": t, "k": steps.pop(), "params": list(params)}
        if rest:
            keep = self.param_groups
            self.param_groups = rest
            try:
                super().step()
            finally:
                self.param_groups = keep
        return loss

    def _launch_tables(self, t, lr, beta1, beta2, eps, step, stream=None):
        bc1 = 1.0 - beta1 ** step
        bc2s = math.sqrt(1.0 - beta2 ** step)
        with torch.cuda.device(t["dev"]):
            st = torch.cuda.current_stream(t["dev"]) if stream is None else stream
            _lib.get().check(_lib.get().cpc_adam_step(t["ps"], t["gs"], t["ms"], t["vs"], t["ns"], t["n"], lr, beta1, beta2, eps,
                                                      bc1, bc2s, st.cuda_stream), "adam_step")

    @torch.no_grad()
    def step_split(self, groups):
        """One optimiser step as SEVERAL launches of the same update: ``groups`` is a list of (parameters, stream) -- each set
        is updated on its stream (where its gradients become final: train.CompositeStep's open-tailed step has three such
        places); every parameter not named runs on the current stream.  Element-wise arithmetic: bit-identical to ``step()``.
        Only on the fast path of ``step()`` (one group, the tensors of the previous call, a shared step count); returns False --
        nothing launched, nothing counted -- when that does not apply, and the caller runs ``step()`` instead."""
        fast = self._fast
        if self._device_step is not None or fast is None or len(self.param_groups) != 1:
            return False
        group = self.param_groups[0]
        params = group["params"]
        if any(p.grad is None for p in params):
            return False
        t = self._tables(group, params)
        if t is None or t is not fast["tables"]:
            return False
        key = tuple(tuple(id(p) for p in ps) for ps, _ in groups)
        sp = t.get("split")
        if sp is None or sp["key"] != key:
            taken = set()
            parts = []
            for ids in key:
                sel = [p for p in params if id(p) in set(ids) and id(p) not in taken]
                taken.update(id(p) for p in sel)
                parts.append(sel)
            parts.append([p for p in params if id(p) not in taken])         # the remainder: current stream
            halves = []
            for sel in parts:
                n = len(sel)
                arr = ctypes.c_void_p * n
                halves.append({"n": n, "dev": t["dev"], "ps": arr(*[p.data_ptr() for p in sel]),
                               "gs": arr(*[p.grad.data_ptr() for p in sel]),
                               "ms": arr(*[self.state[p]["exp_avg"].data_ptr() for p in sel]),
                               "vs": arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in sel]),
                               "ns": (ctypes.c_long * n)(*[p.numel() for p in sel])})
            sp = t["split"] = {"key": key, "halves": halves}
        fast["k"] += 1
        beta1, beta2 = group["betas"]
        lr, eps = float(group["lr"]), float(group["eps"])
        for h, stream in zip(sp["halves"], [st for _, st in groups] + [None]):
            if h["n"]:
                self._launch_tables(h, lr, beta1, beta2, eps, fast["k"], stream=stream)
        return True

    def _launch(self, params, lr, beta1, beta2, eps, step):
        lib = _lib.get()
        n = len(params)
        arr = ctypes.c_void_p * n
        ps = arr(*[p.data_ptr() for p in params])
        gs = arr(*[p.grad.data_ptr() for p in params])
        ms = arr(*[self.state[p]["exp_avg"].data_ptr() for p in params])
        vs = arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in params])
        ns = (ctypes.c_long * n)(*[p.numel() for p in params])
        bc1 = 1.0 - beta1 ** step
        bc2s = math.sqrt(1.0 - beta2 ** step)
        dev = params[0].device
        with torch.cuda.device(dev):
            lib.check(lib.cpc_adam_step(ps, gs, ms, vs, ns, n, lr, beta1, beta2, eps, bc1, bc2s,
                                        torch.cuda.current_stream(dev).cuda_stream), "adam_step")
