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

    @staticmethod
    def _hip_ok(group, params):
        if group["weight_decay"] != 0 or group["amsgrad"] or group.get("maximize", False):
            return False
        if group.get("capturable", False) or group.get("differentiable", False):
            return False
        if isinstance(group["lr"], torch.Tensor):
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
        if rest:
            keep = self.param_groups
            self.param_groups = rest
            try:
                super().step()
            finally:
                self.param_groups = keep
        return loss

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
