"""Adam for the training loops (``torch.optim.Adam(model.parameters(), lr)`` + ``optimizer.step()``,
GripNet-pose.py:104,146; the same line in every other driver) as ONE launch over all parameters.

    opt = gripnet_amd.optim.Adam(model.parameters(), lr=0.01)
    ...
    opt.zero_grad(); loss.backward(); opt.step()

The update is torch/optim/adam.py's (no amsgrad, no maximize); the step counter lives on the device, so a training step that
ends in ``opt.step()`` can be captured in a hipGraph and replayed (the table of tensor addresses travels in the launch's
arguments).  fp32 CUDA parameters only: this is the optimizer of the models of this package, not a general one.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _hip


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("bad Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._table = {}                                       # group index -> (addresses the table was built from, ctypes array)
        self._arrived = {}                                     # group index -> the launch's ticket (device int32, never saved)

    def load_state_dict(self, state_dict):
        """torch's loader, then: the moments and step counters are other tensors now, so the address tables are dropped, and
        the group's step counter (torch copies `param_groups` as they are: a CPU tensor after `torch.load(map_location="cpu")`)
        is moved to the parameters' device by the next `step()`."""
        super().load_state_dict(state_dict)
        self._table.clear()

    def _group_state(self, gi, group):

        dev = None
        for p in group["params"]:
            _hip.require_gpu(p)
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("gripnet_amd.optim.Adam updates contiguous fp32 parameters")
            dev = p.device if dev is None else dev
            if p.device != dev:
                raise ValueError("the parameters of one group must live on one device")
            st = self.state[p]
            if not st:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            for name in ("exp_avg", "exp_avg_sq"):             # (a loaded state: held to what the kernel assumes)
                m = st[name]
                if m.device != p.device or m.dtype != torch.float32 or m.shape != p.shape or not m.is_contiguous():
                    st[name] = m.to(device=p.device, dtype=torch.float32).reshape(p.shape).contiguous()
        # one device counter per group (all its tensors step together).  It is part of `param_groups`, so that a checkpoint
        # carries it; whatever a loader left there (a CPU tensor, a number) becomes an fp32 scalar on the parameters' device.
        step = group.get("step")
        if not (torch.is_tensor(step) and step.device == dev and step.dtype == torch.float32 and step.dim() == 0):
            group["step"] = torch.as_tensor(0.0 if step is None else float(step), dtype=torch.float32).to(dev)
            self._table.pop(gi, None)
        group.pop("_arrived", None)                            # (checkpoints of earlier versions carried the ticket)
        arrived = self._arrived.get(gi)
        if arrived is None or arrived.device != dev:
            arrived = self._arrived[gi] = torch.zeros((1,), dtype=torch.int32, device=dev)
        return dev, arrived

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            dev, arrived = self._group_state(gi, group)
            grads = []
            for p in live:
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32:
                    raise TypeError("gripnet_amd.optim.Adam takes dense fp32 gradients")
                grads.append(g if g.is_contiguous() else g.contiguous())
            key = tuple((p.data_ptr(), g.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel())
                        for p, g in zip(live, grads))
            cached = self._table.get(gi)
            if cached is None or cached[0] != key:             # (gradients are new tensors after zero_grad(set_to_none=True))
                arr = (_hip.AdamTensor * len(live))()
                for k, (p, g) in enumerate(zip(live, grads)):
                    st = self.state[p]
                    arr[k] = _hip.AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                cached = (key, arr)
                self._table[gi] = cached
            b1, b2 = group["betas"]
            _hip._call("gn_adam_step_f32", C.cast(cached[1], C.c_void_p), len(live), _hip.ptr(group["step"]), _hip.ptr(arrived), 4,
                       float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), _hip.stream_ptr(dev))
            if _hip._recorder is not None:
                _hip._recorder.keep.extend(grads)
        return loss
