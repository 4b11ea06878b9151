"""The reference's optimiser as one launch: ``torch.optim.Adam(model.parameters(), lr, weight_decay)`` (transformer/SFT/train.py:621).

``FlatAdam`` takes the same arguments and gives the same updates (L2 weight decay added to the gradient, bias-corrected moments),
but runs on contiguous RANGES instead of tensors: parameters that are views of one buffer with gradients that are views of another
(the fused encoder stack keeps both that way) are updated as a single range, and all ranges go through ``mmt_adam_step`` in one
kernel launch.  Parameters without a gradient are skipped, like in torch."""
import ctypes

import torch

from . import _lib


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FlatAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plans = {}

    def _plan(self, gi, params):
        """Moments of a group live in two flat buffers laid out in the parameters' MEMORY order, so parameters that are neighbours in
        memory have neighbouring moments whatever ranges a step merges them into.  Rebuilt (moments carried over per parameter) when a
        parameter's storage moves, e.g. when the encoder re-seats its parameters into its flat buffer after the optimiser was made."""
        sig = tuple(q.data_ptr() for q in params)
        plan = self._plans.get(gi)
        if plan is not None and plan["sig"] == sig:
            return plan
        order = sorted(params, key=lambda q: q.data_ptr())
        total = sum(q.numel() for q in order)
        dev = order[0].device
        m = torch.zeros(total, dtype=torch.float32, device=dev)
        v = torch.zeros(total, dtype=torch.float32, device=dev)
        off, cur = {}, 0
        for q in order:
            off[id(q)] = cur
            if plan is not None and id(q) in plan["off"]:
                o = plan["off"][id(q)]
                m[cur:cur + q.numel()].copy_(plan["m"][o:o + q.numel()])
                v[cur:cur + q.numel()].copy_(plan["v"][o:o + q.numel()])
            cur += q.numel()
        plan = {"sig": sig, "m": m, "v": v, "off": off, "step": plan["step"] if plan else 0}
        self._plans[gi] = plan
        return plan

    @staticmethod
    def _ranges(params):
        """Merge parameters that are neighbours in memory, with gradients that are neighbours too, into ranges: [(first parameter, n)]."""
        live = [p for p in params if p.grad is not None]
        for p in live:
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous():
                raise RuntimeError("FlatAdam handles contiguous fp32 parameters and gradients on the GPU (no CPU path)")
        live.sort(key=lambda p: p.data_ptr())
        out = []
        for p in live:
            if out:
                q, n, pend, gend = out[-1]
                if p.data_ptr() == pend and p.grad.data_ptr() == gend:
                    out[-1] = (q, n + p.numel(), pend + 4 * p.numel(), gend + 4 * p.numel())
                    continue
            out.append((p, p.numel(), p.data_ptr() + 4 * p.numel(), p.grad.data_ptr() + 4 * p.numel()))
        return [(q, n) for q, n, _, _ in out]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            ranges = self._ranges(group["params"])
            if not ranges:
                continue
            plan = self._plan(gi, group["params"])
            plan["step"] += 1
            nch = len(ranges)
            arr = ctypes.c_void_p * nch
            mb, vb = plan["m"].data_ptr(), plan["v"].data_ptr()
            ps = arr(*[q.data_ptr() for q, _ in ranges])
            gs = arr(*[q.grad.data_ptr() for q, _ in ranges])
            ms = arr(*[mb + 4 * plan["off"][id(q)] for q, _ in ranges])
            vs = arr(*[vb + 4 * plan["off"][id(q)] for q, _ in ranges])
            ns = (ctypes.c_size_t * nch)(*[n for _, n in ranges])
            b1, b2 = group["betas"]
            _lib.check(lib.mmt_adam_step(ps, gs, ms, vs, ns, nch, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                         float(group["weight_decay"]), int(plan["step"]), _lib.stream_ptr()))
        return loss
