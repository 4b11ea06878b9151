"""The reference's optimiser as one launch: ``torch.optim.Adam(model.parameters(), lr, weight_decay)`` (transformer/SFT/train.py:621).

``FlatAdam`` takes the same arguments and gives the same updates (L2 weight decay added to the gradient, bias-corrected moments),
but runs on contiguous RANGES instead of tensors: parameters that are views of one buffer with gradients that are views of another
(the fused encoder stack keeps both that way) are updated as a single range, and all ranges go through ``mmt_adam_step`` in one
kernel launch.  Parameters without a gradient are skipped, like in torch.

State is kept the way ``torch.optim.Adam`` keeps it: ``self.state[p] = {"step", "exp_avg", "exp_avg_sq"}`` per parameter (``step`` a
CPU float tensor, one per parameter: a parameter that first receives a gradient later starts its own bias correction at 1, as in
torch), so ``state_dict()`` / ``load_state_dict()`` round-trip and are interchangeable with ``torch.optim.Adam``'s.  The moment tensors
are VIEWS of two flat buffers per group laid out in the parameters' memory order; the buffers are rebuilt (moments carried over) when
a parameter's storage moves or when ``load_state_dict`` installed fresh tensors."""
import ctypes

import torch

from . import _lib


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FlatAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._plans = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)     # installs fresh per-parameter tensors: the next step re-seats them into flat buffers
        self._plans = {}

    def _plan(self, gi, params):
        """Moments of a group live in two flat buffers laid out in the parameters' MEMORY order, so parameters that are neighbours in
        memory have neighbouring moments whatever ranges a step merges them into.  Rebuilt (moments carried over from ``self.state``)
        when a parameter's storage moves, e.g. when the encoder re-seats its parameters into its flat buffer after the optimiser was
        made, and after ``load_state_dict``."""
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
            n = q.numel()
            st = self.state.get(q)
            if st is not None and "exp_avg" in st:
                m[cur:cur + n].copy_(st["exp_avg"].reshape(-1))
                v[cur:cur + n].copy_(st["exp_avg_sq"].reshape(-1))
                st["exp_avg"] = m[cur:cur + n].view(q.shape)
                st["exp_avg_sq"] = v[cur:cur + n].view(q.shape)
                if not torch.is_tensor(st.get("step")):
                    st["step"] = torch.tensor(float(st.get("step", 0)))
            cur += n
        plan = {"sig": sig, "m": m, "v": v, "off": off}
        self._plans[gi] = plan
        return plan

    def _state_of(self, plan, q):
        st = self.state[q]
        if "exp_avg" not in st:
            o, n = plan["off"][id(q)], q.numel()
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = plan["m"][o:o + n].view(q.shape)
            st["exp_avg_sq"] = plan["v"][o:o + n].view(q.shape)
        return st

    @staticmethod
    def _ranges(params, steps=None):
        """Merge parameters that are neighbours in memory, with gradients that are neighbours too (and, given ``steps``, the same step
        count), into ranges: [(first parameter, n)]."""
        live = [p for p in params if p.grad is not None]
        for p in live:
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous():
                raise RuntimeError("FlatAdam handles contiguous fp32 parameters and gradients on the GPU (no CPU path)")
        live.sort(key=lambda p: p.data_ptr())
        out = []
        for p in live:
            s = steps[id(p)] if steps is not None else 0
            if out:
                q, n, pend, gend, sq = out[-1]
                if p.data_ptr() == pend and p.grad.data_ptr() == gend and s == sq:
                    out[-1] = (q, n + p.numel(), pend + 4 * p.numel(), gend + 4 * p.numel(), sq)
                    continue
            out.append((p, p.numel(), p.data_ptr() + 4 * p.numel(), p.grad.data_ptr() + 4 * p.numel(), s))
        return [(q, n) for q, n, _, _, _ in out]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            plan = self._plan(gi, group["params"])
            steps = {}
            for p in live:                                   # per-parameter step counts, advanced like torch's
                st = self._state_of(plan, p)
                st["step"] += 1
                steps[id(p)] = int(st["step"].item())
            mb, vb = plan["m"].data_ptr(), plan["v"].data_ptr()
            b1, b2 = group["betas"]
            for s in sorted(set(steps.values())):            # normally ONE value: every live parameter has had a gradient since step 1
                ranges = self._ranges([p for p in live if steps[id(p)] == s])
                nch = len(ranges)
                arr = ctypes.c_void_p * nch
                ps = arr(*[q.data_ptr() for q, _ in ranges])
                gs = arr(*[q.grad.data_ptr() for q, _ in ranges])
                ms = arr(*[mb + 4 * plan["off"][id(q)] for q, _ in ranges])
                vs = arr(*[vb + 4 * plan["off"][id(q)] for q, _ in ranges])
                ns = (ctypes.c_size_t * nch)(*[n for _, n in ranges])
                _lib.check(lib.mmt_adam_step(ps, gs, ms, vs, ns, nch, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                             float(group["weight_decay"]), s, _lib.stream_ptr()))
        return loss
