"""Developer helper (GPU): attention core against the oracle at a few odd shapes (a checker script: it imports the oracle, so it is not product code)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch, numpy as np
import oracle
from multimodal_transformer_amd import functional as F
dev = torch.device("cuda:0")
for (B, T, d, h) in ((1, 64, 256, 8), (1, 64, 128, 8), (1, 32, 32, 1), (1, 96, 64, 2)):
    torch.manual_seed(0)
    q, k, v, g = (torch.randn(B, T, d) for _ in range(4))
    dk_ = d // h
    sp = lambda z: z.reshape(B, T, h, dk_).permute(0, 2, 1, 3)
    qd, kd, vd = (t.double().requires_grad_() for t in (q, k, v))
    ctx, _ = oracle.scaled_dot_attention(sp(qd), sp(kd), sp(vd), None)
    ref = ctx.permute(0, 2, 1, 3).reshape(B, T, d)
    (ref * g.double()).sum().backward()
    qg, kg, vg = (t.to(dev).requires_grad_() for t in (q, k, v))
    out = F.sdpa(qg, kg, vg, None, h)
    (out * g.to(dev)).sum().backward()
    print("== B%d T%d d%d h%d (dk=%d)" % (B, T, d, h, dk_))
    for name, a, b in (("ctx", out.detach(), ref.detach()), ("dq", qg.grad, qd.grad), ("dk", kg.grad, kd.grad), ("dv", vg.grad, vd.grad)):
        e = (a.cpu().double() - b)
        rel = e.norm() / b.norm()
        # error by feature-within-head and by position
        ef = e.reshape(B, T, h, dk_).pow(2).sum(dim=(0, 1, 2)).sqrt() / b.reshape(B, T, h, dk_).pow(2).sum(dim=(0, 1, 2)).sqrt()
        et = e.pow(2).sum(dim=(0, 2)).sqrt() / b.pow(2).sum(dim=(0, 2)).sqrt()
        print("  %-3s rel %.3e | by feature: %s | by t(first 8 of each 32): %s" % (
            name, rel, np.array2string(ef.numpy(), precision=2, max_line_width=200), np.array2string(et.numpy()[::4], precision=2, max_line_width=200)))
