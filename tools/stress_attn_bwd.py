#!/usr/bin/env python3
"""Developer tool: the one-kernel attention backward (attn_bwd_pair.h) against the two-kernel path (MMT_NO_FUSED_ATTN_BWD=1) over many
sequence lengths of its range (257..512: every tile count 9..16, ragged and full last tiles), ragged batches, d_k = 16 and a padded
head (d_k = 10), with and without dropout.  Two child processes (the switch is read once per process); the same seeds give the same
masks, so the two paths compute the same gradients up to bf16 rounding of differently ordered sums."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from multimodal_transformer_amd import functional as F
dev = torch.device("cuda:0")
out = {}
for T in list(range(257, 513, 5)) + [288, 320, 352, 384, 416, 448, 480, 512]:
    for (d, h) in ((64, 4), (40, 4)):
        for p in (0.0, 0.1):
            g_ = torch.Generator(device="cpu").manual_seed(T * 7 + d)
            B = 2
            q, k, v, g = (torch.randn(B, T, d, generator=g_).to(dev) for _ in range(4))
            q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
            mask = torch.ones(B, T, 1, device=dev); mask[1, (T * 3) // 5:] = 0
            y = F.sdpa(q, k, v, mask, h, p, 4321 + T)
            y.backward(g)
            out["%%d_%%d_%%g" %% (T, d, p)] = np.concatenate([t.grad.cpu().numpy().reshape(-1) for t in (q, k, v)])
torch.cuda.synchronize()
np.savez(sys.argv[1], **out)
''' % ROOT
res = []
for unfused in (False, True):
    env = dict(os.environ)
    env.pop("MMT_NO_FUSED_ATTN_BWD", None)
    if unfused:
        env["MMT_NO_FUSED_ATTN_BWD"] = "1"
    out = "/tmp/stress_bwd_%d.npz" % unfused
    subprocess.run([sys.executable, "-c", CHILD, out], check=True, env=env)
    res.append(np.load(out))
worst = 0.0
for k in res[0].files:
    a, b = res[0][k].astype(np.float64), res[1][k].astype(np.float64)
    assert np.isfinite(a).all() and np.isfinite(b).all(), k
    rel = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
    worst = max(worst, rel)
    assert rel < 2e-2, (k, rel)
print("%d cases, worst rel-L2 between the one-kernel and the two-kernel backward: %.3e" % (len(res[0].files), worst))
