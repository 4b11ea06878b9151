#!/usr/bin/env python3
"""Developer helper (GPU): run the attention core forward+backward several times on the same inputs and report whether
dq / dk / dv are bit-identical between runs, and where the first differences sit.  python tools/attn_determinism.py [B T d h]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from multimodal_transformer_amd import functional as F

B, T, d, h = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (32, 500, 128, 8)
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
q, k, v, go = (torch.randn(B, T, d, generator=g).to(dev) for _ in range(4))
lengths = [T - (7 * i) % T for i in range(B)]
mask = torch.zeros(B, T, 1, device=dev)
for i, n in enumerate(lengths):
    mask[i, :n] = 1.0
runs = []
nsd = int(sys.argv[sys.argv.index('--sdpa-reps') + 1]) if '--sdpa-reps' in sys.argv else 4
for it in range(0 if '--skip-sdpa' in sys.argv else nsd):
    qg, kg, vg = (t.clone().requires_grad_() for t in (q, k, v))
    out = F.sdpa(qg, kg, vg, mask, h)
    (out * go).sum().backward()
    torch.cuda.synchronize()
    runs.append((out.detach().clone(), qg.grad.clone(), kg.grad.clone(), vg.grad.clone()))
for it in range(1, len(runs)):
    for name, a, b in zip(("out", "dq", "dk", "dv"), runs[0], runs[it]):
        diff = (a != b)
        n = int(diff.sum())
        if n:
            idx = diff.nonzero()[:6].tolist()
            print("run %d %s: %d elements differ, max |d| %.3e, first at (b, t, col) %s" % (it, name, n, float((a - b).abs().max()), idx))
        elif nsd <= 4:
            print("run %d %s: identical" % (it, name))

if "--encoder" in sys.argv:
    # same question for the whole encoder stack (eval mode): which runs differ from run 0, and in which windows
    from multimodal_transformer_amd import multiTransformer as MT
    torch.manual_seed(1)
    N = 6
    enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, 128, 0.1), 0.1), N).to(dev).eval()
    x = torch.randn(B, T, d, generator=g).to(dev)
    ref = None
    bad = 0
    reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 40
    for it in range(reps):
        xg = x.clone().requires_grad_()
        y = enc(xg, mask)
        (y * go).sum().backward()
        torch.cuda.synchronize()
        cur = (y.detach().clone(), xg.grad.clone())
        if ref is None:
            ref = cur
            continue
        for name, a, b_ in zip(("y", "dx"), ref, cur):
            diff = (a != b_)
            if int(diff.sum()):
                bad += 1
                rows = diff.any(dim=2).nonzero()
                bs = sorted(set(int(r_[0]) for r_ in rows))
                ts = sorted(set(int(r_[1]) // 32 for r_ in rows))
                print("encoder run %d %s: %d elements differ (max %.3e); sequences %s; query tiles %s" % (
                    it, name, int(diff.sum()), float((a - b_).abs().max()), bs[:8], ts))
    print("encoder: %d of %d repeats differ from run 0" % (bad, reps - 1))
