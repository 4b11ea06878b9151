#!/usr/bin/env python3
"""Attention-core micro-benchmark (sdpa forward+backward through the C ABI) for profiling runs."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_transformer_amd import functional as F, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--T", type=int, default=500)
ap.add_argument("--d", type=int, default=128)
ap.add_argument("--h", type=int, default=8)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--p", type=float, default=0.0, help="dropout on the probabilities (train mode)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
q, k, v = (torch.randn(a.B, a.T, a.d, device=dev, requires_grad=True) for _ in range(3))
g = torch.randn(a.B, a.T, a.d, device=dev)
mask = torch.ones(a.B, a.T, 1, device=dev)
for _ in range(3):
    F.sdpa(q, k, v, mask, a.h, a.p, 7).backward(g)
torch.cuda.synchronize()
_lib.profile(True)
t0 = time.perf_counter()
for _ in range(a.iters):
    F.sdpa(q, k, v, mask, a.h, a.p, 7).backward(g)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
for name, (ms, n) in sorted(_lib.profile_collect().items()):
    print("%-28s %8.2f us/launch (%d launches)" % (name, 1e3 * ms / n, n))
print("wall %.3f ms/iter" % (1e3 * dt / a.iters))
