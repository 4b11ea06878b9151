#!/usr/bin/env python3
"""Developer tool: one MFT training step (configs[2]: 3 modality stacks + MFN gate, 32 x 300) of the tree given as argv[1] (default:
this one): replayed hipGraph time with the modality streams on, and per-launch-site HIP-event time with the streams serialised.
Used for the same-box comparison of two rounds' trees (`python tools/mft_ab.py tools/bin/r3tree`)."""
import os
import sys
import time

ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_transformer_amd import multiTransformer as MT, _lib
from multimodal_transformer_amd.functional import mse_sum_loss_backward

dev = torch.device("cuda:0")
torch.manual_seed(1)
B, T = 32, 300
mods = ["acoustic", "image", "linguistic"]
dims = {"acoustic": 88, "image": 256, "linguistic": 300}
m = MT.MultiTransformer(mods, dims, device=dev).train()
ps = list(m.parameters())
x = {k: torch.randn(B, T, dims[k], device=dev) for k in mods}
mask = torch.ones(B, T, 1, device=dev)
tgt = torch.rand(B, T, 1, device=dev)


def step():
    for p in ps:
        p.grad = None
    mse_sum_loss_backward(m(x, mask, [T] * B), tgt, B * T)


for _ in range(3):
    step()
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
res = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    res.append(1e3 * (time.perf_counter() - t0) / 20)
print("%s: hipGraph replay %.3f ms/step (runs: %s)" % (ROOT, min(res), " ".join("%.3f" % r for r in res)))
os.environ["MMT_MODALITY_STREAMS"] = "0"
for _ in range(2):
    step()
torch.cuda.synchronize()
_lib.profile(True)
for _ in range(3):
    step()
torch.cuda.synchronize()
prof = _lib.profile_collect()
tot = sum(v[0] for v in prof.values()) / 3
print("  serialised kernel time %.3f ms/step in %d launches" % (tot, sum(v[1] for v in prof.values()) // 3))
for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:14]:
    print("    %-64s %7.3f ms/step  %4d launches" % (k[:64], v[0] / 3, v[1] // 3))
