#!/usr/bin/env python3
"""Developer tool: per-launch-site time of one MFT training step (configs[2] by default; `--c4` for the configs[4] slice), modality
streams serialised so that HIP-event brackets are meaningful, plus the wall time of the eager step."""
import os
import sys
import time

os.environ["MMT_MODALITY_STREAMS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_transformer_amd import multiTransformer as MT, _lib

dev = torch.device("cuda:0")
torch.manual_seed(1)
B, T = (64, 1000) if "--c4" in sys.argv else (32, 300)
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
    (((m(x, mask, [T] * B) - tgt) ** 2).sum() / float(B * T)).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print("eager wall %.3f ms/step" % (1e3 * (time.perf_counter() - t0) / 5))
_lib.profile(True)
for _ in range(3):
    step()
torch.cuda.synchronize()
prof = _lib.profile_collect()
tot = sum(v[0] for v in prof.values()) / 3
print("sum of bracketed sites %.3f ms/step" % tot)
for k, (ms, n) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print("  %-56s %7.3f ms/step  (%d launches/step, %.1f us each)" % (k, ms / 3, n // 3, 1e3 * ms / n))
