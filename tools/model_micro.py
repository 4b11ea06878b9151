#!/usr/bin/env python3
"""Whole-model step timing with the per-site HIP-event profile.
    python tools/model_micro.py mft|sft|sft256|b2 [T] [B]
mft: MultiTransformer (3 modalities + MFN gate, configs[2]); sft: NLPTransformer d=128 (configs[3]);
sft256: NLPTransformer with the reference's default embed_dim=256; b2: UniFullTransformer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_transformer_amd import multiTransformer as MT, _lib
kind = sys.argv[1] if len(sys.argv) > 1 else "mft"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
torch.manual_seed(1)
mods = ["acoustic", "image", "linguistic"]
dims = {"acoustic": 88, "image": 256, "linguistic": 300}
if kind == "mft":
    model = MT.MultiTransformer(mods, dims, device=dev)
    x = {m: torch.randn(B, T, dims[m], device=dev) for m in mods}
elif kind == "sft":
    model = MT.NLPTransformer(512, embed_dim=128, h=8, device=dev); x = torch.tanh(torch.randn(B, T, 512, device=dev))
elif kind == "sft256":
    model = MT.NLPTransformer(512, device=dev); x = torch.tanh(torch.randn(B, T, 512, device=dev))
elif kind == "pipe":
    from multimodal_transformer_amd import models as MM
    pd = {"acoustic": 88, "image": 1000, "linguistic": 300}
    pw = {"acoustic": 10, "image": 30, "linguistic": 33}
    model = MM.MultiCNNTransformer(mods, pd, device=dev)
    model.Transformer = MT.NLPTransformer(512, embed_dim=128, h=8, device=dev)
    x = {m: torch.randn(B, T, pw[m], pd[m], device=dev) for m in mods}
else:
    model = MT.UniFullTransformer(300, device=dev); x = torch.randn(B, T, 300, device=dev)
model.train()
mask = torch.ones(B, T, 1, device=dev)
tgt = torch.rand(B, T, 1, device=dev)
lengths = [T] * B
params = [p for p in model.parameters()]
def step():
    for p in params: p.grad = None
    out = model(x, lengths, mask) if kind == "pipe" else model(x, mask, lengths)
    loss = ((out - tgt) ** 2).sum() / float(B * T)
    loss.backward()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("%s T=%d B=%d: %.3f ms/step eager = %.0f windows/s" % (kind, T, B, dt * 1e3, B * T / dt))
_lib.profile(True)
for _ in range(3): step()
torch.cuda.synchronize()
prof = _lib.profile_collect()
_lib.profile(False)
tot = sum(v[0] for v in prof.values()) / 3
print("sum of profiled kernels %.3f ms/step" % tot)
for name, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("MMT_TOP", "14"))]:
    print("  %-48s %8.3f ms/step  %4d launches/step  %8.1f us/launch" % (name, ms / 3, cnt // 3, 1e3 * ms / cnt))
