#!/usr/bin/env python3
"""Developer helper (GPU): which gradients of a 1-layer encoder stack change between identical runs while ANOTHER process keeps
the GPU busy (start e.g. `python bench.py --workload C5e --steps 6000 ... &` first)?  Prints, per differing run, the parameter
tensors and the sequences of dx that differ from run 0."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from multimodal_transformer_amd import multiTransformer as MT

B, T, d, h = 32, 500, 128, 8
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
torch.manual_seed(1)
enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, 128, 0.1), 0.1), N).to(dev).eval()
x = torch.randn(B, T, d, generator=g).to(dev)
go = torch.randn(B, T, d, generator=g).to(dev)
mask = torch.ones(B, T, 1, device=dev)
for i in range(B):
    mask[i, T - (7 * i) % T:] = 0
names = [n for n, _ in enc.named_parameters()]


def run():
    for p in enc.parameters():
        p.grad = None
    xg = x.clone().requires_grad_()
    y = enc(xg, mask)
    (y * go).sum().backward()
    torch.cuda.synchronize()
    return [y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in enc.parameters()]


ref = run()
bad = 0
for it in range(1, reps):
    cur = run()
    diffs = [n for n, a, b in zip(["y", "dx"] + names, ref, cur) if not torch.equal(a, b)]
    if diffs:
        bad += 1
        rows = (ref[1] != cur[1]).any(dim=2)
        seqs = sorted(set(int(r[0]) for r in rows.nonzero()))
        a, b = ref[1][rows].double(), cur[1][rows].double()
        rel = float((a - b).norm() / a.norm()) if a.numel() else 0.0
        per = ["%d:%d" % (s_, int(rows[s_].sum())) for s_ in seqs[:6]]
        short = [n.replace("layers.0.", "").replace("self_attn.linears.", "lin").replace("sublayer.", "sub") for n in diffs]
        print("run %d: %s | dx rows differing per sequence %s of %d | rel L2 of those rows %.2e" % (it, ", ".join(short), per, T, rel))
print("%d of %d repeats differ" % (bad, reps - 1))
