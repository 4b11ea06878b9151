#!/usr/bin/env python3
"""Developer helper (GPU): does any kernel of the encoder path depend on LDS it did not write?  Runs the C4-size encoder stack
forward+backward, fills every CU's LDS with a bit pattern (mmt_debug_poison_lds), runs again and compares bit for bit."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from multimodal_transformer_amd import functional as F
from multimodal_transformer_amd import multiTransformer as MT

B, T, d, h = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (32, 500, 128, 8)
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
torch.manual_seed(1)
enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, 128, 0.1), 0.1), 6).to(dev).eval()
x = torch.randn(B, T, d, generator=g).to(dev)
go = torch.randn(B, T, d, generator=g).to(dev)
mask = torch.ones(B, T, 1, device=dev)
for i in range(B):
    mask[i, T - (7 * i) % T:] = 0


def run():
    for p in enc.parameters():
        p.grad = None
    xg = x.clone().requires_grad_()
    y = enc(xg, mask)
    (y * go).sum().backward()
    torch.cuda.synchronize()
    return y.detach().clone(), xg.grad.clone(), torch.cat([p.grad.reshape(-1) for p in enc.parameters()]).clone()


ref = run()
for pattern in (0x7FC00000, 0xFFFFFFFF, 0x7F800000, 0x3F800000, 0x00000000):
    F.poison_lds(dev, pattern)
    cur = run()
    for name, a, b in zip(("y", "dx", "gw"), ref, cur):
        diff = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
        n = int(diff.sum())
        print("pattern %08x %s: %s" % (pattern, name, "identical" if n == 0 else "%d elements differ (nan in result: %s)" % (n, bool(torch.isnan(b).any()))))
