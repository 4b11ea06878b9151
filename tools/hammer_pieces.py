#!/usr/bin/env python3
"""Developer helper (GPU): run individual units forward+backward repeatedly while another process shares the GPU and report
which of them stop being bit-reproducible (see tools/hammer_probe.py for the whole stack)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from multimodal_transformer_amd import functional as F
from multimodal_transformer_amd import multiTransformer as MT

B, T, d, h = 32, 500, 128, 8
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
torch.manual_seed(1)
x = torch.randn(B, T, d, generator=g).to(dev)
go = torch.randn(B, T, d, generator=g).to(dev)
mask = torch.ones(B, T, 1, device=dev)
for i in range(B):
    mask[i, T - (7 * i) % T:] = 0
W = (torch.randn(d, d, generator=g) / d ** 0.5).to(dev)
bias = torch.randn(d, generator=g).to(dev)
a2, b2 = torch.ones(d, device=dev), torch.zeros(d, device=dev)
layer = MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, 128, 0.1), 0.1).to(dev).eval()
mha = layer.self_attn
ffn = layer.feed_forward


def grads(mod):
    return [p.grad.clone() for p in mod.parameters()]


def unit_linear():
    xg, wg, bg = x.clone().requires_grad_(), W.clone().requires_grad_(), bias.clone().requires_grad_()
    y = F.linear(xg, wg, bg)
    (y * go).sum().backward()
    return [y.detach(), xg.grad, wg.grad, bg.grad]


def unit_layernorm():
    xg, ag, bg = x.clone().requires_grad_(), a2.clone().requires_grad_(), b2.clone().requires_grad_()
    y = F.layer_norm(xg, ag, bg)
    (y * go).sum().backward()
    return [y.detach(), xg.grad, ag.grad, bg.grad]


def unit_mod(mod, call):
    for p in mod.parameters():
        p.grad = None
    xg = x.clone().requires_grad_()
    y = call(xg)
    (y * go).sum().backward()
    return [y.detach(), xg.grad] + grads(mod)


units = {
    "linear (rowgemm PLAIN + dx + wgrad)": unit_linear,
    "layer_norm": unit_layernorm,
    "MultiHeadedAttention module": lambda: unit_mod(mha, lambda xg: mha(xg, xg, xg, mask)),
    "PositionwiseFeedForward module": lambda: unit_mod(ffn, lambda xg: ffn(xg)),
    "EncoderLayer module (layerwise kernels)": lambda: unit_mod(layer, lambda xg: layer(xg, mask)),
}
for name, fn in units.items():
    ref = [t.clone() for t in fn()]
    torch.cuda.synchronize()
    bad, which = 0, set()
    for it in range(reps):
        cur = fn()
        torch.cuda.synchronize()
        d_ = [i for i, (a, b) in enumerate(zip(ref, cur)) if not torch.equal(a, b)]
        if d_:
            bad += 1
            which.update(d_)
    print("%-44s %d of %d repeats differ%s" % (name, bad, reps, (" (outputs %s; 0 = y, 1 = dx)" % sorted(which)) if bad else ""), flush=True)
