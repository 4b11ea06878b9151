#!/usr/bin/env python3
"""Developer helper (GPU): bit-reproducibility of the whole sequence models (SFT NLPTransformer with its LSTM decoder, MFT
MultiTransformer with the MFN gate) forward+backward while another process shares the GPU (start it first, see hammer_probe.py)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from multimodal_transformer_amd import multiTransformer as MT

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(5)
torch.manual_seed(2)
B, T = 32, 300
lengths = [max(1, T - 9 * i) for i in range(B)]
mask = torch.zeros(B, T, 1, device=dev)
for i, n in enumerate(lengths):
    mask[i, :n] = 1.0
mods = ["acoustic", "image", "linguistic"]
dims = {"acoustic": 88, "image": 256, "linguistic": 300}
models = {
    "NLPTransformer(512, embed_dim=128, h=8)": (MT.NLPTransformer(512, device=dev, embed_dim=128, h=8).to(dev).eval(),
                                               torch.tanh(torch.randn(B, T, 512, generator=g)).to(dev)),
    "NLPTransformer(512) default (d=256 decoder scans)": (MT.NLPTransformer(512, device=dev).to(dev).eval(),
                                                          torch.tanh(torch.randn(B, T, 512, generator=g)).to(dev)),
    "MultiTransformer(3 modalities) + MFN": (MT.MultiTransformer(mods, dims, device=dev).to(dev).eval(),
                                             {m: torch.randn(B, T, dims[m], generator=g).to(dev) for m in mods}),
}
for name, (model, inp) in models.items():
    def run():
        for p in model.parameters():
            p.grad = None
        out = model(inp, mask, lengths)
        out.sum().backward()
        torch.cuda.synchronize()
        return [out.detach().clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None]
    ref = run()
    pnames = ["out"] + [n for n, p in model.named_parameters() if p.grad is not None]
    bad, which = 0, set()
    for _ in range(reps):
        cur = run()
        d_ = [pnames[i] for i, (a, b) in enumerate(zip(ref, cur)) if not torch.equal(a, b)]
        if d_:
            bad += 1
            which.update(d_)
    print("%-52s %d of %d repeats differ%s" % (name, bad, reps, (": " + ", ".join(sorted(which))[:300]) if bad else ""), flush=True)
