#!/usr/bin/env python3
"""Window-encoder kernel timing at the configs[3] per-GPU sizes (B*T = 16000 windows per modality)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_transformer_amd import functional as F, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, W, D, Fo in (("image", 30, 1000, 256), ("linguistic", 33, 300, 300), ("acoustic", 10, 88, 256)):
    x = torch.randn(N, W, D, device=dev)
    w = (torch.randn(Fo, D, 2, device=dev) / (2 * D) ** 0.5).requires_grad_()
    b = torch.zeros(Fo, device=dev, requires_grad=True)
    g = torch.randn(N, Fo, device=dev)
    for _ in range(2):
        out, _ = F.conv_maxpool(x, w, b); (out * g).sum().backward()
    torch.cuda.synchronize()
    _lib.profile(True)
    for _ in range(5):
        out, _ = F.conv_maxpool(x, w, b); (out * g).sum().backward()
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile(False)
    flops = 2.0 * N * (W - 1) * 2 * D * Fo
    xbytes = N * W * D * 4.0
    for site in ("convpool_fwd_kernel", "convpool_bwd_kernel"):
        ms, n = prof[site]
        t = ms / n * 1e-3
        dense = flops if site.endswith("fwd_kernel") else 2.0 * N * 32 * ((W - 1 + 31) // 32) * 2 * D * Fo
        print("%-10s %-20s %8.1f us  algorithmic %.1f TFLOP/s  (issued %.1f)  x read %.2f TB/s" %
              (name, site, t * 1e6, flops / t / 1e12, dense / t / 1e12, xbytes / t / 1e12))
    del x
