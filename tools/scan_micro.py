#!/usr/bin/env python3
"""LSTM-scan micro-benchmark (decoder shape of the SFT model) for profiling runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_transformer_amd import functional as F, _lib
T, B, H = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (500, 32, 128)))
dev = torch.device("cuda:0")
torch.manual_seed(0)
gx = torch.randn(T, B, 4 * H, device=dev, requires_grad=True)
W = (torch.randn(4 * H, H, device=dev) / H ** 0.5).requires_grad_()
c0 = torch.zeros(B, H, device=dev)
g = torch.randn(T, B, H, device=dev)
for _ in range(2):
    h, c = F.lstm_scan(gx, W, None, c0); (h * g).sum().backward()
torch.cuda.synchronize()
_lib.profile(True)
for _ in range(5):
    h, c = F.lstm_scan(gx, W, None, c0); (h * g).sum().backward()
torch.cuda.synchronize()
for name, (ms, n) in sorted(_lib.profile_collect().items()):
    print("%-28s %8.1f us/launch  = %.3f us/step" % (name, 1e3 * ms / n, 1e3 * ms / n / T))
