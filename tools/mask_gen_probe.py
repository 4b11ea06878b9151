import sys, os
sys.path.insert(0, os.getcwd())
import torch
from multimodal_transformer_amd import multiTransformer as MT, _lib
dev = torch.device("cuda:0")
B, T, d, h = 32, 500, 128, 8
x = torch.randn(B, T, d, device=dev)
mask = torch.ones(B, T, 1, device=dev)
for p in (0.1, 0.25, 0.5, 0.3, 0.0999):
    enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d, dropout=p), MT.PositionwiseFeedForward(d, 128, p), p), 6).to(dev).train()
    with torch.no_grad():
        for _ in range(3): enc(x, mask)
        torch.cuda.synchronize()
        _lib.profile(True)
        for _ in range(10): enc(x, mask)
        torch.cuda.synchronize()
        r = _lib.profile_collect()
        _lib.profile(False)
    ms, n = r["attn_mask_gen_kernel"]
    print("p=%.4f thr16=%#x  mask_gen %.2f us/launch" % (p, round(p * 65536), 1e3 * ms / n), flush=True)
