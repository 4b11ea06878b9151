#!/usr/bin/env python3
"""Developer helper (GPU): which intermediate buffer of the 1-layer backward is the first to differ between identical runs while
another process shares the GPU?  Calls the C ABI directly and checksums the workspace (mmt_debug_encoder_bwd_checksums)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from multimodal_transformer_amd import _lib
from multimodal_transformer_amd import multiTransformer as MT

B, T, d, h, f = 32, 500, 128, 8, 128
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
PDROP = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0       # > 0: train mode with a fixed seed
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1
NAMES = ["dx2T", "dhT", "dxa(cur: LN-bwd out)", "dxb(other: dx1)", "lnpart2", "dO R", "dO T", "delta", "dx1T", "dqkv", "dqkvT", "lnpart1"]
lib = _lib.load()
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
torch.manual_seed(1)
enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, f, 0.1), 0.1), N).to(dev).eval()
flat = torch.cat([q.reshape(-1) for q in enc.flat_parameters()]).detach().contiguous()
x = torch.randn(B, T, d, generator=g).to(dev)
dy = torch.randn(B, T, d, generator=g).to(dev)
mask = torch.ones(B, T, 1, device=dev)
for i in range(B):
    mask[i, T - (7 * i) % T:] = 0
nbytes = lib.mmt_encoder_workspace_bytes(B, T, d, h, f, N)
ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
y, dx, dp = torch.empty_like(x), torch.empty_like(x), torch.empty_like(flat)
sums = torch.zeros(12, dtype=torch.int64, device=dev)
st = _lib.stream_ptr()


def run():
    _lib.check(lib.mmt_encoder_forward(_lib.ptr(x), _lib.ptr(mask), _lib.ptr(flat), _lib.ptr(y), _lib.ptr(ws), nbytes, B, T, d, h, f, N, 1e-6, PDROP, 77, st))
    _lib.check(lib.mmt_encoder_backward(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(mask), _lib.ptr(flat), _lib.ptr(dx), _lib.ptr(dp), _lib.ptr(ws), nbytes,
                                        B, T, d, h, f, N, 1e-6, PDROP, 77, st))
    _lib.check(lib.mmt_debug_encoder_bwd_checksums(_lib.ptr(ws), B, T, d, h, f, N, _lib.ptr(sums), st))
    torch.cuda.synchronize()
    return sums.cpu().tolist(), dx.clone()


ref, dxref = run()
bad = 0
for it in range(1, reps):
    cur, dxc = run()
    diff = [NAMES[i] for i in range(12) if cur[i] != ref[i]]
    if diff or not torch.equal(dxc, dxref):
        bad += 1
        print("run %d: %s%s" % (it, ", ".join(diff) if diff else "(no buffer)", "" if torch.equal(dxc, dxref) else " | dx"), flush=True)
print("%d of %d repeats differ" % (bad, reps - 1))
