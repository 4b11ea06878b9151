#!/usr/bin/env python3
"""Developer tool: where does attn_bwd_diag16_kernel<train> spend its tile time?  Needs tools/bin/libmmt_abl.so (built with
-DMMT_ABLATIONS); runs the configs[3] encoder forward+backward with the stamped variant of the kernel (MMT_ABL=7)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MMT_ABL"] = "7"
os.environ["MMT_LIB_PATH"] = os.path.join(ROOT, "tools", "bin", "libmmt_abl.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
from multimodal_transformer_amd import multiTransformer as MT, _lib

SEG = ["loop top", "exp + dropout + dS", "packs + dV/dK MFMAs + patch writes", "patch reads + dQ MFMA",
       "next tile: row constants, operand reads, score MFMAs", "barrier", "dQ accumulation (LDS read-add-write)"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T, d, h = 32, 500, 128, 8
enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, 128, 0.1), 0.1), 6).to(dev).train()
x = torch.randn(B, T, d, device=dev, requires_grad=True)
mask = torch.ones(B, T, 1, device=dev)
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.mmt_debug_set_attn_stamp_buffer.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(B * h * 16 * 16, dtype=torch.int64, device=dev)
for _ in range(5):
    enc(x, mask).sum().backward()
torch.cuda.synchronize()
assert raw.mmt_debug_set_attn_stamp_buffer(ctypes.c_void_p(stamps.data_ptr())) == 0
_lib.profile(True)
for _ in range(5):
    enc(x, mask).sum().backward()
torch.cuda.synchronize()
ms, n = _lib.profile_collect()["attn_bwd_diag16_kernel"]
print("attn_bwd_diag16 (stamped) %.2f us/launch" % (1e3 * ms / n))
full = stamps.cpu().numpy().reshape(-1, 16).astype(float)
live = full[full[:, 9] == 1]
life = live[:, 7]
print("live waves %d: lifetime median %.0f cycles (min %.0f max %.0f) = %.0f per tile" % (len(live), np.median(life), life.min(), life.max(), np.median(life) / 16))
tot = live[:, :7].sum()
for i, nm in enumerate(SEG):
    print("  %-56s %6.0f cycles/tile  %5.1f %%" % (nm, live[:, i].mean() / 16, 100 * live[:, i].sum() / tot))
# prologue (kernel entry -> first step), sweep, epilogue (last step -> stores retired) per workgroup; s_memtime bases differ between XCDs,
# so only differences inside one workgroup are formed
wg = full.reshape(B * h, 16, 16)
pro = wg[:, :, 10].max(axis=1); swp = wg[:, :, 7].max(axis=1); epi = (wg[:, :, 15] - wg[:, :, 14]).max(axis=1)
print("per workgroup, cycles: prologue median %.0f (max %.0f) | sweep median %.0f (min %.0f max %.0f) | epilogue median %.0f (max %.0f) | sum %.0f"
      % (np.median(pro), pro.max(), np.median(swp), swp.min(), swp.max(), np.median(epi), epi.max(), np.median(pro + swp + epi)))
# in-kernel clock: s_memtime ticks (shader clock) over s_memrealtime ticks (constant 100 MHz) around the same body, kernel entry -> stores retired
ck, rt = live[:, 11], live[:, 12]
ok = rt > 0
ghz = ck[ok] / rt[ok] * 0.1
print("in-kernel clock (s_memtime / s_memrealtime x 100 MHz), live waves: median %.3f GHz (5%% %.3f, 95%% %.3f); kernel entry -> exit median %.0f cycles = %.2f us"
      % (np.median(ghz), np.percentile(ghz, 5), np.percentile(ghz, 95), np.median(ck[ok]), np.median(rt[ok]) / 100.0))
span = (full[:, 8].max() - full[full[:, 13] > 0][:, 13].min()) / 100.0
print("first workgroup entry -> last exit of the LAST stamped launch (100 MHz clock, chip-wide): %.2f us" % span)
