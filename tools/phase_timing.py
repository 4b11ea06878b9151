#!/usr/bin/env python3
"""Developer tool: per-phase cycle split of the row-GEMM stages.  Needs tools/bin/libmmt_phase.so (tools/build_phase.sh: the `//@phase`
markers of rowgemm.h turned into cycle stamps in a patched copy of the sources)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MMT_LIB_PATH"] = os.path.join(ROOT, "tools", "bin", "libmmt_phase.so")
sys.path.insert(0, ROOT)
import torch
from multimodal_transformer_amd import multiTransformer as MT, _lib
B, T, d, h = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (32, 500, 128, 8)))
dev = torch.device("cuda:0")
lib = _lib.load()
raw = ctypes.CDLL(_lib.LIB_PATH)
NWG = (B * T + 31) // 32
buf = torch.zeros(NWG * 32, dtype=torch.int64, device=dev)
raw.mmt_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, 128, 0.1), 0.1), 6).to(dev).train()
x = torch.randn(B, T, d, device=dev, requires_grad=True)
mask = torch.ones(B, T, 1, device=dev)
for _ in range(2):
    enc(x, mask).sum().backward()
torch.cuda.synchronize()
assert raw.mmt_debug_set_phase_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
enc(x, mask).sum().backward()
torch.cuda.synchronize()
v = buf.cpu().numpy().reshape(NWG, 4, 8).sum(axis=0)
nwg = (B * T + 31) // 32
names = ["FRAG stages", "LNBWD stages", "PLAIN+LN stages", "PLAIN stages"]
ph = ["A staging(+LN)", "T copy of A", "k-loops+park", "chunk epilogues", "LNBWD epilogue", "-", "-", "-"]
for s in range(4):
    tot = v[s, :8].sum()
    print("%-16s total %8.0f cycles/WG/step: " % (names[s], tot / nwg) + "  ".join("%s %.0f" % (ph[i], v[s, i] / nwg) for i in range(8) if v[s, i]))
