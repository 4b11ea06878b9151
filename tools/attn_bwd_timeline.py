#!/usr/bin/env python3
"""Developer tool (one-off): per-wave timeline of the one-kernel attention backward: when does each wave start the X and Y phase of each
step, how many polls does its progress-word wait take.  Needs tools/bin/libmmt_tl.so (a patched experiment build)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MMT_LIB_PATH"] = os.path.join(ROOT, "tools", "bin", "libmmt_tl.so")
os.environ["MMT_ABL"] = "7"
sys.path.insert(0, ROOT)
import numpy as np, torch
from multimodal_transformer_amd import functional as F, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, T, d, h, p = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 500, 128, 8, 0.1
q, k, v = (torch.randn(B, T, d, device=dev, requires_grad=True) for _ in range(3))
g = torch.randn(B, T, d, device=dev); mask = torch.ones(B, T, 1, device=dev)
for _ in range(5): F.sdpa(q, k, v, mask, h, p, 7).backward(g)
torch.cuda.synchronize()
raw = ctypes.CDLL(_lib.LIB_PATH); raw.mmt_debug_set_attn_stamp_buffer.argtypes = [ctypes.c_void_p]
st = torch.zeros(B * h * 8 * 64, dtype=torch.int64, device=dev)
assert raw.mmt_debug_set_attn_stamp_buffer(ctypes.c_void_p(st.data_ptr())) == 0
for _ in range(3): F.sdpa(q, k, v, mask, h, p, 7).backward(g)
torch.cuda.synchronize()
a = st.cpu().numpy().reshape(B * h, 8, 64).astype(np.int64)
for wg in (100,):
    t0 = a[wg, :, 0].min()
    print("workgroup %d (cycles since the first wave's X(0); rows: waves 0..7 = pairs 0,2,4,6,1,3,5,7)" % wg)
    for w in range(8):
        x = a[wg, w, 0:16] - t0; y = a[wg, w, 16:32] - t0; pl = a[wg, w, 32:48]
        print("  w%d X:" % w, " ".join("%6d" % v for v in x))
        print("     Y:", " ".join("%6d" % (v if v > -10**9 else -1) for v in y), " polls:", " ".join("%d" % v for v in pl))
dx = np.diff(a[:, :, 0:16], axis=2)
print("step length (X(t+1) - X(t)) median over all waves, per step:", " ".join("%d" % v for v in np.median(dx.reshape(-1, 15), axis=0)))
print("X -> Y start (steps >= 2), median: %d" % np.median((a[:, :, 18:32] - a[:, :, 2:16])))
print("polls: mean %.2f, zero-poll share %.2f" % (a[:, :, 34:48].mean(), (a[:, :, 34:48] == 0).mean()))
