#!/usr/bin/env python3
"""Developer tool: where does attn_fwd_kernel<16, train> spend its time?  Needs tools/bin/libmmt_abl.so (tools/build_diag.sh); runs the
configs[3] attention core forward with the stamped twin of the kernel (MMT_ABL=6), after checking that the twin's results are
bit-identical to the product kernel's.  Slots: see the first line of tools/bin/gen/attn_fwd_diag.h."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

SLOTS = [l.split()[1] for l in open(os.path.join(ROOT, "tools", "bin", "gen", "attn_fwd_diag.h")).readline().split("Marker slots:")[1].split(",")]
what = {"prologue": "Q fragment, first K / V tile staged, first barrier", "loop_top": "loop overhead, priority, next tile's loads issued",
        "qk_max": "K fragment read, QK^T, tile maximum", "exp_sum": "rescale test, exponentials, row sums", "mask": "mask word wait + selects",
        "pv": "V fragment reads, packs, PV issue", "stage_wait": "staged tile: global loads landed, LDS writes", "barrier": "workgroup barrier",
        "swept": "after the last tile", "exit": "normalise, stores retired"}


def run(lib, stamped, B, T, d, h, p):
    """one process per library: returns (dq, dk, dv[, stamps, us per launch])"""
    import subprocess
    code = r'''
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, %r)
from multimodal_transformer_amd import functional as F, _lib
dev = torch.device("cuda:0"); torch.manual_seed(0)
B, T, d, h, p = %d, %d, %d, %d, %f
q, k, v = (torch.randn(B, T, d, device=dev, requires_grad=True) for _ in range(3))
g = torch.randn(B, T, d, device=dev); mask = torch.ones(B, T, 1, device=dev)
for _ in range(5): F.sdpa(q, k, v, mask, h, p, 7).backward(g)
torch.cuda.synchronize()
stamps = None
if %d:
    raw = ctypes.CDLL(_lib.LIB_PATH); raw.mmt_debug_set_attn_stamp_buffer.argtypes = [ctypes.c_void_p]
    stamps = torch.zeros(B * h * 16 * 32, dtype=torch.int64, device=dev)
    assert raw.mmt_debug_set_attn_stamp_buffer(ctypes.c_void_p(stamps.data_ptr())) == 0
q.grad = k.grad = v.grad = None
_lib.profile(True)
for _ in range(20):
    q.grad = k.grad = v.grad = None
    y_ = F.sdpa(q, k, v, mask, h, p, 7)
    y_.backward(g)
torch.cuda.synchronize()
prof = _lib.profile_collect()
name = [n for n in prof if n.startswith("attn_fwd")][0]
np.savez(sys.argv[1], dq=y_.detach().cpu().numpy(), dk=k.grad.cpu().numpy(), dv=v.grad.cpu().numpy(), us=1e3 * prof[name][0] / prof[name][1],
         stamps=(stamps.cpu().numpy() if stamps is not None else np.zeros(1)))
''' % (ROOT, B, T, d, h, p, 1 if stamped else 0)
    out = "/tmp/fstamps_%d.npz" % (1 if stamped else 0)
    env = dict(os.environ)
    if lib:
        env["MMT_LIB_PATH"] = lib
    if stamped:
        env["MMT_ABL"] = "6"
    else:
        env.pop("MMT_ABL", None)
    subprocess.run([sys.executable, "-c", code, out], check=True, env=env)
    return np.load(out)


B, T, d, h, p = 32, 500, 128, 8, 0.1
if len(sys.argv) > 1:
    B, T, d, h = (int(a) for a in sys.argv[1:5])
abl = os.path.join(ROOT, "tools", "bin", "libmmt_abl.so")
ref = run(None, False, B, T, d, h, p)
st = run(abl, True, B, T, d, h, p)
for n in ("dq", "dk", "dv"):
    assert np.array_equal(ref[n], st[n]), "stamped twin differs from the product kernel in " + n
print("stamped twin == product kernel, bit for bit (context rows; dK, dV of the backward that consumes its statistics); product %.2f us/launch, stamped %.2f us/launch" % (float(ref["us"]), float(st["us"])))
full = st["stamps"].reshape(-1, 32).astype(float)
full = full[full[:, 31] == 1]
nw = len(full)
life = full[:, 28]
ghz = life / (full[:, 30] - full[:, 29]) * 0.1
print("waves %d; entry -> exit median %.0f cycles (min %.0f, max %.0f) at %.2f GHz = %.2f us" % (nw, np.median(life), life.min(), life.max(), np.median(ghz), np.median(life) / np.median(ghz) / 1e3))
span = (full[:, 30].max() - full[:, 29].min()) / 100.0
print("first entry -> last exit chip-wide (100 MHz clock): %.2f us" % span)
print("  all live waves (cycles per wave; 16 key tiles each):")
for i, n in enumerate(SLOTS[1:], start=1):
    print("    %-12s %8.0f cycles  %5.1f %%   %s" % (n, full[:, i].mean(), 100 * full[:, i].mean() / full[:, 28].mean(), what.get(n, "")))
