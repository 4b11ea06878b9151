#!/usr/bin/env python3
"""Developer tool: where does attn_fwd_kernel<16, train> spend its time?  Runs the configs[3] encoder forward with the
timing-only ablation builds of the kernel (tools/bin/libmmt_abl.so, built with -DMMT_ABLATIONS; results are wrong by design)
and prints the kernel's mean launch time per variant.  One process per variant (the switch is read once).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMMT_ABLATIONS -o tools/bin/libmmt_abl.so multimodal_transformer_amd/csrc/api.hip
  python tools/attn_ablate.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "baseline", 1: "no running max / rescale / lane exchange", 2: "no exp", 3: "no PV product (MFMA)",
         5: "no row sums", 6: "stamped (correct results)"}
SEG = ["K read + QK^T + tile max + lane exchange", "rescale test + exp + row sums", "mask wait + selects", "V reads + packs + PV issue",
       "staged-tile load wait + LDS write", "barrier", "-", "loop top"]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from multimodal_transformer_amd import multiTransformer as MT, _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, T, d, h = 32, 500, 128, 8
    enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, 128, 0.1), 0.1), 6).to(dev).train()
    x = torch.randn(B, T, d, device=dev)
    mask = torch.ones(B, T, 1, device=dev)
    stamps = None
    if os.environ.get("MMT_ABL") == "6":
        import ctypes
        raw = ctypes.CDLL(_lib.LIB_PATH)
        raw.mmt_debug_set_attn_stamp_buffer.argtypes = [ctypes.c_void_p]
        stamps = torch.zeros(1024 * 4 * 16, dtype=torch.int64, device=dev)
        assert raw.mmt_debug_set_attn_stamp_buffer(ctypes.c_void_p(stamps.data_ptr())) == 0
    with torch.no_grad():
        for _ in range(20):
            enc(x, mask)
        torch.cuda.synchronize()
        _lib.profile(True)
        for _ in range(20):
            enc(x, mask)
        torch.cuda.synchronize()
    ms, n = _lib.profile_collect()["attn_fwd_kernel"]
    print("ABL=%s  %-58s attn_fwd %.2f us/launch" % (os.environ.get("MMT_ABL", "0"), NAMES[int(os.environ.get("MMT_ABL", "0"))], 1e3 * ms / n))
    if stamps is not None:
        import numpy as np
        full = stamps.cpu().numpy().reshape(-1, 16).astype(float)
        full = full[full[:, 9] > 0]
        life, t0, t1 = full[:, 6], full[:, 8], full[:, 9]
        print("  waves %d: lifetime median %.0f cycles (min %.0f max %.0f); entry spread %.2f us, exit spread %.2f us, first entry -> last exit %.2f us"
              % (len(full), np.median(life), life.min(), life.max(), (t0.max() - t0.min()) / 100.0, (t1.max() - t1.min()) / 100.0, (t1.max() - t0.min()) / 100.0))
        hw, xcc = full[:, 10].astype(np.int64), full[:, 11].astype(np.int64) & 0xF
        cu = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xF)      # (XCC, SE, SH, CU)
        ids, counts = np.unique(cu, return_counts=True)
        print("  distinct CUs used %d; waves per CU: %s" % (len(ids), dict(zip(*np.unique(counts, return_counts=True)))))
        per_cu_life = {c: life[cu == i].max() for i, c in zip(ids, counts)}
        for c in sorted(set(counts)):
            sel = np.isin(cu, ids[counts == c])
            print("    CUs holding %2d waves: lifetime median %.0f max %.0f" % (c, np.median(life[sel]), life[sel].max()))
        print("    lifetime percentiles 5/25/50/75/95: " + " ".join("%.0f" % np.percentile(life, q) for q in (5, 25, 50, 75, 95)))
        for x in range(8):
            sel = xcc == x
            if sel.any():
                print("    XCC %d: %4d waves, lifetime median %.0f  max %.0f; entry median +%.2f us, exit median +%.2f us" % (
                    x, sel.sum(), np.median(life[sel]), life[sel].max(), (np.median(t0[sel]) - t0.min()) / 100, (np.median(t1[sel]) - t0.min()) / 100))
        widx = np.arange(len(stamps) // 16)[stamps.cpu().numpy().reshape(-1, 16)[:, 9] > 0]
        bx = ((widx // 4) % 32) // 8                     # tile quad of the workgroup (attn_block: rem >> 3 within a group of 8 * nx ids)
        for q in range(4):
            sel = bx == q
            print("    query-tile quad %d: lifetime median %.0f" % (q, np.median(life[sel])))
        simd = (hw >> 4) & 3
        for q in range(4):
            print("    SIMD %d: %d waves, lifetime median %.0f" % (q, (simd == q).sum(), np.median(life[simd == q])))
        slow, fast = life > np.percentile(life, 90), life < np.percentile(life, 10)
        print("    segment cycles/tile of the slowest 10 %% of waves vs the fastest 10 %%:")
        for i, nm in enumerate(SEG):
            if full[:, i].sum() > 0 and i != 6:
                print("      %-44s %6.0f   %6.0f" % (nm, full[slow, i].mean() / 16, full[fast, i].mean() / 16))
        v = full[:, :8].copy()
        v[:, 6] = 0
        tot = v.sum(axis=1)
        print("  per wave and launch: %.0f cycles in the 16 tiles (median), per tile %.0f" % (float(__import__("numpy").median(tot)), float(__import__("numpy").median(tot)) / 16))
        for i, nm in enumerate(SEG):
            if v[:, i].sum() > 0:
                print("    %-44s %6.0f cycles/tile  %5.1f %%" % (nm, v[:, i].mean() / 16, 100 * v[:, i].sum() / tot.sum()))
else:
    for a in ([int(os.environ["MMT_ONLY"])] if os.environ.get("MMT_ONLY") else sorted(NAMES)):
        env = dict(os.environ, MMT_ABL=str(a), MMT_LIB_PATH=os.path.join(ROOT, "tools", "bin", "libmmt_abl.so"))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)
