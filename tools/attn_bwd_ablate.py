#!/usr/bin/env python3
"""Developer tool: what does each part of attn_bwd_diag16_kernel<train> own of a launch?  Runs the configs[3] attention core
(forward + backward through the C ABI) with the timing-only ablation instances of the kernel (tools/bin/libmmt_abl.so, built with
-DMMT_ABLATIONS; MMT_BABL = bit mask of the parts left out, attn_bwd_diag.h; results are wrong by design) and prints the kernel's mean
launch time per variant.  One process per variant (the switch is read once per process).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMMT_ABLATIONS -o tools/bin/libmmt_abl.so multimodal_transformer_amd/csrc/api.hip
  python tools/attn_bwd_ablate.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BITS = {1: "barriers", 2: "dQ read-add-write", 4: "exp/dropout/dS", 8: "dV/dK MFMA", 16: "patch + dQ MFMA", 32: "next scores",
        64: "prologue loads", 128: "epilogue stores"}
MASKS = [0, 1, 2, 3, 4, 8, 16, 32, 64, 128, 192, 7, 24, 28, 56, 60, 59, 63, 255]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from multimodal_transformer_amd import functional as F, _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    B, T, d, h = 32, 500, 128, 8
    q, k, v = (torch.randn(B, T, d, device=dev, requires_grad=True) for _ in range(3))
    g = torch.randn(B, T, d, device=dev)
    mask = torch.ones(B, T, 1, device=dev)
    for _ in range(10):
        F.sdpa(q, k, v, mask, h, 0.1, 7).backward(g)
    torch.cuda.synchronize()
    _lib.profile(True)
    for _ in range(30):
        F.sdpa(q, k, v, mask, h, 0.1, 7).backward(g)
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    ms, n = prof["attn_bwd_diag16_kernel"]
    m = int(os.environ.get("MMT_BABL", "0"))
    what = " + ".join(nm for b, nm in BITS.items() if m & b) or "baseline (nothing left out)"
    print("BABL=%-3d attn_bwd_diag16 %6.2f us/launch   without: %s" % (m, 1e3 * ms / n, what), flush=True)
else:
    for m in MASKS:
        env = dict(os.environ, MMT_LIB_PATH=os.path.join(ROOT, "tools", "bin", "libmmt_abl.so"))
        if m:
            env["MMT_BABL"] = str(m)
        else:
            env.pop("MMT_BABL", None)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)
