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
         4: "operands from global memory: no LDS staging, no barrier", 5: "no row sums"}

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
else:
    for a in sorted(NAMES):
        env = dict(os.environ, MMT_ABL=str(a), MMT_LIB_PATH=os.path.join(ROOT, "tools", "bin", "libmmt_abl.so"))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)
