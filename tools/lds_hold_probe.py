#!/usr/bin/env python3
"""Platform probe (GPU): does LDS content survive while another process shares the GPU?  Run with a second GPU process active
(e.g. `python bench.py --workload C5e --steps 8000 ... &`).  For several LDS sizes, workgroups hold a pattern for a few ms and
re-check it."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from multimodal_transformer_amd import _lib

lib = _lib.load()
dev = torch.device("cuda:0")
for kb in (32, 60, 64, 68, 96, 150, 160):
    bad = torch.zeros(2, dtype=torch.int32, device=dev)
    for rep in range(20):
        _lib.check(lib.mmt_debug_lds_hold(kb * 1024, 8, 512, _lib.ptr(bad), _lib.stream_ptr()))
    torch.cuda.synchronize()
    b = bad.cpu().tolist()
    print("LDS %3d KB: %d workgroups ran, %d words changed" % (kb, b[1], b[0]), flush=True)

bad = torch.zeros(2, dtype=torch.int32, device=dev)
for rep in range(40):
    _lib.check(lib.mmt_debug_vgpr_hold(8, 1024, _lib.ptr(bad), _lib.stream_ptr()))
torch.cuda.synchronize()
b = bad.cpu().tolist()
print("VGPRs (224 per lane): %d workgroups ran, %d registers changed" % (b[1], b[0]), flush=True)

bad = torch.zeros(2, dtype=torch.int32, device=dev)
for rep in range(40):
    _lib.check(lib.mmt_debug_compute_hold(100000, 2048, _lib.ptr(bad), _lib.stream_ptr()))
torch.cuda.synchronize()
b = bad.cpu().tolist()
print("arithmetic (division, shuffles, FMA): %d workgroups ran, %d results changed" % (b[1], b[0]), flush=True)
