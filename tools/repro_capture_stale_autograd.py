#!/usr/bin/env python3
"""Minimal repro of the host segfault in torch.cuda.graphs.capture_end (round-3 records gpurun_out/dbg4, dbg5), and of the guard.

A training step is captured while the autograd graph of an earlier EAGER step is still referenced.  No kernel of this repository is
involved in case `plain`: the model is a torch.nn.Linear.

  python tools/repro_capture_stale_autograd.py            # runs cases `clean` and `guarded`, each in a child process, and reports
  python tools/repro_capture_stale_autograd.py --with-plain   # MANUAL USE ONLY: also the case that crashes while holding the GPU
  case plain   : torch only, stale reference kept, torch.cuda.graph() directly        -> expected: the child dies (SIGSEGV) in capture_end
  case guarded : same, through multimodal_transformer_amd.graphs.capture_step          -> expected: StaleAutogradGraphError, exit code 0
  case clean   : no stale reference, through capture_step                              -> expected: captured and replayed, exit code 0
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    case = sys.argv[2]
    sys.path.insert(0, ROOT)
    import torch
    dev = torch.device("cuda:0")
    lin = torch.nn.Linear(64, 64).to(dev)
    x = torch.randn(8, 64, device=dev)

    def step():
        lin.weight.grad = None
        lin.bias.grad = None
        y = lin(x).sum()
        y.backward()
        return y

    keep = step()                       # eager warm-up on the default stream; `keep` holds its autograd graph (AccumulateGrad nodes)
    if case == "clean":
        keep = keep.detach()
    torch.cuda.synchronize()
    if case == "plain":
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        print("capturing", flush=True)
        with torch.cuda.graph(g):
            step()
        print("captured", flush=True)
        g.replay()
        torch.cuda.synchronize()
        print("replayed", flush=True)
    else:
        from multimodal_transformer_amd import graphs
        try:
            g, _ = graphs.capture_step(step)
        except graphs.StaleAutogradGraphError as e:
            print("refused before capture_begin:", str(e)[:90], "...", flush=True)
            sys.exit(0 if case == "guarded" else 3)
        g.replay()
        torch.cuda.synchronize()
        print("captured and replayed", flush=True)
        sys.exit(0 if case == "clean" else 4)
else:
    cases = ("clean", "guarded") + (("plain",) if "--with-plain" in sys.argv else ())
    for case in cases:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", case], capture_output=True, text=True, timeout=300)
        tail = (r.stdout.strip().splitlines() or [""])[-1]
        err = [l for l in r.stderr.splitlines() if "Fatal" in l or "Segmentation" in l or "Error" in l][:2]
        print("case %-8s exit code %4d   last line: %s   %s" % (case, r.returncode, tail, " | ".join(err)), flush=True)
