"""Build libmmt_hip.so (gfx950) in-tree with hipcc.  `python -m multimodal_transformer_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmt_hip.so")
SOURCES = ["api.hip"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    headers = [f for f in os.listdir(CSRC) if f.endswith(".h")]
    deps = [os.path.join(CSRC, f) for f in SOURCES + headers] + [os.path.join(HERE, "..", "include", "mmt_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # NOTE: do NOT add `-mllvm -amdgpu-mfma-vgpr-form`: on ROCm 7.2 it deterministically miscompiles
    # attn_bwd_dq_kernel<32,*> (27 % wrong dQ; same source is correct without it) — see DESIGN.md.
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-result", "-Wno-unused-value", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
