import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


# ---- child-process launcher ------------------------------------------------------------------------------------------
# A process that has initialised the GPU must not exec another program (the GPU boxes refuse it), and pytest's own process has by
# the time a multi-process test runs.  So a plain helper process is started HERE, before anything touches the GPU; it starts the
# children (e.g. the two bench.py ranks of test_gpu_bench_rehearsal.py) on request and hands their output back.
_LAUNCHER_SRC = r"""
import json, subprocess, sys
for line in sys.stdin:
    req = json.loads(line)
    try:
        r = subprocess.run(req["cmd"], env=req["env"], cwd=req["cwd"], capture_output=True, text=True, timeout=req["timeout"])
        out = {"rc": r.returncode, "stdout": r.stdout[-20000:], "stderr": r.stderr[-20000:]}
    except subprocess.TimeoutExpired as e:
        out = {"rc": -9, "stdout": (e.stdout or b"").decode(errors="replace")[-20000:] if isinstance(e.stdout, bytes) else (e.stdout or ""),
               "stderr": "timeout"}
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
"""
_launcher = None


def run_in_fresh_process(cmd, env=None, timeout=600):
    """Run `cmd` as a child of the launcher (never of this process) -> {"rc", "stdout", "stderr"}; None if no launcher is running."""
    import json
    if _launcher is None or _launcher.poll() is not None:
        return None
    e = dict(os.environ if env is None else env)
    _launcher.stdin.write(json.dumps({"cmd": list(cmd), "env": e, "cwd": ROOT, "timeout": timeout}) + "\n")
    _launcher.stdin.flush()
    return json.loads(_launcher.stdout.readline())


def pytest_configure(config):
    global _launcher
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    expr = config.getoption("markexpr", "") or ""
    if "not gpu" not in expr and _launcher is None:
        import subprocess
        try:
            _launcher = subprocess.Popen([sys.executable, "-c", _LAUNCHER_SRC], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        except OSError:
            _launcher = None


def pytest_unconfigure(config):
    global _launcher
    if _launcher is not None:
        try:
            _launcher.stdin.close()
            _launcher.wait(timeout=10)
        except Exception:  # noqa: BLE001
            _launcher.kill()
        _launcher = None


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=True) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def grad_close(a, b, rtol, atol=1e-6):
    """||a-b||_2 <= rtol*||b||_2 + atol*sqrt(n): relative test with an absolute floor, for gradients
    that are analytically zero (e.g. the key-projection bias: softmax is shift-invariant)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum())) <= rtol * float(np.sqrt((b ** 2).sum())) + atol * np.sqrt(a.size)
