"""The N-rank branch of bench.py — what the driver's 8-GPU scaling run executes — rehearsed on ONE GPU: two fresh rank processes
(started by bench.py itself from a plain `python bench.py --gpus 2`, the driver's form of the call; that relay process is a child of
the conftest launcher, i.e. not of a process that holds the GPU) run the step over gloo
(MMT_BENCH_REHEARSAL=1: both ranks on cuda:0).  Checks the contract of the JSON line: one line, from rank 0, whole-job value,
n_gpus, weak scaling, the gradient exchange timed, and that sharding + the SUM all-reduce leave a finite, positive throughput.
It is a code-path test, not a performance number (the real curve needs the 8-GPU node)."""
import json
import os
import sys

import pytest

import conftest

pytestmark = pytest.mark.gpu


def test_two_rank_bench_line():
    env = dict(os.environ, MMT_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    # called the way the driver calls it: plain `python bench.py --gpus 2 ...`, no launcher around it — bench.py starts its own ranks
    cmd = [sys.executable, os.path.join(conftest.ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-full-model", "--profile-steps", "0"]
    res = conftest.run_in_fresh_process(cmd, env, timeout=900)
    if res is None:
        pytest.skip("no launcher process (tests were not started through pytest_configure with GPU tests selected)")
    assert res["rc"] == 0, res["stderr"][-3000:]
    lines = [l for l in res["stdout"].splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3 and out["warmup"] == 1
    assert out["metric"] == "windows/sec fwd+bwd" and out["unit"] == "windows/s" and out["higher_is_better"] is True
    assert out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp2"       # 32 sequences per rank
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - 2 * 32 * 500 / (out["ms_per_step"] * 1e-3)) < 1e-2 * out["value"]   # whole-job windows / max-over-ranks time
    assert out["allreduce_ms"] > 0                          # the exchange ran and was timed
    assert out["with_adam"]["ms_per_step"] > 0
