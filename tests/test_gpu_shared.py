"""Opt-in GPU test (MMT_TEST_SHARED_GPU=1): bit-reproducibility of the encoder stack's forward+backward while a SECOND process
keeps the GPU busy.  This is the condition under which the retired LayerNorm-backward epilogue variant produced
runs that differed by a rounding; on an unshared GPU it never showed.  Not part of the default suite: it starts another GPU
process and takes about a minute."""
import os
import subprocess
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("MMT_TEST_SHARED_GPU") != "1", reason="opt-in: MMT_TEST_SHARED_GPU=1")]


def test_encoder_stack_is_bit_reproducible_while_the_gpu_is_shared():
    from multimodal_transformer_amd import multiTransformer as MT
    hammer = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "C5e", "--steps", "6000", "--warmup", "2",
                               "--no-full-model", "--no-cpu-baseline", "--profile-steps", "0"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    try:
        time.sleep(20)                                      # the other process imports torch and reaches its timed loop
        assert hammer.poll() is None, "the second GPU process ended before the test started"
        dev = torch.device("cuda:0")
        B, T, d, h, N = 32, 500, 128, 8, 6
        g = torch.Generator(device="cpu").manual_seed(3)
        torch.manual_seed(1)
        enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, 128, 0.1), 0.1), N).to(dev).eval()
        x = torch.randn(B, T, d, generator=g).to(dev)
        go = torch.randn(B, T, d, generator=g).to(dev)
        mask = torch.ones(B, T, 1, device=dev)
        for i in range(B):
            mask[i, T - (7 * i) % T:] = 0

        def run():
            for p in enc.parameters():
                p.grad = None
            xg = x.clone().requires_grad_()
            y = enc(xg, mask)
            (y * go).sum().backward()
            torch.cuda.synchronize()
            return [y.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in enc.parameters()]

        ref = run()
        differing = 0
        for _ in range(40):
            differing += any(not torch.equal(a, b) for a, b in zip(ref, run()))
        assert hammer.poll() is None, "the second GPU process ended during the test: the GPU was not shared throughout"
        assert differing == 0, "%d of 40 repeats differ from the first run while the GPU is shared" % differing
    finally:
        hammer.terminate()
        try:
            hammer.wait(timeout=30)
        except subprocess.TimeoutExpired:
            hammer.kill()
