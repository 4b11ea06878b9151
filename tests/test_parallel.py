"""Data-parallel path on CPU: world_size 2, gloo.  Checks that sharding sequences across ranks with
(a) the loss scaled by the GLOBAL window count and (b) a SUM all-reduce of flat gradient buffers equals the
single-process full-batch step of the reference (transformer/SFT/train.py:133-141).  The oracle stands in for
the HIP compute here (CPU container); the product's bucket / all-reduce code is what is under test."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
import recipe as R
from multimodal_transformer_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


D, H, N, T = 40, 4, 2, 12
LENGTHS = [12, 9, 7, 3]


def _params():
    from multimodal_transformer_amd import multiTransformer as MT
    enc = MT.Encoder(MT.EncoderLayer(D, MT.MultiHeadedAttention(H, D), MT.PositionwiseFeedForward(D, 128, 0.1), 0.1), N)
    return R.gen_params(R.shapes_of(enc.state_dict()), 21)


def _flat_step(p32, x, mask, tgt, n_windows):
    """One fwd+bwd with all parameters as views of ONE flat leaf (as the fused encoder returns its gradients)."""
    names = list(p32)
    flat = torch.cat([p32[k].reshape(-1) for k in names]).clone().requires_grad_()
    views, off = {}, 0
    for k in names:
        n = p32[k].numel()
        views[k] = flat[off:off + n].view(p32[k].shape)
        off += n
    y = oracle.encoder_stack(views, "", x, mask, H)
    loss = (((y - tgt) * mask) ** 2).sum() / float(n_windows)
    loss.backward()
    return flat


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        p32 = _params()
        B = len(LENGTHS)
        x = R.gen_normal("dp:x", (B, T, D), 21)
        tgt = R.gen_normal("dp:t", (B, T, D), 21)
        mask = R.prefix_mask(LENGTHS, T)
        lo, hi = parallel.shard_batch(B, rank, world)
        n_global = parallel.global_window_count(LENGTHS[lo:hi])
        assert n_global == sum(LENGTHS)
        flat = _flat_step(p32, x[lo:hi], mask[lo:hi], tgt[lo:hi], n_global)
        # the product path hands parameters gradient VIEWS of one flat buffer; emulate with nn.Parameter-likes
        holders = []
        off = 0
        for k in p32:
            n = p32[k].numel()
            q = torch.nn.Parameter(p32[k].clone())
            q.grad = flat.grad[off:off + n].view(p32[k].shape)
            holders.append(q)
            off += n
        loose = torch.nn.Parameter(torch.zeros(3))
        loose.grad = torch.full((3,), float(rank + 1))
        dead = torch.nn.Parameter(torch.zeros(2))            # no gradient: must be skipped
        bases, ls = parallel.gradient_buckets(holders + [loose, dead])
        assert len(bases) == 1 and bases[0].data_ptr() == flat.grad.data_ptr() and len(ls) == 1
        ncoll = parallel.allreduce_gradients(holders + [loose, dead])
        assert ncoll == 2
        assert torch.equal(loose.grad, torch.full((3,), 3.0))
        if rank == 0:
            ret.put(flat.grad.clone().numpy())
    finally:
        dist.destroy_process_group()


def test_dp_equals_full_batch_step():
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = ret.get()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    p32 = _params()
    x = R.gen_normal("dp:x", (len(LENGTHS), T, D), 21)
    tgt = R.gen_normal("dp:t", (len(LENGTHS), T, D), 21)
    mask = R.prefix_mask(LENGTHS, T)
    full = _flat_step(p32, x, mask, tgt, sum(LENGTHS)).grad.numpy()
    np.testing.assert_allclose(got, full, rtol=2e-4, atol=2e-6)


def test_shard_batch_covers_everything():
    for n in (1, 7, 32, 33):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_batch(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_a_noop():
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    assert parallel.allreduce_gradients([p]) == 0
    assert parallel.global_window_count([3, 4]) == 7
