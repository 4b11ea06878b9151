"""The data-parallel exchange on hardware: HIP gradients of the fused encoder through ``parallel.allreduce_gradients`` under an
RCCL ("nccl") process group of ONE rank — RCCL initialisation, the flat-gradient-buffer discovery and the in-place collective
all run on the GPU; a SUM over one rank must leave every gradient bit-identical.  (The N > 1 arithmetic is covered by the
world-size-2 gloo test in test_parallel.py; the driver runs the real 8-GPU job.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import recipe as R

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_hip_gradients_through_one_rank_rccl_group():
    from multimodal_transformer_amd import multiTransformer as MT, parallel
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    dev = torch.device("cuda:0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        model = MT.NLPTransformer(512, embed_dim=128, h=8, N=2, device=dev).eval()
        model.load_state_dict(R.gen_params(R.shapes_of(model.state_dict()), 37))
        B, T, lengths = 3, 40, [40, 22, 5]
        x = torch.tanh(R.gen_normal("rccl:x", (B, T, 512), 37)).to(dev)
        mask = R.prefix_mask(lengths, T).to(dev)
        n_global = parallel.global_window_count(lengths)
        assert n_global == sum(lengths)
        loss = (model(x, mask, lengths) ** 2).sum() / float(n_global)
        loss.backward()
        params = list(model.parameters())
        bases, loose = parallel.gradient_buckets(params)
        assert len(bases) == 1 and len(loose) > 0                 # the encoder's flat buffer + embed / decoder / read-out gradients
        before = [p.grad.clone() for p in params]
        base_ptr = bases[0].data_ptr()
        ncoll = parallel.allreduce_gradients(params, force=True)
        torch.cuda.synchronize()
        assert ncoll == 2                                         # one in-place collective on the flat buffer, one coalesced
        for p, b in zip(params, before):
            assert torch.equal(p.grad, b)                         # SUM over one rank: bit-identical
        assert parallel.gradient_buckets(params)[0][0].data_ptr() == base_ptr      # reduced in place, no staging copy
    finally:
        dist.destroy_process_group()
