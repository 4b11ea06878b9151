#!/usr/bin/env python3
"""Probe: can the gradient all-reduce (RCCL, one rank) be captured in the same hipGraph as the encoder step?"""
import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from multimodal_transformer_amd import multiTransformer as MT, parallel
dev = torch.device("cuda:0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["MASTER_ADDR"] = "127.0.0.1"
s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
enc = MT.Encoder(MT.EncoderLayer(128, MT.MultiHeadedAttention(8, 128), MT.PositionwiseFeedForward(128, 128, 0.1), 0.1), 2).to(dev).train()
x = torch.randn(4, 100, 128, device=dev, requires_grad=True)
mask = torch.ones(4, 100, 1, device=dev)
params = list(enc.parameters())
def step():
    for p in params: p.grad = None
    enc(x, mask).sum().backward()
    return parallel.allreduce_gradients(params, force=True)
for _ in range(3): step()
torch.cuda.synchronize()
for mode in ("thread_local", "global"):
    try:
        st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(2): step()
        torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=mode):
            n = step()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        print("mode %s: captured %d collectives; replay ok; grad norm %.4f" % (mode, n, float(params[0].grad.norm())), flush=True)
    except Exception as e:  # noqa: BLE001
        print("mode %s: FAILED: %s" % (mode, str(e).splitlines()[0]), flush=True)
        torch.cuda.synchronize()
dist.destroy_process_group()
