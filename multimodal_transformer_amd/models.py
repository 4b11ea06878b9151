"""Drop-in for the reference's ``models.py`` window-encoder front-end: ``CNN``, ``Highway`` and the
``MultiCNNTransformer`` wrappers of the three Transformer variants — same class names, constructor arguments,
``forward(inputs, length, mask)`` contract and ``state_dict`` keys (so reference checkpoints load unchanged):

    transformer/SFT/models.py:27-142      early fusion: concat -> tanh(fusionLayer) -> NLPTransformer   (``MultiCNNTransformer``)
    transformer/MFT/models.py:27-138      per-modality dict -> MultiTransformer (MFN gate)               (``MultiCNNTransformerMFT``)
    transformer/B2-Trans/models.py:27-134 single modality -> UniFullTransformer                           (``MultiCNNTransformerB2``)

The reference walks the batch in a Python loop (SFT/models.py:123) and runs Conv1d + MaxPool1d per sequence; windows are
independent, so here all B*T windows of a modality go through ONE fused conv-GEMM + max-pool HIP kernel
(``csrc/convpool.h``), the Highway layer through the row GEMM, and the modalities on concurrent streams.
There is no CPU path.
"""
import torch
import torch.nn as nn

from . import _lib, functional as F_hip
from .multiTransformer import (MultiTransformer, NLPTransformer, UniFullTransformer, UniTransformer, _MOD_STREAMS,
                               _hip_device)


class Highway(nn.Module):
    """x_gate * proj(x) + (1 - x_gate) * x with a LINEAR projection (the reference applies no ReLU) — SFT/models.py:27-55."""

    def __init__(self, word_embed_size):
        super().__init__()
        self.word_embed_size = word_embed_size
        self.linear_projection = nn.Linear(word_embed_size, word_embed_size, bias=True)
        self.linear_gate = nn.Linear(word_embed_size, word_embed_size, bias=True)

    def forward(self, x_conv_out, dropout_p=0.0, seed=0):
        """``dropout_p`` / ``seed``: the front-end's Dropout(0.3) on the Highway output (SFT/models.py:132-134), fused into the combine"""
        return F_hip.highway(x_conv_out, self.linear_projection.weight, self.linear_projection.bias,
                             self.linear_gate.weight, self.linear_gate.bias, dropout_p, seed)       # drop(gate*proj + (1-gate)*x)


class CNN(nn.Module):
    """Conv1d(word_embed_size -> window_embed_size, k) + max over all positions — SFT/models.py:57-79.
    ``forward`` takes the reference's (batch, word_embed_size, window_length) layout; ``forward_windows`` takes the
    natural (N, window_length, word_embed_size) rows that the kernel reads (no permute copy)."""

    def __init__(self, word_embed_size=300, window_embed_size=128, k=2):
        super().__init__()
        self.k = k
        self.f = window_embed_size
        self.word_embed_size = word_embed_size
        self.window_embed_size = window_embed_size
        self.conv1d = nn.Conv1d(word_embed_size, window_embed_size, k, bias=True)

    def forward_windows(self, x):
        if self.k != 2:
            raise NotImplementedError("CNN: only the reference's kernel size k=2 is implemented on the HIP path")
        out, _ = F_hip.conv_maxpool(x, self.conv1d.weight, self.conv1d.bias)
        return out

    def forward(self, x_reshape):
        return self.forward_windows(x_reshape.permute(0, 2, 1).contiguous())


class _FrontEnd(nn.Module):
    window_embed_size = {"linguistic": 300, "emotient": 20, "acoustic": 256, "image": 256}     # SFT/models.py:90

    def _build(self, mods, dims, k):
        self.mods = mods
        self.dims = dims
        self.CNN, self.Highway = {}, {}
        total = 0
        for mod in mods:
            self.CNN[mod] = CNN(dims[mod], self.window_embed_size[mod], k)
            self.Highway[mod] = Highway(self.window_embed_size[mod])
            self.add_module("cnn_{}".format(mod), self.CNN[mod])
            self.add_module("highway_{}".format(mod), self.Highway[mod])
            total += self.window_embed_size[mod]
        return total

    def _encode(self, inputs):
        """{mod: (B,T,W,D)} -> {mod: (B,T,F_mod)}: conv+pool, Highway, Dropout(0.3), one stream per modality."""
        outs = {}
        p = float(self.dropout.p) if self.training else 0.0
        main, streams = _MOD_STREAMS.begin(self.device, len(self.mods))
        for i, (mod, st) in enumerate(zip(self.mods, streams)):
            with torch.cuda.stream(st):
                x = inputs[mod]
                B, T, W, D = x.shape
                e = self.CNN[mod].forward_windows(x.reshape(B * T, W, D))
                seed = _lib.next_dropout_seed(x.device, 6, holder=self, index=i) if p > 0.0 else 0      # one seed state per modality stream
                e = self.Highway[mod](e, p, seed)
                outs[mod] = e.reshape(B, T, -1)
        _MOD_STREAMS.end(main, streams, list(outs.values()))
        return outs


class MultiCNNTransformer(_FrontEnd):
    """SFT: transformer/SFT/models.py:81-142."""

    def __init__(self, mods, dims, fuse_embed_size=512, k=2, device=torch.device("cuda:0")):
        super().__init__()
        total = self._build(mods, dims, k)
        self.fusionLayer = nn.Linear(total, fuse_embed_size)
        if len(mods) > 1:
            self.Transformer = NLPTransformer(fuse_embed_size, device=device)
        else:
            self.Transformer = UniTransformer(total, device=device)
        self.dropout = nn.Dropout(p=0.3)
        self.device = _hip_device(device)
        self.to(self.device)

    def forward(self, inputs, length, mask=None):
        outs = self._encode(inputs)
        if len(self.mods) > 1:
            cat = F_hip.cat_cols([outs[m] for m in self.mods])
            fused = F_hip.linear(cat, self.fusionLayer.weight, self.fusionLayer.bias, act=2)      # tanh, :138
            return self.Transformer(fused, mask, length)
        return self.Transformer(outs[self.mods[0]], mask, length)


class MultiCNNTransformerMFT(_FrontEnd):
    """MFT: transformer/MFT/models.py:81-138 (``embed_dims`` replaces the fixed window embed sizes; no fusion layer)."""

    def __init__(self, mods, dims, embed_dims, fuse_embed_size=256, k=2, device=torch.device("cuda:0")):
        super().__init__()
        self.window_embed_size = embed_dims
        total = self._build(mods, dims, k)
        if len(mods) > 1:
            self.Transformer = MultiTransformer(mods=mods, window_embed_size=self.window_embed_size, device=device)
        else:
            self.Transformer = UniTransformer(total, device=device)
        self.dropout = nn.Dropout(p=0.3)
        self.device = _hip_device(device)
        self.to(self.device)

    def forward(self, inputs, length, mask=None):
        outs = self._encode(inputs)
        if len(self.mods) > 1:
            return self.Transformer(outs, mask, length)
        return self.Transformer(outs[self.mods[0]], mask, length)


class MultiCNNTransformerB2(_FrontEnd):
    """B2-Trans: transformer/B2-Trans/models.py:81-134 (single modality; the fusion layer is commented out there)."""

    def __init__(self, mods, dims, fuse_embed_size=512, k=2, device=torch.device("cuda:0")):
        super().__init__()
        total = self._build(mods, dims, k)
        self.Transformer = UniFullTransformer(total, device=device)
        self.dropout = nn.Dropout(p=0.3)
        self.device = _hip_device(device)
        self.to(self.device)

    def forward(self, inputs, length, mask=None):
        outs = self._encode(inputs)
        if len(outs) > 1:
            return self.Transformer(F_hip.cat_cols([outs[m] for m in self.mods]), mask, length)
        return self.Transformer(outs[self.mods[0]], mask, length)
