"""Drop-in module surface for the reference's ``multiTransformer.py`` on MI355X.

Same class names, constructor keywords, ``forward`` signatures and ``state_dict`` keys as
transformer/{SFT,MFT,B2-Trans}/multiTransformer.py, so a reference checkpoint
loads and the reference ``models.py`` / ``train.py`` can import this module in its place.
All arithmetic runs in hand-written HIP kernels through ``libmmt_hip.so``; calling a module
on CPU tensors raises (there is no CPU path).

Deviations, all documented in DESIGN.md:
  * ``MultiHeadedAttention.attn`` stays ``None``: the (B,h,T,T) probability tensor is never
    materialised (the reference writes it at :59 and never reads it).
  * train-mode dropout draws from a counter-based generator inside the kernels, not from torch's
    global generator: training-mode parity with the reference is statistical, eval-mode is numerical.
"""
import copy

import torch
import torch.nn as nn

from . import functional as F_hip


def clones(module, N):
    """N independent deep copies (all start from the same initial values, as in the reference :78-79)."""
    return nn.ModuleList(copy.deepcopy(module) for _ in range(N))


def _hip_device(device):
    return device if torch.cuda.is_available() else torch.device("cpu")


class LayerNorm(nn.Module):
    """Reference LayerNorm (:81-91): unbiased std, eps added to the std."""

    def __init__(self, features, eps=1e-6):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x):
        return F_hip.layer_norm(x, self.a_2, self.b_2, self.eps)


class PositionwiseFeedForward(nn.Module):
    """w_2(dropout(relu(w_1(x))))  (:9-20)."""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        hidden = F_hip.linear(x, self.w_1.weight, self.w_1.bias, act=1)
        return F_hip.linear(self.dropout(hidden), self.w_2.weight, self.w_2.bias)


def attention(query, key, value, mask=None, dropout=None):
    """Scaled dot-product attention on (B,h,T,d_k) tensors (:22-34).

    ``mask`` is the reference's (B,1,T,1) (or (B,T,1)) float mask: rows where it is 0 are blanked.
    Returns (context (B,h,T,d_k), None): the probabilities are not materialised.
    """
    if dropout is not None and getattr(dropout, "p", 0.0) > 0.0 and getattr(dropout, "training", False):
        raise NotImplementedError("attention(): probability dropout is only available through the fused Encoder path")
    B, h, T, dk = query.shape

    def merge(z):
        return z.transpose(1, 2).reshape(B, T, h * dk)

    m = None
    if mask is not None:
        if mask.numel() != B * T:
            raise NotImplementedError("attention(): only the query-row mask (B,1,T,1)/(B,T,1) of the reference is supported")
        m = mask.reshape(B, T, 1)
    ctx = F_hip.sdpa(merge(query), merge(key), merge(value), m, h)
    return ctx.reshape(B, T, h, dk).transpose(1, 2), None


class MultiHeadedAttention(nn.Module):
    """(:36-65).  ``linears`` = [query, key, value, output] projections."""

    def __init__(self, h, d_model, dropout=0.1):
        super().__init__()
        assert d_model % h == 0
        self.d_k = d_model // h
        self.h = h
        self.linears = clones(nn.Linear(d_model, d_model), 4)
        self.attn = None
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, query, key, value, mask=None):
        if self.training and self.dropout.p > 0.0:
            raise NotImplementedError("MultiHeadedAttention alone: train-mode probability dropout is only available "
                                      "through the fused Encoder path; call .eval() or set dropout=0")
        B = query.size(0)
        q, k, v = (F_hip.linear(x, l.weight, l.bias) for l, x in zip(self.linears, (query, key, value)))
        m = None
        if mask is not None:
            if mask.numel() != B * query.size(1):
                raise NotImplementedError("only the reference's query-row mask of shape (B,T,1) is supported")
            m = mask.reshape(B, -1, 1)
        ctx = F_hip.sdpa(q, k, v, m, self.h)
        return F_hip.linear(ctx, self.linears[3].weight, self.linears[3].bias)


class SublayerConnection(nn.Module):
    """x + dropout(sublayer(norm(x)))  (:93-104)."""

    def __init__(self, size, dropout):
        super().__init__()
        self.norm = LayerNorm(size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, sublayer):
        return x + self.dropout(sublayer(self.norm(x)))


class EncoderLayer(nn.Module):
    """(:106-116)."""

    def __init__(self, size, self_attn, feed_forward, dropout):
        super().__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.sublayer = clones(SublayerConnection(size, dropout), 2)
        self.size = size

    def forward(self, x, mask):
        x = self.sublayer[0](x, lambda z: self.self_attn(z, z, z, mask))
        return self.sublayer[1](x, self.feed_forward)

    def _is_standard(self):
        a, f = self.self_attn, self.feed_forward
        return (type(a) is MultiHeadedAttention and type(f) is PositionwiseFeedForward
                and a.linears[0].weight.shape == (self.size, self.size)
                and f.w_1.weight.shape[1] == self.size and f.w_2.weight.shape[0] == self.size)

    def _flat_order(self):
        """Parameters in the order the fused stack expects (= registration order of the reference)."""
        a, f, s = self.self_attn, self.feed_forward, self.sublayer
        out = []
        for l in a.linears:
            out += [l.weight, l.bias]
        out += [f.w_1.weight, f.w_1.bias, f.w_2.weight, f.w_2.bias,
                s[0].norm.a_2, s[0].norm.b_2, s[1].norm.a_2, s[1].norm.b_2]
        return out


class Encoder(nn.Module):
    """N layers and a final LayerNorm (:67-76), executed as one fused HIP stack (one forward and one
    backward library call for the whole stack) when the layers are the standard composition."""

    def __init__(self, layer, N):
        super().__init__()
        self.layers = clones(layer, N)
        self.norm = LayerNorm(layer.size)
        self._seed_counter = 0

    def _fusable(self):
        if len(self.layers) == 0:
            return False
        l0 = self.layers[0]
        if not all(type(l) is EncoderLayer and l._is_standard() for l in self.layers):
            return False
        h, f = l0.self_attn.h, l0.feed_forward.w_1.weight.shape[0]
        p = l0.sublayer[0].dropout.p
        for l in self.layers:
            if l.self_attn.h != h or l.feed_forward.w_1.weight.shape[0] != f or l.size != l0.size:
                return False
            if any(abs(q - p) > 0 for q in (l.sublayer[0].dropout.p, l.sublayer[1].dropout.p,
                                            l.self_attn.dropout.p, l.feed_forward.dropout.p)):
                return False
        return True

    def flat_parameters(self):
        ps = []
        for l in self.layers:
            ps += l._flat_order()
        return ps + [self.norm.a_2, self.norm.b_2]

    def forward(self, x, mask):
        if not self._fusable():
            for layer in self.layers:
                x = layer(x, mask)
            return self.norm(x)
        l0 = self.layers[0]
        p = l0.sublayer[0].dropout.p if self.training else 0.0
        flat = torch.cat([q.reshape(-1) for q in self.flat_parameters()])
        self._seed_counter += 1
        seed = (torch.initial_seed() * 1000003 + self._seed_counter) & 0x7FFFFFFFFFFFFFFF if p > 0.0 else 0
        return F_hip.encoder_stack(x, mask, flat, l0.self_attn.h, l0.feed_forward.w_1.weight.shape[0], len(self.layers),
                                   eps=self.norm.eps, dropout_p=p, seed=seed)
