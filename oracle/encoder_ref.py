"""Oracle: encoder stack (LayerNorm, multi-head attention, position-wise FFN, pre-norm
residual layers) as pure functions over a ``{state_dict_name: tensor}`` mapping.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Each function names the reference lines
it restates (paths relative to /root/reference/).  Everything is written dtype-generic so a
test can run it in fp64 to obtain tight reference gradients through torch autograd.

Eval-mode by default (every nn.Dropout of the reference is the identity).  torch's dropout
stream cannot be reproduced by the HIP path, so for train mode each function takes the
dropout MULTIPLIERS explicitly (mask * 1/(1-p), shaped like the tensor nn.Dropout is applied
to): a test extracts the masks the kernels used (mmt_debug_dropout_mask) and replays the
reference arithmetic with them.
"""
import math
import re

import torch


def layer_norm(x, a_2, b_2, eps=1e-6):
    """transformer/MFT/multiTransformer.py:88-91.

    NOT torch.nn.LayerNorm: the spread is the *unbiased* standard deviation (divide by
    d-1) and ``eps`` is added to the standard deviation, outside the square root.
    """
    d = x.shape[-1]
    mu = x.sum(dim=-1, keepdim=True) / d
    centred = x - mu
    sigma = torch.sqrt((centred * centred).sum(dim=-1, keepdim=True) / (d - 1))
    return a_2 * centred / (sigma + eps) + b_2


def scaled_dot_attention(q, k, v, row_mask=None, prob_drop=None):
    """transformer/MFT/multiTransformer.py:22-34; ``prob_drop`` = multiplier of nn.Dropout on p_attn (:32-33).

    q, k, v: (B, h, T, d_k).  ``row_mask`` is the reference's (B, T, 1) float mask after
    ``unsqueeze(1)`` -> (B, 1, T, 1): it broadcasts along the KEY axis, so a zero entry
    blanks an entire QUERY row with -1e9 (keys are never masked).  A blanked row soft-maxes
    to exactly 1/T per key.
    """
    d_k = q.shape[-1]
    scores = (q @ k.transpose(-2, -1)) / math.sqrt(d_k)
    if row_mask is not None:
        blank = (row_mask == 0)
        scores = torch.where(blank, torch.full_like(scores, -1e9), scores)
    probs = torch.softmax(scores, dim=-1)
    if prob_drop is not None:
        probs = probs * prob_drop
    return probs @ v, probs


def _affine(p, name, x):
    return x @ p[name + ".weight"].transpose(0, 1) + p[name + ".bias"]


def multi_head_attention(p, prefix, query, key, value, mask, h, prob_drop=None):
    """transformer/MFT/multiTransformer.py:47-65.  ``prefix`` ends before ``linears``."""
    B, d = query.shape[0], query.shape[-1]
    d_k = d // h
    row_mask = None if mask is None else mask.unsqueeze(1)

    def split(z):
        return z.reshape(B, -1, h, d_k).permute(0, 2, 1, 3)

    q = split(_affine(p, prefix + "linears.0", query))
    k = split(_affine(p, prefix + "linears.1", key))
    v = split(_affine(p, prefix + "linears.2", value))
    ctx, _ = scaled_dot_attention(q, k, v, row_mask, prob_drop)
    merged = ctx.permute(0, 2, 1, 3).reshape(B, -1, h * d_k)
    return _affine(p, prefix + "linears.3", merged)


def feed_forward(p, prefix, x, hidden_drop=None):
    """transformer/MFT/multiTransformer.py:19-20; ``hidden_drop`` = multiplier of the dropout after the ReLU."""
    hid = torch.relu(_affine(p, prefix + "w_1", x))
    if hidden_drop is not None:
        hid = hid * hidden_drop
    return _affine(p, prefix + "w_2", hid)


def encoder_layer(p, prefix, x, mask, h, drops=None):
    """transformer/MFT/multiTransformer.py:103-104 and :114-116 (pre-norm residual).
    ``drops``: None (eval) or {"attn": (B,h,T,T), "sub0": (B,T,d), "ffn": (B,T,f), "sub1": (B,T,d)} multipliers."""
    dr = drops or {}
    n0 = layer_norm(x, p[prefix + "sublayer.0.norm.a_2"], p[prefix + "sublayer.0.norm.b_2"])
    a = multi_head_attention(p, prefix + "self_attn.", n0, n0, n0, mask, h, dr.get("attn"))
    x = x + (a * dr["sub0"] if "sub0" in dr else a)
    n1 = layer_norm(x, p[prefix + "sublayer.1.norm.a_2"], p[prefix + "sublayer.1.norm.b_2"])
    f = feed_forward(p, prefix + "feed_forward.", n1, dr.get("ffn"))
    return x + (f * dr["sub1"] if "sub1" in dr else f)


def count_layers(p, prefix):
    pat = re.compile(re.escape(prefix) + r"layers\.(\d+)\.")
    idx = {int(m.group(1)) for m in (pat.match(k) for k in p) if m}
    return max(idx) + 1 if idx else 0


def encoder_stack(p, prefix, x, mask, h, drops=None):
    """transformer/MFT/multiTransformer.py:73-76: N layers, then the final LayerNorm.  ``drops``: per-layer list."""
    for i in range(count_layers(p, prefix)):
        x = encoder_layer(p, "%slayers.%d." % (prefix, i), x, mask, h, None if drops is None else drops[i])
    return layer_norm(x, p[prefix + "norm.a_2"], p[prefix + "norm.b_2"])
