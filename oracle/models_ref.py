"""Oracle: the three sequence models that call the encoder stack (TEST INFRASTRUCTURE —
see oracle/__init__.py).  Eval-mode restatements keyed by the reference's state_dict names.
"""
import torch

from .encoder_ref import encoder_stack
from .mfn_ref import lstm_cell, mfn_gate


def _fc(p, name, x):
    return x @ p[name + ".weight"].transpose(0, 1) + p[name + ".bias"]


def multi_transformer(p, inputs, mask, mods, h=8, prefix=""):
    """MFT fuse point — transformer/MFT/multiTransformer.py:288-313.

    inputs: {mod: (B, T, e_mod)}.  Per modality: Linear embed (:296) -> its own encoder
    stack (:299) -> time-major view (:300); then the MFN gate (:303) and the mask multiply (:310).
    The registered-but-unused ``attn{mod}``/``ff{mod}`` parameters (:275-276) take no part.
    """
    gate_in = {}
    for mod in mods:
        e = _fc(p, "%sembed_%s" % (prefix, mod), inputs[mod])
        e = encoder_stack(p, "%stransformer_%s." % (prefix, mod), e, mask, h)
        gate_in[mod] = e.permute(1, 0, 2)
    return mfn_gate(p, prefix + "mfn.", gate_in, mods) * mask.to(inputs[mods[0]].dtype)


def lstm_decoder_head(p, enc, prefix=""):
    """Autoregressive one-layer LSTM decoder + MLP of the SFT model —
    transformer/SFT/multiTransformer.py:463-482.

    Step t consumes ``[o_{t-1} ; enc[:, t]]`` (:473) where o is the LSTM output (= h for one
    layer); initial (h, c) = learned ``dec_h0``/``dec_c0`` broadcast over the batch (:465-466),
    o_{-1} = 0 (:469).  Returns (B, T, 1) before the mask multiply.
    """
    B, T, d = enc.shape
    w = [p[prefix + "decoder." + n] for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    h = p[prefix + "dec_h0"][0].expand(B, d)
    c = p[prefix + "dec_c0"][0].expand(B, d)
    o_prev = torch.zeros(B, d, dtype=enc.dtype, device=enc.device)
    outs = []
    for t in range(T):
        h, c = lstm_cell(torch.cat([o_prev, enc[:, t]], dim=-1), h, c, *w)
        o_prev = h
        outs.append(h)
    o_all = torch.stack(outs, dim=1)                                       # (B,T,d)
    return _fc(p, prefix + "out.2", torch.relu(_fc(p, prefix + "out.0", o_all)))


def nlp_transformer(p, x, mask, h=8, prefix=""):
    """SFT sequence model — transformer/SFT/multiTransformer.py:457-484 (eval: the embed
    Dropout(0.1) at :431 is the identity)."""
    e = torch.relu(_fc(p, prefix + "embed.1", x))
    enc = encoder_stack(p, prefix + "encoder.", e, mask, h)
    return lstm_decoder_head(p, enc, prefix) * mask.to(x.dtype)


def uni_full_transformer(p, x, mask, h=8, prefix=""):
    """B2-Trans sequence model — transformer/B2-Trans/multiTransformer.py:408-420."""
    enc = encoder_stack(p, prefix + "encoder.", _fc(p, prefix + "embed", x), mask, h)
    out = _fc(p, prefix + "out.2", torch.relu(_fc(p, prefix + "out.0", enc)))
    return out * mask.to(x.dtype)
