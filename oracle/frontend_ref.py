"""Oracle: the window-encoder front-end (CNN over the tokens/frames of a window, global max-pool, Highway)
and the whole `MultiCNNTransformer` models that own the sequence models — pure functions over a
``{state_dict_name: tensor}`` mapping.

TEST INFRASTRUCTURE — see ``oracle/__init__.py``.  Paths relative to /root/reference/.  Eval mode (the
Dropout(0.3) of transformer/SFT/models.py:103,130 is the identity) unless a multiplier is passed.
"""
import torch

from .models_ref import multi_transformer, nlp_transformer, uni_full_transformer, lstm_decoder_head
from .encoder_ref import encoder_stack


def cnn_maxpool(x, weight, bias):
    """transformer/SFT/models.py:57-79: Conv1d(D -> F, kernel k, bias) along the W positions of each window, then
    MaxPool1d over ALL output positions (the pool length equals the conv output length, so its stride is moot).
    No activation in between.  x: (N, W, D);  weight: (F, D, k);  -> (N, F), argmax positions (N, F)."""
    N, W, D = x.shape
    F_, _, k = weight.shape
    L = W - k + 1
    # conv as a sum over taps of (N, L, D) @ (D, F)
    y = bias.view(1, 1, F_).expand(N, L, F_)
    for j in range(k):
        y = y + x[:, j:j + L, :] @ weight[:, :, j].transpose(0, 1)
    out, arg = y.max(dim=1)
    return out, arg


def highway(p, prefix, x):
    """transformer/SFT/models.py:27-55: gate * proj + (1 - gate) * x with proj LINEAR (no ReLU in the reference)."""
    proj = x @ p[prefix + "linear_projection.weight"].transpose(0, 1) + p[prefix + "linear_projection.bias"]
    gate = torch.sigmoid(x @ p[prefix + "linear_gate.weight"].transpose(0, 1) + p[prefix + "linear_gate.bias"])
    return gate * proj + (1.0 - gate) * x


def window_encoder(p, mod, x, drop=None):
    """One modality of MultiCNNTransformer.forward (transformer/SFT/models.py:118-132): the reference loops over the
    batch; the windows are independent, so (B,T,W,D) is simply flattened to B*T windows.  -> (B,T,F)."""
    B, T, W, D = x.shape
    out, _ = cnn_maxpool(x.reshape(B * T, W, D), p["cnn_%s.conv1d.weight" % mod], p["cnn_%s.conv1d.bias" % mod])
    hw = highway(p, "highway_%s." % mod, out)
    if drop is not None:
        hw = hw * drop
    return hw.reshape(B, T, -1)


def _sub(p, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def multi_cnn_transformer_sft(p, mods, inputs, mask, h=8):
    """transformer/SFT/models.py:113-142: per-modality window encoders, concat, tanh(fusionLayer), NLPTransformer
    (one modality: UniTransformer on the single encoder's output, :139-140)."""
    outs = [window_encoder(p, m, inputs[m]) for m in mods]
    tp = _sub(p, "Transformer.")
    if len(outs) > 1:
        cat = torch.cat(outs, dim=2)
        fused = torch.tanh(cat @ p["fusionLayer.weight"].transpose(0, 1) + p["fusionLayer.bias"])
        return nlp_transformer(tp, fused, mask, h)
    e = outs[0] @ tp["embed.weight"].transpose(0, 1) + tp["embed.bias"]
    enc = encoder_stack(tp, "encoder.", e, mask, h)
    return lstm_decoder_head(tp, enc) * mask.to(e.dtype)


def multi_cnn_transformer_mft(p, mods, inputs, mask, h=8):
    """transformer/MFT/models.py:111-138: per-modality window encoders feed MultiTransformer as a dict."""
    outs = {m: window_encoder(p, m, inputs[m]) for m in mods}
    return multi_transformer(_sub(p, "Transformer."), outs, mask, mods, h)


def multi_cnn_transformer_b2(p, mods, inputs, mask, h=8):
    """transformer/B2-Trans/models.py:81-134: one modality, window encoder -> UniFullTransformer."""
    out = window_encoder(p, mods[0], inputs[mods[0]])
    return uni_full_transformer(_sub(p, "Transformer."), out, mask, h)
