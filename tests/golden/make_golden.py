#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE implementation (build container only).

Run:  MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's classes are imported from /root/reference (read-only, never copied), filled
with the deterministic weights of ``recipe.py``, run in eval mode on CPU in fp32, and their
outputs / gradients are written as small fixtures.  The fixtures are data only; nothing of
the reference's source travels.  Each variant directory of the reference defines modules
with the same names, so each is loaded under a private module name.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe as R  # noqa: E402

REF = "/root/reference/transformer"
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True


def load_ref(variant):
    spec = importlib.util.spec_from_file_location("ref_%s" % variant.replace("-", "_"),
                                                  os.path.join(REF, variant, "multiTransformer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def fill(module, seed=R.SEED):
    params = R.gen_params(R.shapes_of(module.state_dict()), seed)
    module.load_state_dict(params)
    module.eval()
    return params


def grads_of(module):
    return {n: (None if p.grad is None else R.to_np(p.grad)) for n, p in module.named_parameters()}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def main():
    torch.manual_seed(1)
    torch.set_num_threads(4)
    mft = load_ref("MFT")
    sft = load_ref("SFT")
    b2 = load_ref("B2-Trans")
    cpu = torch.device("cpu")

    # ---- LayerNorm alone -----------------------------------------------------------
    ln = mft.LayerNorm(128)
    w = fill(ln)
    x = R.gen_normal("ln:x", (4, 50, 128), R.SEED).requires_grad_()
    g = R.gen_normal("ln:g", (4, 50, 128), R.SEED)
    y = ln(x)
    (y * g).sum().backward()
    save("ln_d128", out=R.to_np(y), dx=R.to_np(x.grad), checksum=R.weights_checksum(w),
         **{"grad:" + k: v for k, v in grads_of(ln).items()})

    # ---- MultiHeadedAttention alone ---------------------------------------------------
    mha = mft.MultiHeadedAttention(8, 128)
    w = fill(mha)
    lengths = [50, 40, 30, 7]
    mask = R.prefix_mask(lengths, 50)
    x = R.gen_normal("mha:x", (4, 50, 128), R.SEED).requires_grad_()
    g = R.gen_normal("mha:g", (4, 50, 128), R.SEED)
    y = mha(x, x, x, mask)
    (y * g).sum().backward()
    save("mha_d128_h8", out=R.to_np(y), dx=R.to_np(x.grad), checksum=R.weights_checksum(w),
         lengths=np.array(lengths), **{"grad:" + k: v for k, v in grads_of(mha).items()})

    # ---- PositionwiseFeedForward alone --------------------------------------------------
    ffn = mft.PositionwiseFeedForward(128, R.D_FF, 0.1)
    w = fill(ffn)
    x = R.gen_normal("ffn:x", (4, 50, 128), R.SEED).requires_grad_()
    g = R.gen_normal("ffn:g", (4, 50, 128), R.SEED)
    y = ffn(x)
    (y * g).sum().backward()
    save("ffn_d128", out=R.to_np(y), dx=R.to_np(x.grad), checksum=R.weights_checksum(w),
         **{"grad:" + k: v for k, v in grads_of(ffn).items()})

    # ---- Encoder stacks ----------------------------------------------------------------
    for name, d, h, n, B, T, lengths in R.ENCODER_CASES:
        enc = mft.Encoder(mft.EncoderLayer(d, mft.MultiHeadedAttention(h, d),
                                           mft.PositionwiseFeedForward(d, R.D_FF, 0.1), 0.1), n)
        w = fill(enc)
        mask = R.prefix_mask(lengths, T)
        x = R.gen_normal(name + ":x", (B, T, d), R.SEED).requires_grad_()
        g = R.gen_normal(name + ":g", (B, T, d), R.SEED)
        y = enc(x, mask)
        (y * g).sum().backward()
        save(name, out=R.to_np(y), dx=R.to_np(x.grad), checksum=R.weights_checksum(w),
             lengths=np.array(lengths), **{"grad:" + k: v for k, v in grads_of(enc).items()})

    # ---- MFN gate alone -------------------------------------------------------------------
    mods = R.MODS_AVL
    mfn = mft.MFN(mods, {m: 256 for m in mods}, 1, device=cpu)
    w = fill(mfn)
    T, B = 20, 3
    ins = {m: R.gen_normal("mfn:" + m, (T, B, 256), R.SEED).requires_grad_() for m in mods}
    g = R.gen_normal("mfn:g", (B, T, 1), R.SEED)
    y = mfn(ins)
    (y * g).sum().backward()
    save("mfn_avl", out=R.to_np(y), checksum=R.weights_checksum(w),
         **{"dx:" + m: R.to_np(ins[m].grad) for m in mods},
         **{"grad:" + k: v for k, v in grads_of(mfn).items()})

    # ---- whole sequence models (valence outputs + loss + gradient norms) ---------------------
    def model_case(name, model, inputs, lengths, T, extra_full=()):
        w = fill(model)
        mask = R.prefix_mask(lengths, T)
        target = R.gen_uniform(name + ":target", (len(lengths), T, 1), R.SEED) * mask
        out = model(inputs, mask, lengths)
        loss = ((out - target) ** 2).sum() / float(sum(lengths))
        loss.backward()
        gr = grads_of(model)
        arrays = dict(out=R.to_np(out), loss=np.float64(loss.item()), checksum=R.weights_checksum(w),
                      lengths=np.array(lengths))
        for k, v in gr.items():
            arrays["gnorm:" + k] = np.float64(-1.0 if v is None else np.sqrt((v.astype(np.float64) ** 2).sum()))
        for k in extra_full:
            arrays["grad:" + k] = gr[k]
        save(name, **arrays)

    lengths = [50, 40, 30, 7]
    m = mft.MultiTransformer(mods, R.EMBED_AVL, device=cpu)
    ins = {md: R.gen_normal("model_mft:" + md, (4, 50, R.EMBED_AVL[md]), R.SEED) for md in mods}
    model_case("model_mft_avl", m, ins, lengths, 50,
               extra_full=("mfn.out_fc2.weight", "embed_acoustic.bias", "transformer_image.norm.a_2",
                           "transformer_linguistic.layers.0.sublayer.0.norm.b_2"))

    for name, mods_s, embed_s in R.MFT_SWEEP:                              # transformer/MFT/train.py:538-552
        m = mft.MultiTransformer(mods_s, embed_s, device=cpu)
        ins = {md: R.gen_normal(name + ":" + md, (4, 50, embed_s[md]), R.SEED) for md in mods_s}
        model_case(name, m, ins, lengths, 50,
                   extra_full=("mfn.out_fc2.weight", "mfn.att1_fc1.weight", "mfn.gamma1_fc1.weight", "embed_acoustic.weight"))

    for name, kw in (("model_sft_d128", dict(embed_dim=128, h=8)), ("model_sft_d40", dict(embed_dim=40, h=4)),
                     ("model_sft_default", dict())):
        m = sft.NLPTransformer(512, device=cpu, **kw)
        x = torch.tanh(R.gen_normal(name + ":x", (4, 50, 512), R.SEED))   # post-fusion tanh, SFT/models.py:138
        model_case(name, m, x, lengths, 50,
                   extra_full=("out.2.weight", "embed.1.bias", "encoder.norm.a_2", "dec_h0"))

    m = b2.UniFullTransformer(300, device=cpu)
    x = R.gen_normal("model_b2:x", (4, 50, 300), R.SEED)
    model_case("model_b2_text", m, x, lengths, 50, extra_full=("out.2.weight", "embed.bias", "encoder.norm.b_2"))

    # ---- eval_ccc (numpy formula of SFT/train.py:42-50, restated inline: train.py is not importable
    #      without the SEND dataset paths, so this is the one fixture not produced by reference code)
    a = R.gen_uniform("ccc:a", (125,), R.SEED).numpy().astype(np.float64)
    b = (0.7 * a + 0.3 * R.gen_uniform("ccc:b", (125,), R.SEED).numpy()).astype(np.float64)
    covar = np.cov(a, b, bias=True)[0][1]
    ccc = 2 * covar / (np.var(a) + np.var(b) + (np.mean(b) - np.mean(a)) ** 2)
    save("ccc", a=a, b=b, ccc=np.float64(ccc))


if __name__ == "__main__":
    main()
