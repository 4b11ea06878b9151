#!/usr/bin/env python3
"""Generate the window-encoder fixtures (tests/golden/fe_*.npz) from the REFERENCE implementation
(build container only).

Run:  MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_frontend.py

Same rules as make_golden.py: the reference's classes (transformer/<variant>/models.py and the multiTransformer.py
it imports) are imported from /root/reference, filled with recipe.py's deterministic weights, run in eval mode on
CPU in fp32; only inputs-by-recipe and expected outputs / gradients are stored.
"""
import importlib
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


def load_models(variant):
    """import <variant>/models.py (it does `from multiTransformer import ...`, so the variant directory must lead sys.path
    and stale same-named modules of another variant must be dropped first)."""
    for name in ("models", "multiTransformer"):
        sys.modules.pop(name, None)
    sys.path.insert(0, os.path.join(REF, variant))
    try:
        mod = importlib.import_module("models")
    finally:
        sys.path.pop(0)
    for name in ("models", "multiTransformer"):
        sys.modules.pop(name, None)
    return mod


def fill(module, seed):
    params = R.gen_params(R.shapes_of(module.state_dict()), seed)
    module.load_state_dict(params)
    module.eval()
    return params


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def main():
    torch.manual_seed(1)
    torch.set_num_threads(4)
    cpu = torch.device("cpu")
    sft = load_models("SFT")

    # ---- CNN (conv k=2 + global max-pool) alone -----------------------------------------------
    for name, D, F, W, N in R.CNN_CASES:
        cnn = sft.CNN(D, F, 2)
        w = fill(cnn, R.SEED)
        x = R.gen_normal(name + ":x", (N, W, D), R.SEED)
        g = R.gen_normal(name + ":g", (N, F), R.SEED)
        y = cnn(x.permute(0, 2, 1))
        (y * g).sum().backward()
        gw = R.to_np(cnn.conv1d.weight.grad)
        save(name, out=R.to_np(y), checksum=R.weights_checksum(w), gb=R.to_np(cnn.conv1d.bias.grad),
             gw_norm=np.float64(np.sqrt((gw.astype(np.float64) ** 2).sum())), gw_head=gw[:, :8, :].copy(),
             gw_tail=gw[:, -8:, :].copy())

    # ---- Highway alone -------------------------------------------------------------------------
    hw = sft.Highway(256)
    w = fill(hw, R.SEED)
    x = R.gen_normal("fe_highway:x", (12, 256), R.SEED).requires_grad_()
    g = R.gen_normal("fe_highway:g", (12, 256), R.SEED)
    y = hw(x)
    (y * g).sum().backward()
    save("fe_highway", out=R.to_np(y), dx=R.to_np(x.grad), checksum=R.weights_checksum(w),
         **{"grad:" + k: R.to_np(p.grad) for k, p in hw.named_parameters()})

    # ---- whole models: raw windows -> valence ----------------------------------------------------
    def model_case(name, model, mods, dims, lengths, T, extra_full):
        w = fill(model, R.SEED)
        B = len(lengths)
        mask = R.prefix_mask(lengths, T)
        inputs = {m: R.gen_normal("%s:%s" % (name, m), (B, T, R.FE_WINDOW[m], dims[m]), R.SEED) for m in mods}
        target = R.gen_uniform(name + ":target", (B, T, 1), R.SEED) * mask
        out = model(inputs, lengths, mask)
        loss = ((out - target) ** 2).sum() / float(sum(lengths))
        loss.backward()
        arrays = dict(out=R.to_np(out), loss=np.float64(loss.item()), checksum=R.weights_checksum(w), lengths=np.array(lengths))
        for k, p in model.named_parameters():
            arrays["gnorm:" + k] = np.float64(-1.0 if p.grad is None else float(p.grad.double().pow(2).sum().sqrt()))
        for k in extra_full:
            arrays["grad:" + k] = R.to_np(dict(model.named_parameters())[k].grad)
        save(name, **arrays)

    mods = R.MODS_AVL
    lengths = [6, 4]
    model_case("fe_model_sft", sft.MultiCNNTransformer(mods, R.FE_DIMS, device=cpu), mods, R.FE_DIMS, lengths, 6,
               ("fusionLayer.bias", "cnn_acoustic.conv1d.bias", "highway_image.linear_gate.bias", "Transformer.out.2.weight"))
    mftm = load_models("MFT")
    model_case("fe_model_mft", mftm.MultiCNNTransformer(mods, R.FE_DIMS, R.FE_EMBED_MFT, device=cpu), mods, R.FE_DIMS, lengths, 6,
               ("cnn_linguistic.conv1d.bias", "highway_acoustic.linear_projection.bias", "Transformer.mfn.out_fc2.weight"))
    b2m = load_models("B2-Trans")
    model_case("fe_model_b2", b2m.MultiCNNTransformer(["linguistic"], R.FE_DIMS, device=cpu), ["linguistic"], R.FE_DIMS, lengths, 6,
               ("cnn_linguistic.conv1d.bias", "Transformer.out.2.weight"))


if __name__ == "__main__":
    main()
