"""Deterministic recipe for golden-test weights and inputs.

Shared by ``make_golden.py`` (which loads the values into the *reference* classes to
capture expected outputs) and by the tests (which load the same values into the oracle and
into the HIP-backed modules).  Fixtures therefore hold only inputs-by-recipe, expected
outputs/gradients and a checksum of the regenerated weights — never weight blobs.

Values come from torch's CPU mt19937 generator seeded per tensor *name*, so they do not
depend on parameter iteration order.  ``weights_checksum`` guards against a generator
change between the container that wrote a fixture and the one replaying it.
"""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch

MODS_AVL = ["acoustic", "image", "linguistic"]          # transformer/MFT/train.py:541-548 order
EMBED_AVL = {"acoustic": 88, "image": 256, "linguistic": 300}  # transformer/MFT/train.py:552
# the other models of the reference's MFT sweep (transformer/MFT/train.py:538-552: VA / AL / VAL x acoustic embed 88 / 44):
# two-modality gates (2H = 272: see oracle.mfn_ref.HIDDEN) and the 44-wide acoustic embed run through other segment
# tables and row-GEMM paddings than VAL-88.  (fixture name, modalities in the sweep's order, window_embed_size)
MFT_SWEEP = [
    ("model_mft_va88", ["acoustic", "image"], {"acoustic": 88, "image": 256}),
    ("model_mft_al88", ["acoustic", "linguistic"], {"acoustic": 88, "linguistic": 300}),
    ("model_mft_val44", ["acoustic", "image", "linguistic"], {"acoustic": 44, "image": 256, "linguistic": 300}),
]


def _gen(tag, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(tag.encode())) % (2 ** 63 - 1))
    return g


def gen_tensor(name, shape, seed):
    """One parameter tensor (fp32) from its state_dict name and shape."""
    g = _gen(name, seed)
    shape = tuple(shape)
    if name.endswith("a_2"):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if name.endswith("b_2") or "dec_h0" in name or "dec_c0" in name:
        return 0.1 * torch.randn(shape, generator=g)
    if len(shape) == 1:                                   # biases
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.1
    fan_in = shape[-1] if len(shape) == 2 else int(math.prod(shape[1:]))      # conv weights (F, D, k): D*k inputs per output
    gain = 3.0 if (".linears.0." in name or ".linears.1." in name) else 1.0   # peaky attention
    bound = gain / math.sqrt(fan_in)
    return (torch.rand(shape, generator=g) * 2 - 1) * bound


def gen_params(shapes, seed):
    """shapes: mapping name -> shape (e.g. from ``module.state_dict()``)."""
    return OrderedDict((n, gen_tensor(n, tuple(s), seed)) for n, s in shapes.items())


def shapes_of(state_dict):
    return OrderedDict((k, tuple(v.shape)) for k, v in state_dict.items())


def weights_checksum(params):
    """float64 sum of |w| over all tensors, in sorted-name order."""
    return float(sum(params[k].double().abs().sum().item() for k in sorted(params)))


def gen_normal(tag, shape, seed):
    return torch.randn(tuple(shape), generator=_gen("input:" + tag, seed))


def gen_uniform(tag, shape, seed):
    return torch.rand(tuple(shape), generator=_gen("input:" + tag, seed))


def prefix_mask(lengths, T):
    """(B, T, 1) float mask of prefix ones — transformer/SFT/train.py:101-104."""
    m = torch.zeros(len(lengths), T, 1)
    for i, n in enumerate(lengths):
        m[i, :n] = 1.0
    return m


# ----------------------------------------------------------------------------- cases
# Encoder-level cases: (name, d_model, heads, N layers, B, T, lengths)
ENCODER_CASES = [
    ("enc_d40_h4_n6",   40, 4, 6, 4, 50, [50, 40, 30, 7]),
    ("enc_d128_h8_n2", 128, 8, 2, 4, 50, [50, 40, 30, 7]),
    ("enc_d256_h8_n2", 256, 8, 2, 4, 50, [50, 40, 30, 7]),
    ("enc_d128_h8_n1_t500", 128, 8, 1, 2, 500, [500, 350]),
    ("enc_d128_h8_n2_full", 128, 8, 2, 3, 64, [64, 64, 64]),           # no padding at all
    ("enc_d128_h8_n2_padded", 128, 8, 2, 4, 40, [40, 3, 2, 1]),        # heavily padded
]
D_FF = 128
SEED = 1


def to_np(t):
    return t.detach().cpu().numpy().astype(np.float32)


# ---- window-encoder front-end cases (make_golden_frontend.py) ------------------------------------------
# (name, raw feature size D, window embed size F, positions per window W, windows N)
CNN_CASES = [
    ("fe_cnn_small", 40, 64, 7, 10),
    ("fe_cnn_audio", 88, 256, 10, 6),         # transformer/SFT/train.py:534 dims, models.py:90 embed sizes
    ("fe_cnn_text", 300, 300, 33, 4),
    ("fe_cnn_image", 1000, 256, 30, 3),
    ("fe_cnn_tall", 24, 32, 70, 3),           # more than 32 conv positions per window
]
FE_DIMS = {"linguistic": 300, "emotient": 20, "acoustic": 88, "image": 1000}
FE_WINDOW = {"linguistic": 33, "emotient": 9, "acoustic": 10, "image": 30}
FE_EMBED_MFT = {"linguistic": 300, "emotient": 20, "acoustic": 88, "image": 256}     # transformer/MFT/train.py:552 (A_dim = 88)
