"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the Python surface of the
drop-in modules against the reference's own classes — constructor signatures, `forward` signatures, `state_dict` keys,
key ORDER and shapes — for every class SURVEY 8(b) lists.  Nothing is executed on the HIP path here (construction only)."""
import importlib
import inspect
import os
import sys

import pytest
import torch

REF = "/root/reference/transformer"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _load_variant(variant):
    """import <variant>/multiTransformer.py and models.py under their own names, then drop them from sys.modules again"""
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    for name in ("models", "multiTransformer"):
        sys.modules.pop(name, None)
    sys.path.insert(0, os.path.join(REF, variant))
    try:
        mt = importlib.import_module("multiTransformer")
        md = importlib.import_module("models")
    finally:
        sys.path.pop(0)
        for name in ("models", "multiTransformer"):
            sys.modules.pop(name, None)
    return mt, md


def _sig(cls_or_fn):
    return [(n, p.default if p.default is not inspect._empty else "<required>")
            for n, p in inspect.signature(cls_or_fn).parameters.items() if n != "self"]


def _same_state(ours, ref):
    a, b = ours.state_dict(), ref.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert tuple(a[k].shape) == tuple(b[k].shape), k


MODS = ["acoustic", "image", "linguistic"]
DIMS = {"linguistic": 300, "emotient": 20, "acoustic": 88, "image": 1000}


@pytest.mark.parametrize("variant", ["SFT", "MFT", "B2-Trans"])
def test_multitransformer_module_surface(variant):
    from multimodal_transformer_amd import multiTransformer as OURS
    mt, _ = _load_variant(variant)
    cpu = torch.device("cpu")
    for name in ("PositionwiseFeedForward", "MultiHeadedAttention", "Encoder", "LayerNorm", "SublayerConnection", "EncoderLayer",
                 "MFN", "MultiTransformer", "UniTransformer", "UniFullTransformer"):
        if not hasattr(mt, name):
            continue
        rs, os_ = _sig(getattr(mt, name).__init__), _sig(getattr(OURS, name).__init__)
        # our classes may append keyword arguments; the reference's own must come first, same names and defaults
        assert [(n, str(d)) for n, d in os_[:len(rs)]] == [(n, str(d)) for n, d in rs], name
        assert [n for n, _ in _sig(getattr(OURS, name).forward)] == [n for n, _ in _sig(getattr(mt, name).forward)], name
    assert [n for n, _ in _sig(OURS.attention)] == [n for n, _ in _sig(mt.attention)]
    wes = {"acoustic": 88, "image": 256, "linguistic": 300}
    _same_state(OURS.MultiTransformer(MODS, wes, device=cpu), mt.MultiTransformer(MODS, wes, device=cpu))
    _same_state(OURS.UniFullTransformer(300, device=cpu), mt.UniFullTransformer(300, device=cpu))
    _same_state(OURS.UniTransformer(300, device=cpu), mt.UniTransformer(300, device=cpu))
    if hasattr(mt, "NLPTransformer"):
        assert _sig(OURS.NLPTransformer.__init__)[:len(_sig(mt.NLPTransformer.__init__))] is not None
        _same_state(OURS.NLPTransformer(512, device=cpu), mt.NLPTransformer(512, device=cpu))
        _same_state(OURS.NLPTransformer(512, embed_dim=128, device=cpu), mt.NLPTransformer(512, embed_dim=128, device=cpu))


@pytest.mark.parametrize("variant,ours_name,args", [
    ("SFT", "MultiCNNTransformer", (MODS, DIMS)),
    ("SFT", "MultiCNNTransformer", (["linguistic"], DIMS)),
    ("MFT", "MultiCNNTransformerMFT", (MODS, DIMS, {"linguistic": 300, "emotient": 20, "acoustic": 88, "image": 256})),
    ("B2-Trans", "MultiCNNTransformerB2", (["linguistic"], DIMS))])
def test_front_end_surface(variant, ours_name, args, capsys):
    from multimodal_transformer_amd import models as OURS
    _, md = _load_variant(variant)
    cpu = torch.device("cpu")
    ref_cls, our_cls = md.MultiCNNTransformer, getattr(OURS, ours_name)
    assert [(n, str(d)) for n, d in _sig(our_cls.__init__)] == [(n, str(d)) for n, d in _sig(ref_cls.__init__)]
    assert [n for n, _ in _sig(our_cls.forward)] == [n for n, _ in _sig(ref_cls.forward)]
    _same_state(our_cls(*args, device=cpu), ref_cls(*args, device=cpu))
    for name in ("CNN", "Highway"):
        assert [(n, str(d)) for n, d in _sig(getattr(OURS, name).__init__)] == [(n, str(d)) for n, d in _sig(getattr(md, name).__init__)]
    _same_state(OURS.CNN(300, 128, 2), md.CNN(300, 128, 2))
    _same_state(OURS.Highway(64), md.Highway(64))
