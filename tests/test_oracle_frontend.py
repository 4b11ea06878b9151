"""CPU: the oracle's window-encoder front-end (oracle/frontend_ref.py) against the fixtures captured from the
reference's models.py (tests/golden/make_golden_frontend.py -> fe_*.npz).  State-dict shapes are written out
by hand from the reference constructors (transformer/SFT/models.py:57-111, MFT/models.py:81-108,
B2-Trans/models.py:81-103) so that the test does not depend on the product modules."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import oracle
import recipe as R
from conftest import load_golden, rel_l2, grad_close
from test_oracle_golden import _model_shapes, _params, _grad_scale, ATOL, GRAD_RTOL

WES_SFT = {"linguistic": 300, "emotient": 20, "acoustic": 256, "image": 256}       # transformer/SFT/models.py:90


def frontend_shapes(mods, wes, kind):
    s = OrderedDict()
    for m in mods:
        s["cnn_%s.conv1d.weight" % m] = (wes[m], R.FE_DIMS[m], 2)
        s["cnn_%s.conv1d.bias" % m] = (wes[m],)
        for n in ("linear_projection", "linear_gate"):
            s["highway_%s.%s.weight" % (m, n)] = (wes[m], wes[m])
            s["highway_%s.%s.bias" % (m, n)] = (wes[m],)
    if kind == "sft":
        s["fusionLayer.weight"] = (512, sum(wes[m] for m in mods))
        s["fusionLayer.bias"] = (512,)
        inner = _model_shapes("sft", d=256)
    elif kind == "mft":
        inner = _model_shapes("mft")
    else:
        inner = _model_shapes("b2")
    for k, v in inner.items():
        s["Transformer." + k] = v
    return s


@pytest.mark.parametrize("case", R.CNN_CASES, ids=[c[0] for c in R.CNN_CASES])
def test_cnn_maxpool(case):
    name, D, F, W, N = case
    fx = load_golden(name)
    p = _params(OrderedDict([("conv1d.weight", (F, D, 2)), ("conv1d.bias", (F,))]), fx)
    x = R.gen_normal(name + ":x", (N, W, D), R.SEED)
    g = R.gen_normal(name + ":g", (N, F), R.SEED)
    out, arg = oracle.cnn_maxpool(x, p["conv1d.weight"], p["conv1d.bias"])
    (out * g).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), fx["out"], atol=ATOL, rtol=3e-6)     # |out| ~ 40 at D = 1000: fp32 round-off
    assert arg.min() >= 0 and arg.max() <= W - 2
    gw = p["conv1d.weight"].grad.numpy()
    assert rel_l2(p["conv1d.bias"].grad.numpy(), fx["gb"]) < GRAD_RTOL
    assert abs(np.sqrt((gw.astype(np.float64) ** 2).sum()) - float(fx["gw_norm"])) < GRAD_RTOL * float(fx["gw_norm"])
    assert rel_l2(gw[:, :8, :], fx["gw_head"]) < GRAD_RTOL and rel_l2(gw[:, -8:, :], fx["gw_tail"]) < GRAD_RTOL


def test_highway():
    fx = load_golden("fe_highway")
    shapes = OrderedDict((n + s, sh) for n in ("linear_projection", "linear_gate") for s, sh in ((".weight", (256, 256)), (".bias", (256,))))
    p = _params(shapes, fx)
    x = R.gen_normal("fe_highway:x", (12, 256), R.SEED).requires_grad_()
    g = R.gen_normal("fe_highway:g", (12, 256), R.SEED)
    y = oracle.highway(p, "", x)
    (y * g).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), fx["out"], atol=ATOL, rtol=0)
    assert rel_l2(x.grad.numpy(), fx["dx"]) < GRAD_RTOL
    for k in fx:
        if k.startswith("grad:"):
            assert grad_close(p[k[5:]].grad.numpy(), fx[k], GRAD_RTOL, 3e-7 * _grad_scale(fx)), k


@pytest.mark.parametrize("name,kind,mods,fn", [
    ("fe_model_sft", "sft", R.MODS_AVL, oracle.multi_cnn_transformer_sft),
    ("fe_model_mft", "mft", R.MODS_AVL, oracle.multi_cnn_transformer_mft),
    ("fe_model_b2", "b2", ["linguistic"], oracle.multi_cnn_transformer_b2)])
def test_multi_cnn_transformer(name, kind, mods, fn):
    fx = load_golden(name)
    wes = R.FE_EMBED_MFT if kind == "mft" else WES_SFT
    p = _params(frontend_shapes(mods, wes, kind), fx)
    lengths = list(fx["lengths"])
    B, T = len(lengths), 6
    mask = R.prefix_mask(lengths, T)
    inputs = {m: R.gen_normal("%s:%s" % (name, m), (B, T, R.FE_WINDOW[m], R.FE_DIMS[m]), R.SEED) for m in mods}
    out = fn(p, mods, inputs, mask)
    target = R.gen_uniform(name + ":target", (B, T, 1), R.SEED) * mask
    loss = oracle.masked_mse_sum_loss(out, target, lengths)
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), fx["out"], atol=ATOL, rtol=0)
    assert abs(loss.item() - float(fx["loss"])) < 1e-5 * max(1.0, abs(float(fx["loss"])))
    assert (out.detach().numpy()[mask.numpy() == 0] == 0).all()
    gmax = max(float(fx[k]) for k in fx if k.startswith("gnorm:"))      # floor for analytically-zero gradients (key biases)
    for k in fx:
        if k.startswith("gnorm:"):
            g = p[k[6:]].grad
            if float(fx[k]) < 0:
                assert g is None or float(g.abs().sum()) == 0.0, k
            else:
                n = float(g.double().pow(2).sum().sqrt())
                assert abs(n - float(fx[k])) <= 1e-3 * float(fx[k]) + 1e-6 * gmax, k
        if k.startswith("grad:"):
            assert grad_close(p[k[5:]].grad.numpy(), fx[k], 1e-3, 1e-6 * _grad_scale(fx)), k
