"""Pin the CPU oracle against fixtures captured from the reference (tests/golden/make_golden.py).

fp32 oracle vs fp32 reference: the two differ only by summation order, so the bound is a few
ulps of the largest activations: 2e-5 absolute on O(1..10) outputs, 2e-4 relative-L2 on gradients.
"""
import numpy as np
import pytest
import torch

import oracle
import recipe as R
from conftest import load_golden, rel_l2, grad_close

ATOL = 2e-5
GRAD_RTOL = 2e-4


def _params(shapes, fx):
    p = R.gen_params(shapes, R.SEED)
    assert abs(R.weights_checksum(p) - float(fx["checksum"])) <= 1e-6 * float(fx["checksum"]), \
        "weight recipe regenerated different values than the fixture was made with"
    return {k: v.clone().requires_grad_() for k, v in p.items()}


def _shapes_from_grads(fx):
    return {k[5:]: fx[k].shape for k in fx if k.startswith("grad:")}


def _grad_scale(fx):
    """Largest gradient entry in the fixture: the absolute floor for analytically-zero gradients
    (rounding noise scales with the magnitudes that cancel, not with the zero result)."""
    return max([1.0] + [float(np.abs(fx[k]).max()) for k in fx if k.startswith("grad:")])


def _check_grads(p, fx):
    floor = 3e-7 * _grad_scale(fx)
    for k in fx:
        if k.startswith("grad:"):
            got = p[k[5:]].grad
            assert got is not None, k
            assert grad_close(got.numpy(), fx[k], GRAD_RTOL, floor), k


def test_layer_norm():
    fx = load_golden("ln_d128")
    p = _params(_shapes_from_grads(fx), fx)
    x = R.gen_normal("ln:x", (4, 50, 128), R.SEED).requires_grad_()
    g = R.gen_normal("ln:g", (4, 50, 128), R.SEED)
    y = oracle.layer_norm(x, p["a_2"], p["b_2"])
    (y * g).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), fx["out"], atol=ATOL, rtol=0)
    assert rel_l2(x.grad.numpy(), fx["dx"]) < GRAD_RTOL
    _check_grads(p, fx)


def test_multi_head_attention():
    fx = load_golden("mha_d128_h8")
    p = _params(_shapes_from_grads(fx), fx)
    mask = R.prefix_mask(list(fx["lengths"]), 50)
    x = R.gen_normal("mha:x", (4, 50, 128), R.SEED).requires_grad_()
    g = R.gen_normal("mha:g", (4, 50, 128), R.SEED)
    y = oracle.multi_head_attention(p, "", x, x, x, mask, 8)
    (y * g).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), fx["out"], atol=ATOL, rtol=0)
    assert rel_l2(x.grad.numpy(), fx["dx"]) < GRAD_RTOL
    _check_grads(p, fx)


def test_feed_forward():
    fx = load_golden("ffn_d128")
    p = _params(_shapes_from_grads(fx), fx)
    x = R.gen_normal("ffn:x", (4, 50, 128), R.SEED).requires_grad_()
    g = R.gen_normal("ffn:g", (4, 50, 128), R.SEED)
    y = oracle.feed_forward(p, "", x)
    (y * g).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), fx["out"], atol=ATOL, rtol=0)
    assert rel_l2(x.grad.numpy(), fx["dx"]) < GRAD_RTOL
    _check_grads(p, fx)


@pytest.mark.parametrize("case", R.ENCODER_CASES, ids=[c[0] for c in R.ENCODER_CASES])
def test_encoder_stack(case):
    name, d, h, n, B, T, lengths = case
    fx = load_golden(name)
    p = _params(_shapes_from_grads(fx), fx)
    assert oracle.count_layers(p, "") == n
    mask = R.prefix_mask(lengths, T)
    x = R.gen_normal(name + ":x", (B, T, d), R.SEED).requires_grad_()
    g = R.gen_normal(name + ":g", (B, T, d), R.SEED)
    y = oracle.encoder_stack(p, "", x, mask, h)
    (y * g).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), fx["out"], atol=ATOL, rtol=0)
    assert rel_l2(x.grad.numpy(), fx["dx"]) < GRAD_RTOL
    _check_grads(p, fx)


def test_padded_query_rows_are_uniform():
    """Mask semantics (transformer/MFT/multiTransformer.py:29-31,48-50): a blanked QUERY row attends
    uniformly (exactly 1/T) over ALL keys, padded ones included; keys are never masked."""
    q = torch.randn(2, 3, 9, 4)
    k = torch.randn(2, 3, 9, 4)
    v = torch.randn(2, 3, 9, 4)
    mask = R.prefix_mask([9, 5], 9).unsqueeze(1)
    ctx, probs = oracle.scaled_dot_attention(q, k, v, mask)
    assert torch.equal(probs[1, :, 5:], torch.full((3, 4, 9), 1.0 / 9))
    assert (probs[1, :, :5, 5:] > 0).all()                      # valid rows still see padded keys
    np.testing.assert_allclose(ctx[1, :, 5:].numpy(), v[1].mean(dim=1, keepdim=True).expand(3, 4, 4).numpy(),
                               atol=1e-6)


def test_mfn_gate():
    fx = load_golden("mfn_avl")
    p = _params(_shapes_from_grads(fx), fx)
    mods = R.MODS_AVL
    ins = {m: R.gen_normal("mfn:" + m, (20, 3, 256), R.SEED).requires_grad_() for m in mods}
    g = R.gen_normal("mfn:g", (3, 20, 1), R.SEED)
    y = oracle.mfn_gate(p, "", ins, mods)
    (y * g).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), fx["out"], atol=ATOL, rtol=0)
    for m in mods:
        assert rel_l2(ins[m].grad.numpy(), fx["dx:" + m]) < GRAD_RTOL, m
    _check_grads(p, fx)


def _model_shapes(kind, **kw):
    """State-dict shapes of the three sequence models, written out from SURVEY.md §8(b)."""
    from collections import OrderedDict
    s = OrderedDict()

    def linear(name, o, i):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    def encoder(prefix, d, n=6, f=128):
        for i in range(n):
            L = "%slayers.%d." % (prefix, i)
            for j in range(4):
                linear(L + "self_attn.linears.%d" % j, d, d)
            linear(L + "feed_forward.w_1", f, d)
            linear(L + "feed_forward.w_2", d, f)
            for j in range(2):
                s[L + "sublayer.%d.norm.a_2" % j] = (d,)
                s[L + "sublayer.%d.norm.b_2" % j] = (d,)
        s[prefix + "norm.a_2"] = (d,)
        s[prefix + "norm.b_2"] = (d,)

    if kind == "sft":
        d = kw["d"]
        linear("embed.1", d, 512)
        encoder("encoder.", d)
        s["decoder.weight_ih_l0"] = (4 * d, 2 * d)
        s["decoder.weight_hh_l0"] = (4 * d, d)
        s["decoder.bias_ih_l0"] = (4 * d,)
        s["decoder.bias_hh_l0"] = (4 * d,)
        s["dec_h0"] = (1, 1, d)
        s["dec_c0"] = (1, 1, d)
        linear("out.0", 128, d)
        linear("out.2", 1, 128)
    elif kind == "b2":
        linear("embed", 256, 300)
        encoder("encoder.", 256)
        linear("out.0", 128, 256)
        linear("out.2", 1, 128)
    elif kind == "mft":
        mods = kw.get("mods") or R.MODS_AVL
        embed = kw.get("embed") or R.EMBED_AVL
        for m in mods:
            linear("embed_" + m, 256, embed[m])
            for j in range(4):
                linear("attn%s.linears.%d" % (m, j), 256, 256)      # dead parameters
            linear("ff%s.w_1" % m, 128, 256)
            linear("ff%s.w_2" % m, 256, 128)
            encoder("transformer_%s." % m, 256)
        H = sum(oracle.mfn_ref.HIDDEN[m] for m in mods)
        for m in mods:
            h = oracle.mfn_ref.HIDDEN[m]
            s["mfn.lstm_%s.weight_ih" % m] = (4 * h, 256)
            s["mfn.lstm_%s.weight_hh" % m] = (4 * h, h)
            s["mfn.lstm_%s.bias_ih" % m] = (4 * h,)
            s["mfn.lstm_%s.bias_hh" % m] = (4 * h,)
        linear("mfn.att1_fc1", 128, 2 * H)
        linear("mfn.att1_fc2", 2 * H, 128)
        linear("mfn.att2_fc1", 256, 2 * H)
        linear("mfn.att2_fc2", 128, 256)
        linear("mfn.gamma1_fc1", 64, 2 * H + 128)
        linear("mfn.gamma1_fc2", 128, 64)
        linear("mfn.gamma2_fc1", 64, 2 * H + 128)
        linear("mfn.gamma2_fc2", 128, 64)
        linear("mfn.out_fc1", 64, H + 128)
        linear("mfn.out_fc2", 1, 64)
    return s


def _check_model(fx, p, out, target_tag, lengths, T):
    mask = R.prefix_mask(lengths, T)
    target = R.gen_uniform(target_tag + ":target", (len(lengths), T, 1), R.SEED) * mask
    loss = oracle.masked_mse_sum_loss(out, target, lengths)
    loss.backward()
    np.testing.assert_allclose(out.detach().numpy(), fx["out"], atol=ATOL, rtol=0)
    assert abs(loss.item() - float(fx["loss"])) < 1e-5 * max(1.0, abs(float(fx["loss"])))
    # zero where the mask is zero (bit-exact: a product with 0.0)
    assert (out.detach().numpy()[mask.numpy() == 0] == 0).all()
    assert oracle.eval_ccc(fx["out"], out.detach().numpy()) > 1 - 1e-6
    for k in fx:
        if k.startswith("gnorm:"):
            g = p[k[6:]].grad
            if float(fx[k]) < 0:
                assert g is None or float(g.abs().sum()) == 0.0, k    # dead parameter (A11)
            else:
                n = float(g.double().pow(2).sum().sqrt())
                assert abs(n - float(fx[k])) <= 5e-4 * max(float(fx[k]), 1e-6), k
        if k.startswith("grad:"):
            assert grad_close(p[k[5:]].grad.numpy(), fx[k], GRAD_RTOL, 3e-7 * _grad_scale(fx)), k


@pytest.mark.parametrize("name,d,h", [("model_sft_d128", 128, 8), ("model_sft_d40", 40, 4), ("model_sft_default", 256, 8)])
def test_nlp_transformer(name, d, h):
    fx = load_golden(name)
    p = _params(_model_shapes("sft", d=d), fx)
    lengths = list(fx["lengths"])
    x = torch.tanh(R.gen_normal(name + ":x", (4, 50, 512), R.SEED))
    out = oracle.nlp_transformer(p, x, R.prefix_mask(lengths, 50), h)
    _check_model(fx, p, out, name, lengths, 50)


def test_uni_full_transformer():
    fx = load_golden("model_b2_text")
    p = _params(_model_shapes("b2"), fx)
    lengths = list(fx["lengths"])
    x = R.gen_normal("model_b2:x", (4, 50, 300), R.SEED)
    out = oracle.uni_full_transformer(p, x, R.prefix_mask(lengths, 50), 8)
    _check_model(fx, p, out, "model_b2_text", lengths, 50)


def test_multi_transformer():
    fx = load_golden("model_mft_avl")
    p = _params(_model_shapes("mft"), fx)
    lengths = list(fx["lengths"])
    mods = R.MODS_AVL
    ins = {m: R.gen_normal("model_mft:" + m, (4, 50, R.EMBED_AVL[m]), R.SEED) for m in mods}
    out = oracle.multi_transformer(p, ins, R.prefix_mask(lengths, 50), mods, 8)
    _check_model(fx, p, out, "model_mft_avl", lengths, 50)


@pytest.mark.parametrize("name,mods,embed", R.MFT_SWEEP)
def test_multi_transformer_sweep_shapes(name, mods, embed):
    """the other models of the reference's MFT sweep (transformer/MFT/train.py:538-552): VA-88, AL-88, VAL-44"""
    fx = load_golden(name)
    p = _params(_model_shapes("mft", mods=mods, embed=embed), fx)
    lengths = list(fx["lengths"])
    ins = {m: R.gen_normal(name + ":" + m, (4, 50, embed[m]), R.SEED) for m in mods}
    out = oracle.multi_transformer(p, ins, R.prefix_mask(lengths, 50), mods, 8)
    _check_model(fx, p, out, name, lengths, 50)


def test_eval_ccc():
    fx = load_golden("ccc")
    assert abs(oracle.eval_ccc(fx["a"], fx["b"]) - float(fx["ccc"])) < 1e-12
    assert abs(oracle.eval_ccc(fx["a"], fx["a"]) - 1.0) < 1e-12
