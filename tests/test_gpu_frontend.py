"""GPU parity of the window-encoder front-end (csrc/convpool.h, multimodal_transformer_amd/models.py) against the
fixtures captured from the reference's models.py and against the CPU oracle.

Tolerances as in test_gpu_parity.py (bf16 MFMA operands, fp32 accumulation): outputs <= 2e-2 rel-L2.  The max-pool
adds an index choice: where two conv positions of a window tie to within bf16 round-off the kernel may pick the other
one, which reroutes that (window, channel)'s whole gradient — the same mechanism as a ReLU mask flip — so weight
gradients are compared (a) tightly against the exact gradient FOR THE KERNEL'S OWN argmax, and (b) against the
reference's with the RELU_GRAD_RTOL bound; the argmax itself must be a position whose fp64 conv value lies within
round-off of the true maximum (bit-exact index semantics cannot be asked of a reduced-precision product).
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import oracle
import recipe as R
from conftest import load_golden, rel_l2, grad_close

pytestmark = pytest.mark.gpu

OUT_RTOL = 2e-2
GRAD_RTOL = 4e-2
RELU_GRAD_RTOL = 9e-2
CCC_MIN = 1 - 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _conv_fp64(x, w, b):
    """all conv values (N, W-1, F) in fp64"""
    N, W, D = x.shape
    y = b.double().view(1, 1, -1) + x[:, :-1, :].double() @ w[:, :, 0].double().t() + x[:, 1:, :].double() @ w[:, :, 1].double().t()
    return y


@pytest.mark.parametrize("case", R.CNN_CASES + [("fe_cnn_odd", 52, 70, 5, 37), ("fe_cnn_w2", 16, 8, 2, 9)],
                         ids=[c[0] for c in R.CNN_CASES] + ["fe_cnn_odd", "fe_cnn_w2"])
def test_conv_maxpool(dev, case):
    import multimodal_transformer_amd.functional as F
    name, D, Fo, W, N = case
    shapes = OrderedDict([("conv1d.weight", (Fo, D, 2)), ("conv1d.bias", (Fo,))])
    p = R.gen_params(shapes, R.SEED)
    x = R.gen_normal(name + ":x", (N, W, D), R.SEED)
    g = R.gen_normal(name + ":g", (N, Fo), R.SEED)
    w, b = p["conv1d.weight"], p["conv1d.bias"]
    wd, bd = w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
    out, arg = F.conv_maxpool(x.to(dev), wd, bd)
    (out * g.to(dev)).sum().backward()
    out_c, arg_c = out.detach().cpu(), arg.cpu().long()
    y = _conv_fp64(x, w, b)                                            # (N, W-1, F)
    ref_out, ref_arg = y.max(dim=1)
    r = rel_l2(out_c.numpy(), ref_out.numpy())
    agree = float((arg_c == ref_arg).float().mean())
    print("%-14s out rel_l2 %.3e   argmax agreement %.4f" % (name, r, agree))
    assert r < OUT_RTOL
    assert arg_c.min() >= 0 and arg_c.max() <= W - 2
    # the chosen position is a maximum up to bf16 round-off of the product (|terms| summed ~ sqrt(2D) sigma)
    chosen = y.gather(1, arg_c.unsqueeze(1)).squeeze(1)
    slack = 3e-2 * float(y.std()) + 1e-6
    assert float((ref_out - chosen).max()) <= slack
    assert agree > 0.9
    try:
        fx = load_golden(name)
    except FileNotFoundError:
        fx = None
    if fx is not None:
        assert rel_l2(out_c.numpy(), fx["out"]) < OUT_RTOL
    # (a) exact gradient for the kernel's own argmax
    gw = torch.zeros(Fo, D, 2, dtype=torch.float64)
    xd = x.double()
    n_idx = torch.arange(N).unsqueeze(1).expand(N, Fo)
    for j in range(2):
        rows = xd[n_idx, arg_c + j]                                    # (N, F, D)
        gw[:, :, j] = (g.double().unsqueeze(2) * rows).sum(dim=0)
    ra = rel_l2(wd.grad.cpu().numpy(), gw.numpy())
    rb = rel_l2(bd.grad.cpu().numpy(), g.double().sum(0).numpy())
    print("%-14s dW vs exact-for-own-argmax %.3e   db %.3e" % (name, ra, rb))
    assert ra < 1e-2 and rb < 1e-5
    # (b) against the reference's gradient
    if fx is not None:
        # a (window, channel) pair whose argmax differs contributes its gradient at another position: with a fraction
        # q of such pairs the relative-L2 distance to the reference is ~ sqrt(2q), whatever the arithmetic precision
        gwk = wd.grad.cpu().numpy()
        bound = GRAD_RTOL + 1.5 * np.sqrt(2.0 * (1.0 - agree))
        rh, rtl = rel_l2(gwk[:, :8, :], fx["gw_head"]), rel_l2(gwk[:, -8:, :], fx["gw_tail"])
        print("%-14s dW vs reference: head %.3e tail %.3e (bound %.3e from argmax agreement)" % (name, rh, rtl, bound))
        assert rh < bound and rtl < bound
        assert abs(np.sqrt((gwk.astype(np.float64) ** 2).sum()) - float(fx["gw_norm"])) < bound * float(fx["gw_norm"])
        assert rel_l2(bd.grad.cpu().numpy(), fx["gb"]) < 1e-4


def test_conv_maxpool_is_per_window(dev):
    """a window's result does not depend on its neighbours or its place in the batch (bit-exact), N not a multiple of 8"""
    import multimodal_transformer_amd.functional as F
    N, W, D, Fo = 21, 12, 88, 256
    x = R.gen_normal("cpw:x", (N, W, D), 3).to(dev)
    w = (R.gen_normal("cpw:w", (Fo, D, 2), 3) / np.sqrt(2 * D)).to(dev)
    b = R.gen_normal("cpw:b", (Fo,), 3).to(dev)
    out, arg = F.conv_maxpool(x, w, b)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(0)).to(dev)
    out2, arg2 = F.conv_maxpool(x[perm].contiguous(), w, b)
    assert torch.equal(out2, out[perm]) and torch.equal(arg2, arg[perm])
    out3, _ = F.conv_maxpool(x[5:6].contiguous(), w, b)
    assert torch.equal(out3[0], out[5])


def test_conv_maxpool_rejects_what_it_cannot_do(dev):
    import multimodal_transformer_amd.functional as F
    x = torch.zeros(3, 5, 18, device=dev)
    with pytest.raises(RuntimeError):
        F.conv_maxpool(x, torch.zeros(8, 18, 2, device=dev), torch.zeros(8, device=dev))          # D % 4 != 0
    with pytest.raises(NotImplementedError):
        F.conv_maxpool(torch.zeros(3, 5, 16, device=dev), torch.zeros(8, 16, 3, device=dev), torch.zeros(8, device=dev))
    with pytest.raises(RuntimeError):
        F.conv_maxpool(torch.zeros(3, 5, 16), torch.zeros(8, 16, 2), torch.zeros(8))               # CPU tensors


def _load_named(model, seed=R.SEED):
    p32 = R.gen_params(R.shapes_of(model.state_dict()), seed)
    model.load_state_dict(p32)
    return p32


def test_highway_golden(dev):
    from multimodal_transformer_amd import models as M
    fx = load_golden("fe_highway")
    hw = M.Highway(256)
    p32 = _load_named(hw)
    assert abs(R.weights_checksum(p32) - float(fx["checksum"])) <= 1e-6 * float(fx["checksum"])
    hw = hw.to(dev)
    x = R.gen_normal("fe_highway:x", (12, 256), R.SEED).to(dev).requires_grad_()
    g = R.gen_normal("fe_highway:g", (12, 256), R.SEED).to(dev)
    y = hw(x)
    (y * g).sum().backward()
    assert rel_l2(y.detach().cpu().numpy(), fx["out"]) < OUT_RTOL
    assert rel_l2(x.grad.cpu().numpy(), fx["dx"]) < GRAD_RTOL
    for n, p in hw.named_parameters():
        assert rel_l2(p.grad.cpu().numpy(), fx["grad:" + n]) < GRAD_RTOL, n


@pytest.mark.parametrize("shape", [(12, 256), (7, 44), (1, 4)])
def test_highway_train_mode_mask_replay(dev, shape):
    """The front-end's Dropout(0.3) rides in the Highway combine (mmt_highway_forward / _backward, stream 3000): the mask the kernels
    applied is read back through mmt_debug_dropout_mask and replayed through the oracle, forward and backward (odd element counts: the
    generator decides index PAIRS, the last element of an odd n stands alone)."""
    import oracle
    from multimodal_transformer_amd import models as M, functional as F
    rows, n = shape
    hw = M.Highway(n)
    p32 = _load_named(hw)
    hw = hw.to(dev)
    x_c = R.gen_normal("fe_highway_drop:x%d" % n, (rows, n), R.SEED)
    g_c = R.gen_normal("fe_highway_drop:g%d" % n, (rows, n), R.SEED)
    x = x_c.to(dev).requires_grad_()
    pdrop, seed = 0.3, 4242
    y = hw(x, pdrop, seed)
    (y * g_c.to(dev)).sum().backward()
    keep, scale = F.dropout_mask(pdrop, seed, 3000, rows * n, dev)
    keep = keep.reshape(rows, n).cpu()
    assert abs(scale - 1.0 / (1.0 - pdrop)) < 1e-3
    if rows * n >= 1000:
        assert abs(float(keep.float().mean()) - (1.0 - pdrop)) < 0.05
    pd = {k: v.double().requires_grad_() for k, v in p32.items()}
    xd = x_c.double().requires_grad_()
    ref = oracle.highway(pd, "", xd) * keep.double() * scale
    (ref * g_c.double()).sum().backward()
    assert torch.equal(y.detach().cpu() == 0, ~keep | (ref.detach() == 0)), "dropped elements must be exactly zero"
    assert grad_close_t(y.detach().cpu(), ref.detach(), OUT_RTOL)
    assert grad_close_t(x.grad.cpu(), xd.grad, GRAD_RTOL)
    for nme, prm in hw.named_parameters():
        assert grad_close_t(prm.grad.cpu(), pd[nme].grad, GRAD_RTOL), nme
    # a second call with another seed draws another mask; the same seed reproduces the first
    y2 = hw(x.detach(), pdrop, seed + 1)
    y3 = hw(x.detach(), pdrop, seed)
    assert torch.equal(y3, y.detach())
    if rows * n >= 64:
        assert not torch.equal(y2, y.detach())


def grad_close_t(a, b, rtol):
    from conftest import grad_close
    return grad_close(a.double().numpy(), b.double().numpy(), rtol, atol=1e-5)


@pytest.mark.parametrize("name,cls,mods,extra", [
    ("fe_model_sft", "MultiCNNTransformer", R.MODS_AVL, ()),
    ("fe_model_mft", "MultiCNNTransformerMFT", R.MODS_AVL, (R.FE_EMBED_MFT,)),
    ("fe_model_b2", "MultiCNNTransformerB2", ["linguistic"], ())])
def test_multi_cnn_transformer_golden(dev, name, cls, mods, extra):
    from multimodal_transformer_amd import models as M, eval_ccc
    fx = load_golden(name)
    model = getattr(M, cls)(mods, R.FE_DIMS, *extra, device=dev)
    p32 = _load_named(model)
    assert abs(R.weights_checksum(p32) - float(fx["checksum"])) <= 1e-6 * float(fx["checksum"]), "state_dict differs from the reference's"
    model = model.to(dev).eval()
    lengths = list(fx["lengths"])
    B, T = len(lengths), 6
    mask = R.prefix_mask(lengths, T)
    inputs = {m: R.gen_normal("%s:%s" % (name, m), (B, T, R.FE_WINDOW[m], R.FE_DIMS[m]), R.SEED).to(dev) for m in mods}
    target = (R.gen_uniform(name + ":target", (B, T, 1), R.SEED) * mask).to(dev)
    out = model(inputs, lengths, mask.to(dev))
    loss = ((out - target) ** 2).sum() / float(sum(lengths))
    loss.backward()
    o = out.detach().cpu().numpy()
    r = rel_l2(o, fx["out"])
    print("%-14s valence rel_l2 %.3e  loss %.6f (ref %.6f)" % (name, r, loss.item(), float(fx["loss"])))
    assert o.shape == fx["out"].shape and r < OUT_RTOL
    assert (o[mask.numpy() == 0] == 0).all()
    assert abs(loss.item() - float(fx["loss"])) < 2e-2 * max(abs(float(fx["loss"])), 1e-3)
    floor = 1e-3 * max(float(fx[k]) for k in fx if k.startswith("gnorm:"))
    worst = 0.0
    for n, p in model.named_parameters():
        ref = float(fx["gnorm:" + n])
        if ref < 0:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, n
            continue
        assert p.grad is not None, n
        got = float(p.grad.double().pow(2).sum().sqrt())
        if ref > floor:
            worst = max(worst, abs(got - ref) / ref)
        assert abs(got - ref) <= RELU_GRAD_RTOL * ref + floor, (n, got, ref)
    print("%-14s worst |grad-norm| deviation %.3e" % (name, worst))
    for k in fx:
        if k.startswith("grad:"):
            got = dict(model.named_parameters())[k[5:]].grad.cpu().numpy()
            assert rel_l2(got, fx[k]) < RELU_GRAD_RTOL, k
