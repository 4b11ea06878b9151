"""GPU parity tests for the T-sequential scans and the sequence models (HIP path vs oracle / golden).

Tolerances as in test_gpu_parity.py (bf16 MFMA operands, fp32 state): outputs within OUT_RTOL relative-L2,
gradients within GRAD_RTOL (RELU_GRAD_RTOL where a ReLU lies on the path), valence CCC >= 1 - 1e-3, and the
mask product exact (outputs are exactly 0 where mask == 0).
"""
import numpy as np
import pytest
import torch

import oracle
import recipe as R
from conftest import load_golden, rel_l2, grad_close
from test_gpu_parity import OUT_RTOL, GRAD_RTOL, RELU_GRAD_RTOL, CCC_MIN, _report, mta

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _oracle_lstm(gx, W, h0, c0):
    T, B, H4 = gx.shape
    H = H4 // 4
    h = h0 if h0 is not None else torch.zeros(B, H, dtype=gx.dtype)
    c = c0 if c0 is not None else torch.zeros(B, H, dtype=gx.dtype)
    eye = torch.eye(4 * H, dtype=gx.dtype)
    zb = torch.zeros(4 * H, dtype=gx.dtype)
    hs, cs = [], []
    for t in range(T):
        # the oracle's cell with the pre-projected gx[t] as its input and an identity input weight
        h, c = oracle.lstm_cell(gx[t], h, c, eye, W, zb, zb)
        hs.append(h)
        cs.append(c)
    return torch.stack(hs), torch.stack(cs)


@pytest.mark.parametrize("T,B,H,init", [(20, 3, 88, False), (7, 17, 48, False), (50, 4, 128, True), (12, 33, 40, True),
                                        (1, 1, 16, False), (9, 5, 256, True),
                                        # > 256 sequences: 2 per workgroup; > 512: 4+ per workgroup and the per-lane loader
                                        (6, 300, 88, True), (5, 257, 64, True), (4, 259, 128, False), (5, 601, 48, True), (4, 1100, 128, False),
                                        # hidden sizes above 128: four workgroups per sequence up to 32 sequences, one beyond
                                        (40, 32, 256, True), (7, 3, 200, True), (5, 1, 132, False), (6, 40, 256, True), (3, 300, 160, False),
                                        # T = 1000 (configs[4]): the MFN cell sizes and the reference-default decoder size
                                        (1000, 2, 88, False), (1000, 3, 48, True), (1000, 2, 256, True)])
def test_lstm_scan(dev, T, B, H, init):
    tag = "lstm%d_%d_%d" % (T, B, H)
    gx = R.gen_normal(tag + "gx", (T, B, 4 * H), 13)
    W = R.gen_normal(tag + "w", (4 * H, H), 13) / np.sqrt(H)
    h0 = 0.5 * R.gen_normal(tag + "h0", (B, H), 13) if init else None
    c0 = 0.5 * R.gen_normal(tag + "c0", (B, H), 13) if init else None
    gh, gc = R.gen_normal(tag + "gh", (T, B, H), 13), R.gen_normal(tag + "gc", (T, B, H), 13)
    leaves = [t.double().requires_grad_() if t is not None else None for t in (gx, W, h0, c0)]
    rh, rc = _oracle_lstm(*leaves)
    ((rh * gh.double()).sum() + (rc * gc.double()).sum()).backward()
    F = mta().functional
    gl = [t.to(dev).requires_grad_() if t is not None else None for t in (gx, W, h0, c0)]
    h_all, c_all = F.lstm_scan(*gl)
    ((h_all * gh.to(dev)).sum() + (c_all * gc.to(dev)).sum()).backward()
    assert _report(tag + " h", h_all.detach().cpu(), rh.detach()) < OUT_RTOL
    assert _report(tag + " c", c_all.detach().cpu(), rc.detach()) < OUT_RTOL
    for name, a, b in zip(("dgx", "dW", "dh0", "dc0"), gl, leaves):
        if a is not None:
            assert _report(tag + " " + name, a.grad.cpu(), b.grad) < GRAD_RTOL, name
    F.check_device_errors()                 # the four-CU scans' exchange time-out word must be zero


def _oracle_mem_scan(apre, chat, Wm, W2, b2, drop=None):
    """the memory recurrence of transformer/MFT/multiTransformer.py:221-224 behind the batched `attended` part of both gamma fc1 layers;
    `drop` (T,B,128): the gamma{1,2}_dropout multipliers on relu(fc1) (:222-223), gamma1's 64 units first"""
    T, B, U = apre.shape
    MD, HG = chat.shape[-1], W2.shape[-1]
    mem = torch.zeros(B, MD, dtype=apre.dtype)
    out = []
    for t in range(T):
        u = torch.relu(apre[t] + mem @ Wm.t())
        if drop is not None:
            u = u * drop[t]
        g1 = torch.sigmoid(u[:, :HG] @ W2[0].t() + b2[0])
        g2 = torch.sigmoid(u[:, HG:] @ W2[1].t() + b2[1])
        mem = g1 * mem + g2 * chat[t]
        out.append(mem)
    return torch.stack(out)


@pytest.mark.parametrize("T,B", [(20, 3), (1, 1), (33, 18), (5, 300), (4, 257), (3, 1030), (1000, 2)])
def test_mfn_mem_scan(dev, T, B):
    tag = "mem%d_%d" % (T, B)
    apre = R.gen_normal(tag + "a", (T, B, 128), 17)
    chat = torch.tanh(R.gen_normal(tag + "c", (T, B, 128), 17))
    Wm = R.gen_normal(tag + "wm", (128, 128), 17) / np.sqrt(128)
    W2 = R.gen_normal(tag + "w2", (2, 128, 64), 17) / 8
    b2 = 0.1 * R.gen_normal(tag + "b2", (2, 128), 17)
    g = R.gen_normal(tag + "g", (T, B, 128), 17)
    leaves = [t.double().requires_grad_() for t in (apre, chat, Wm, W2, b2)]
    ref = _oracle_mem_scan(*leaves)
    (ref * g.double()).sum().backward()
    gl = [t.to(dev).requires_grad_() for t in (apre, chat, Wm, W2, b2)]
    out = mta().functional.mfn_mem_scan(*gl)
    (out * g.to(dev)).sum().backward()
    assert _report(tag + " mem", out.detach().cpu(), ref.detach()) < OUT_RTOL
    for name, a, b in zip(("dapre", "dchat", "dWm", "dW2", "db2"), gl, leaves):
        assert _report(tag + " " + name, a.grad.cpu(), b.grad) < RELU_GRAD_RTOL, name


@pytest.mark.parametrize("T,B", [(20, 3), (300, 33), (6, 300), (5, 1030)])
def test_mfn_mem_scan_train_mode_mask_replay(dev, T, B):
    """gamma{1,2}_dropout = Dropout(0.2) inside the memory scan (transformer/MFT/multiTransformer.py:168,172,222-223): the mask the
    forward kernel draws (stream 1000, index = (t B + b) 128 + unit) is rebuilt with mmt_debug_dropout_mask and replayed through the
    oracle — the memory trajectory AND every gradient, so a backward that regenerated or applied another mask than the forward would
    fail here.  B = 3 / 33: one sequence per workgroup; 300: two; 1030: the general kernel."""
    F = mta().functional
    p, seed = 0.2, 4242 + T
    tag = "memdrop%d_%d" % (T, B)
    apre = R.gen_normal(tag + "a", (T, B, 128), 19)
    chat = torch.tanh(R.gen_normal(tag + "c", (T, B, 128), 19))
    Wm = R.gen_normal(tag + "wm", (128, 128), 19) / np.sqrt(128)
    W2 = R.gen_normal(tag + "w2", (2, 128, 64), 19) / 8
    b2 = 0.1 * R.gen_normal(tag + "b2", (2, 128), 19)
    g = R.gen_normal(tag + "g", (T, B, 128), 19)
    gl = [t.to(dev).requires_grad_() for t in (apre, chat, Wm, W2, b2)]
    out = F.mfn_mem_scan(*gl, dropout_p=p, seed=seed)
    (out * g.to(dev)).sum().backward()
    keep, sc = F.dropout_mask(p, seed, 1000, T * B * 128, dev)
    assert abs(float(1 - keep.float().mean()) - p) < 4 * np.sqrt(p * (1 - p) / keep.numel()) + 1e-3
    drop = (keep.reshape(T, B, 128).double() * sc).cpu()
    leaves = [t.double().requires_grad_() for t in (apre, chat, Wm, W2, b2)]
    ref = _oracle_mem_scan(*leaves, drop=drop)
    (ref * g.double()).sum().backward()
    plain = _oracle_mem_scan(*[t.double() for t in (apre, chat, Wm, W2, b2)])
    assert rel_l2(plain.numpy(), ref.detach().numpy()) > 1e-2, "the mask changed nothing: the test would not see a wrong one"
    assert _report(tag + " mem", out.detach().cpu(), ref.detach()) < OUT_RTOL
    for name, a, b in zip(("dapre", "dchat", "dWm", "dW2", "db2"), gl, leaves):
        assert _report(tag + " " + name, a.grad.cpu(), b.grad) < RELU_GRAD_RTOL, name
    out2 = F.mfn_mem_scan(*[t.detach() for t in gl], dropout_p=p, seed=seed)
    assert torch.equal(out2, out.detach()), "same seed, same mask"


@pytest.mark.parametrize("T,B", [(20, 3), (300, 33)])
def test_mfn_gate_train_mode_mask_replay(dev, T, B, monkeypatch):
    """The whole MFN module in TRAIN mode (gamma dropouts 0.2 inside the scan, out_dropout 0.5 behind the read-out's ReLU,
    transformer/MFT/multiTransformer.py:168-176,222-223,245) against the oracle run with the masks the kernels drew: output, input
    gradients of every modality and every parameter gradient at the eval-mode tolerances."""
    MT, F, L = mta().multiTransformer, mta().functional, mta()._lib
    mods = R.MODS_AVL
    mfn = MT.MFN(mods, {m: 256 for m in mods}, 1, device=dev)
    p32 = _load_into(mfn, seed=31)
    mfn = mfn.to(dev).train()
    seeds = {2: 777001 + T, 4: 777002 + T}
    real = L.next_dropout_seed
    monkeypatch.setattr(L, "next_dropout_seed", lambda device, site, holder=None, index=0: seeds[site] if site in seeds else real(device, site, holder, index))
    x = {m: R.gen_normal("mfndrop:" + m, (T, B, 256), 31) for m in mods}
    g = R.gen_normal("mfndrop:g", (B, T, 1), 31)
    ins = {m: x[m].to(dev).requires_grad_() for m in mods}
    y = mfn(ins)
    (y * g.to(dev)).sum().backward()
    pg, po = mfn.gamma1_dropout.p, mfn.out_dropout.p
    assert pg == 0.2 and po == 0.5                      # the reference's constants (:168,172,176)
    kg, sg = F.dropout_mask(pg, seeds[2], 1000, T * B * 128, dev)
    ko, so = F.dropout_mask(po, seeds[4], 2001, T * B * 64, dev)       # linear out-dropout: index m * NP + n, NP = 64
    gd = (kg.reshape(T, B, 128).double() * sg).cpu()
    od = (ko.reshape(T, B, 64).double() * so).cpu()
    pd = {k: v.double().requires_grad_() for k, v in p32.items()}
    xd = {m: x[m].double().requires_grad_() for m in mods}
    ref = oracle.mfn_gate(pd, "", xd, mods, gamma_drop=(gd[..., :64], gd[..., 64:]), out_drop=od)
    (ref * g.double()).sum().backward()
    tag = "mfn train %dx%d" % (T, B)
    assert _report(tag + " out", y.detach().cpu(), ref.detach()) < OUT_RTOL
    for m in mods:
        assert _report(tag + " dx:" + m, ins[m].grad.cpu(), xd[m].grad) < RELU_GRAD_RTOL
    scale = max(float(v.grad.abs().max()) for v in pd.values())
    for n, q in mfn.named_parameters():
        got, want = q.grad.cpu().numpy(), pd[n].grad.numpy()
        assert grad_close(got, want, RELU_GRAD_RTOL, 3e-3 * scale), "%s: rel-L2 %.3e" % (n, rel_l2(got, want))


def _load_into(module, seed=R.SEED):
    p32 = R.gen_params(R.shapes_of(module.state_dict()), seed)
    module.load_state_dict(p32)
    return p32


def test_mfn_gate_golden(dev):
    fx = load_golden("mfn_avl")
    MT = mta().multiTransformer
    mods = R.MODS_AVL
    mfn = MT.MFN(mods, {m: 256 for m in mods}, 1, device=dev)
    p32 = _load_into(mfn)
    assert abs(R.weights_checksum(p32) - float(fx["checksum"])) <= 1e-6 * float(fx["checksum"])
    mfn = mfn.to(dev).eval()
    ins = {m: R.gen_normal("mfn:" + m, (20, 3, 256), R.SEED).to(dev).requires_grad_() for m in mods}
    g = R.gen_normal("mfn:g", (3, 20, 1), R.SEED).to(dev)
    y = mfn(ins)
    (y * g).sum().backward()
    assert y.shape == (3, 20, 1)
    assert _report("mfn out", y.detach().cpu(), fx["out"]) < OUT_RTOL
    for m in mods:
        assert _report("mfn dx:" + m, ins[m].grad.cpu(), fx["dx:" + m]) < RELU_GRAD_RTOL
    scale = max(float(np.abs(fx[k]).max()) for k in fx if k.startswith("grad:"))
    for n, p in mfn.named_parameters():
        got, ref = p.grad.cpu().numpy(), fx["grad:" + n]
        if not grad_close(got, ref, RELU_GRAD_RTOL, 3e-3 * scale):
            _report("mfn FAIL d" + n, got, ref)
        assert grad_close(got, ref, RELU_GRAD_RTOL, 3e-3 * scale), n


def _check_model(dev, fx, model, out_fn, name, lengths, T):
    mask = R.prefix_mask(lengths, T)
    target = (R.gen_uniform(name + ":target", (len(lengths), T, 1), R.SEED) * mask).to(dev)
    out = out_fn(mask.to(dev))
    loss = ((out - target) ** 2).sum() / float(sum(lengths))
    loss.backward()
    o = out.detach().cpu().numpy()
    assert o.shape == fx["out"].shape
    r = _report(name + " valence", o, fx["out"])
    ccc = mta().eval_ccc(fx["out"], o)
    print("%-44s CCC %.6f   loss %.6f (ref %.6f)" % (name, ccc, loss.item(), float(fx["loss"])))
    assert r < OUT_RTOL and ccc >= CCC_MIN
    assert (o[mask.numpy() == 0] == 0).all()                       # exact zeros where mask == 0
    assert abs(loss.item() - float(fx["loss"])) < 2e-2 * max(abs(float(fx["loss"])), 1e-3)
    worst = 0.0
    # absolute floor for analytically-zero gradients (key-projection bias: softmax is shift invariant)
    floor = 1e-3 * max(float(fx[k]) for k in fx if k.startswith("gnorm:"))
    for n, p in model.named_parameters():
        ref = float(fx["gnorm:" + n])
        if ref < 0:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, n     # dead parameters of the reference
            continue
        assert p.grad is not None, n
        got = float(p.grad.double().pow(2).sum().sqrt())
        if ref > floor:
            worst = max(worst, abs(got - ref) / ref)
        assert abs(got - ref) <= RELU_GRAD_RTOL * ref + floor, (n, got, ref)
    print("%-44s worst |grad-norm| deviation %.3e" % (name, worst))
    for k in fx:
        if k.startswith("grad:"):
            got = dict(model.named_parameters())[k[5:]].grad.cpu().numpy()
            assert _report(name + " d" + k[5:], got, fx[k]) < RELU_GRAD_RTOL, k


@pytest.mark.parametrize("name,kw", [("model_sft_d128", dict(embed_dim=128, h=8)), ("model_sft_d40", dict(embed_dim=40, h=4)),
                                     ("model_sft_default", dict())])
def test_nlp_transformer_golden(dev, name, kw):
    fx = load_golden(name)
    MT = mta().multiTransformer
    model = MT.NLPTransformer(512, device=dev, **kw)
    _load_into(model)
    model = model.to(dev).eval()
    lengths = list(fx["lengths"])
    x = torch.tanh(R.gen_normal(name + ":x", (4, 50, 512), R.SEED)).to(dev)
    _check_model(dev, fx, model, lambda mask: model(x, mask, lengths), name, lengths, 50)


def test_uni_full_transformer_golden(dev):
    fx = load_golden("model_b2_text")
    MT = mta().multiTransformer
    model = MT.UniFullTransformer(300, device=dev)
    _load_into(model)
    model = model.to(dev).eval()
    lengths = list(fx["lengths"])
    x = R.gen_normal("model_b2:x", (4, 50, 300), R.SEED).to(dev)
    _check_model(dev, fx, model, lambda mask: model(x, mask, lengths), "model_b2_text", lengths, 50)


def test_multi_transformer_golden(dev):
    fx = load_golden("model_mft_avl")
    MT = mta().multiTransformer
    mods = R.MODS_AVL
    model = MT.MultiTransformer(mods, R.EMBED_AVL, device=dev)
    _load_into(model)
    model = model.to(dev).eval()
    lengths = list(fx["lengths"])
    ins = {m: R.gen_normal("model_mft:" + m, (4, 50, R.EMBED_AVL[m]), R.SEED).to(dev) for m in mods}
    _check_model(dev, fx, model, lambda mask: model(ins, mask, lengths), "model_mft_avl", lengths, 50)


@pytest.mark.parametrize("name,mods,embed", R.MFT_SWEEP)
def test_multi_transformer_sweep_shapes_golden(dev, name, mods, embed):
    """the other models of the reference's MFT sweep (transformer/MFT/train.py:538-552): two-modality gates (VA, AL) and the 44-wide
    acoustic embed run through other concatenation tables and row-GEMM paddings than VAL-88"""
    fx = load_golden(name)
    MT = mta().multiTransformer
    model = MT.MultiTransformer(mods, embed, device=dev)
    _load_into(model)
    model = model.to(dev).eval()
    lengths = list(fx["lengths"])
    ins = {m: R.gen_normal(name + ":" + m, (4, 50, embed[m]), R.SEED).to(dev) for m in mods}
    _check_model(dev, fx, model, lambda mask: model(ins, mask, lengths), name, lengths, 50)


def test_mse_sum_loss_and_gradient(dev):
    """fused training loss (mmt_mse_sum_forward) = MSELoss(reduction='sum') / sum(lengths) of the reference's train loop
    (transformer/SFT/train.py:133-139), value and gradient, incl. a size that is not a multiple of 4 and an upstream scale"""
    from multimodal_transformer_amd import functional as F
    for shape, denom in [((7, 33, 1), 190.0), ((32, 500, 128), 16000.0), ((3, 5), 1.0)]:
        p = R.gen_uniform("mse:p", shape, 3)
        t = R.gen_uniform("mse:t", shape, 3)
        pd = p.double().requires_grad_(True)
        ref = torch.nn.MSELoss(reduction="sum")(pd, t.double()) / denom
        (3.0 * ref).backward()
        pg = p.to(dev).requires_grad_(True)
        got = F.mse_sum_loss(pg, t.to(dev), denom)
        (3.0 * got).backward()
        assert abs(got.item() - ref.item()) <= 2e-6 * abs(ref.item()), (got.item(), ref.item())
        assert rel_l2(pg.grad.cpu().double(), pd.grad) < 1e-6
        # deterministic: same bits on a second call
        assert F.mse_sum_loss(pg, t.to(dev), denom).item() == got.item()
    with pytest.raises(ValueError):
        F.mse_sum_loss(torch.zeros(2, 3, device=dev), torch.zeros(3, 2, device=dev), 1.0)


def test_loss_backward_helper_and_flat_parameter_storage(dev):
    """mse_sum_loss_backward == mse_sum_loss(...).backward() bit for bit; the Encoder re-seats its parameters as views of ONE buffer
    (no concatenation per step): optimiser updates and load_state_dict reach the kernels, .to() / .float() re-flatten"""
    from multimodal_transformer_amd import functional as F, multiTransformer as MT
    enc = MT.Encoder(MT.EncoderLayer(64, MT.MultiHeadedAttention(4, 64), MT.PositionwiseFeedForward(64, 96, 0.1), 0.1), 2)
    assert enc._fusable()
    p32 = R.gen_params(R.shapes_of(enc.state_dict()), 31)
    enc.load_state_dict(p32)
    enc = enc.to(dev).eval()
    x = R.gen_normal("flat:x", (3, 20, 64), 31).to(dev).requires_grad_(True)
    mask = R.prefix_mask([20, 11, 5], 20).to(dev)
    tgt = R.gen_uniform("flat:t", (3, 20, 64), 31).to(dev)
    F.mse_sum_loss(enc(x, mask), tgt, 36.0).backward()
    g1 = [q.grad.clone() for q in enc.parameters()] + [x.grad.clone()]
    ps = enc.flat_parameters()
    base = enc._flat.data_ptr()
    off = 0
    for q in ps:                                        # every parameter is a view into the flat buffer, in the library's order
        assert q.data_ptr() == base + 4 * off
        off += q.numel()
    for q in list(enc.parameters()) + [x]:
        q.grad = None
    loss = F.mse_sum_loss_backward(enc(x, mask), tgt, 36.0)
    for a, b in zip(g1, [q.grad for q in enc.parameters()] + [x.grad]):
        assert torch.equal(a, b)
    ref = oracle.encoder_stack({k: v.double() for k, v in p32.items()}, "", x.detach().cpu().double(), mask.cpu().double(), 4)
    assert abs(loss.item() - float(((ref - tgt.cpu().double()) ** 2).sum() / 36.0)) < 2e-2 * abs(loss.item())
    # an in-place update (what optimisers and load_state_dict do) is seen by the next forward
    y0 = enc(x, mask).detach().clone()
    with torch.no_grad():
        enc.norm.b_2.add_(0.5)
    assert torch.allclose(enc(x, mask).detach(), y0 + 0.5, atol=1e-6)
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    enc.load_state_dict({k: v.cpu() for k, v in p32.items()})
    assert torch.allclose(enc(x, mask).detach(), y0, atol=1e-6)
    enc.load_state_dict(sd)
    # .double().float() replaces every parameter's storage: the stack re-flattens by itself
    enc = enc.double().float()
    assert enc._flat is None
    assert torch.allclose(enc(x, mask).detach(), y0 + 0.5, atol=1e-6) and enc._flat is not None


def test_batched_ccc_on_device(dev):
    """device-side per-sequence CCC (mmt_ccc_forward) vs the values the reference's eval_ccc gave for the same rows
    (fixture `batching`, generated from transformer/SFT/train.py:42-50) and vs the numpy restatement at a large size"""
    fx = load_golden("batching")
    mt = mta()
    lengths = [int(v) for v in fx["lengths"]]
    pred, tgt = torch.from_numpy(fx["ccc:pred"]).to(dev), torch.from_numpy(fx["in:target"]).to(dev)
    got = mt.batched_ccc(pred.unsqueeze(2), tgt.unsqueeze(2), lengths).cpu().numpy()
    ref = fx["ccc:full"]
    for g, r, L in zip(got, ref, lengths):
        if L < 2:
            assert np.isnan(g)
        else:
            assert abs(g - r) < 1e-6 * max(1.0, abs(r)), (g, r)
    B, T = 64, 1000
    p = R.gen_uniform("ccc:p", (B, T), 3)
    t = (0.6 * p + 0.4 * R.gen_uniform("ccc:t", (B, T), 3))
    ls = [T - 13 * i for i in range(B)]
    got = mt.batched_ccc(p.to(dev), t.to(dev), ls).cpu().numpy()
    for b in range(B):
        assert abs(got[b] - mt.eval_ccc(t[b, :ls[b]].numpy(), p[b, :ls[b]].numpy())) < 1e-9


def test_evaluate_loop_matches_per_sequence_oracle(dev):
    """batching.evaluate with its default batch_size=1 (the reference's evaluation, transformer/SFT/train.py:210-214): per-sequence
    CCC, their mean and the loss per window equal the oracle's on the UNPADDED sequences.  Then the reason for that default is
    pinned: in a padded batch the shorter sequences attend to the padding windows (keys are never masked, SFT/multiTransformer.py:29-30),
    the HIP path reproduces that too (parity with the oracle on the same padded batch), and the result is far from the unpadded one."""
    from multimodal_transformer_amd import batching
    MT = mta().multiTransformer
    model = MT.NLPTransformer(512, embed_dim=40, h=4, N=2, device=dev)
    p32 = _load_into(model, 19)
    model = model.to(dev).eval()
    n, T = 7, 30
    lengths = [30, 12, 25, 30, 2, 17, 9]
    x = torch.tanh(R.gen_normal("evalloop:x", (n, T, 512), 19))
    tgt = R.gen_uniform("evalloop:t", (n, T), 19)

    class Wrap(torch.nn.Module):                        # evaluate() calls model(data_dict, lengths, mask) like the reference's wrapper
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, data, lengths, mask):
            return self.m(data["fused"], mask, lengths)

    stats = batching.evaluate(Wrap(model).eval(), {"fused": x.numpy()}, tgt.numpy(), lengths, device=dev)
    assert not model.training                            # evaluate() restores the mode it found
    cccs, sq, unpadded = [], 0.0, {}
    for i, L in enumerate(lengths):
        yo = oracle.nlp_transformer(p32, x[i:i + 1, :L], R.prefix_mask([L], L), 4)[0, :, 0].detach().numpy()
        unpadded[i] = yo
        sq += float(((yo - tgt[i, :L].numpy()) ** 2).sum())
        if L > 1:
            cccs.append(mta().eval_ccc(yo, tgt[i, :L].numpy()))
    assert len(stats["per_sequence_ccc"]) == len(cccs)
    for a, b in zip(stats["per_sequence_ccc"], cccs):    # batch_size=1: evaluation order = input order
        assert abs(a - b) < 2e-3, (a, b)
    assert abs(stats["ccc"] - float(np.mean(cccs))) < 1e-3
    assert abs(stats["loss"] - sq / sum(lengths)) < 2e-2 * sq / sum(lengths)
    # a padded batch: parity with the oracle on the SAME padded batch, and a visibly different valence for the shorter sequences
    mask3 = R.prefix_mask(lengths[:3], T)
    with torch.no_grad():
        got = model(x[:3].to(dev), mask3.to(dev), lengths[:3]).cpu().numpy()
    ref = oracle.nlp_transformer(p32, x[:3], mask3, 4).detach().numpy()
    assert rel_l2(got, ref) < 2e-2
    for i in (1, 2):
        L = lengths[i]
        assert rel_l2(got[i, :L, 0], unpadded[i]) > 0.2, "padding windows are keys of the shorter sequences in the reference"


def test_glue_ops_match_torch(dev):
    """The data-movement ops that replace torch.cat / permute / slicing / broadcasting around the scans (csrc/glue.h, mmt_copy2d): forward
    values equal torch's own ops exactly (they only move fp32 numbers; sums of two terms are commutative), gradients to round-off."""
    F = mta().functional
    B, T, d = 3, 7, 20
    x = R.gen_normal("glue:x", (B, T, d), 3).to(dev)
    m = R.prefix_mask([7, 4, 1], T).to(dev)
    g = R.gen_normal("glue:g", (B, T, d), 3).to(dev)

    def both(fn_hip, fn_ref, *leaves):
        a = [t.clone().requires_grad_() for t in leaves]
        b = [t.clone().requires_grad_() for t in leaves]
        ya, yb = fn_hip(*a), fn_ref(*b)
        assert ya.shape == yb.shape and torch.equal(ya, yb)
        go = R.gen_normal("glue:go%d" % ya.numel(), tuple(ya.shape), 3).to(dev)
        ya.backward(go)
        yb.backward(go)
        for p, q in zip(a, b):
            qg = q.grad if q.grad is not None else torch.zeros_like(q)      # (torch leaves the gradient of an unused leaf undefined)
            assert torch.allclose(p.grad, qg, rtol=2e-6, atol=1e-6)         # (a broadcast row's gradient is a sum over the batch: order may differ)

    both(F.time_major, lambda t: t.permute(1, 0, 2).contiguous(), x)
    xt = x.permute(1, 0, 2).contiguous()
    both(lambda t: F.batch_major(t, m), lambda t: t.permute(1, 0, 2) * m, xt)
    both(lambda t: F.batch_major(t), lambda t: t.permute(1, 0, 2).contiguous(), xt)
    W = R.gen_normal("glue:W", (12, 44), 3).to(dev)
    both(F.add2, lambda a, b: a + b, W, W.flip(0).contiguous())
    Wi, Wh = R.gen_normal("glue:Wi", (16, 8), 3).to(dev), R.gen_normal("glue:Wh", (16, 4), 3).to(dev)
    bi, bh = R.gen_normal("glue:bi", (16,), 3).to(dev), R.gen_normal("glue:bh", (16,), 3).to(dev)
    for k in range(4):          # the four outputs of the decoder's operand pack, each against the slicing / sums it replaces
        both(lambda a, b, c, e, k=k: F.decoder_pack(a, b, c, e)[k],
             lambda a, b, c, e, k=k: (a[:, 4:].contiguous(), a[:, :4] + b, b * 1.0, c + e)[k], Wi, Wh, bi, bh)
    v = R.gen_normal("glue:v", (44,), 3).to(dev)
    both(F.add2, lambda a, b: a + b, v, v.flip(0).contiguous())
    row = R.gen_normal("glue:row", (1, 1, d), 3).to(dev)
    both(lambda t, r: F.add_row0(t * 1.0, r), lambda t, r: torch.cat([t[:1] + r.reshape(1, 1, -1), t[1:]], 0), xt, row)
    both(lambda r: F.broadcast_rows(r, 5), lambda r: r.reshape(1, -1).expand(5, d).contiguous(), row)
    parts = [R.gen_normal("glue:p%d" % i, (T, B, w), 3).to(dev) for i, w in enumerate((8, 3, 20))]
    both(lambda *t: F.cat_cols(t), lambda *t: torch.cat(t, dim=-1), *parts)


@pytest.mark.parametrize("in_p,out_p", [(0.1, 0.0), (0.0, 0.5), (0.25, 0.5)])
def test_linear_fused_dropout_replay(dev, in_p, out_p):
    """Dropout fused into the affine map — on the input (NLPTransformer's Dropout -> Linear -> ReLU embed, transformer/SFT/multiTransformer.py:431-433)
    and behind the ReLU (MFN's out_dropout, transformer/MFT/multiTransformer.py:244-245): the masks the kernel used are rebuilt with
    mmt_debug_dropout_mask (streams 2000 / 2001) and replayed through plain fp64 torch, forward and all three gradients."""
    F = mta().functional
    M, K, N, seed = 70, 300, 96, 99
    KP, NP = -(-K // 64) * 64, -(-N // 64) * 64
    x = R.gen_normal("ldrop:x", (M, K), 7)
    W = R.gen_normal("ldrop:W", (N, K), 7) * 0.08
    b = R.gen_normal("ldrop:b", (N,), 7) * 0.1
    g = R.gen_normal("ldrop:g", (M, N), 7)
    xs, Ws, bs = (t.to(dev).requires_grad_() for t in (x, W, b))
    y = F.linear(xs, Ws, bs, act=1, in_dropout=in_p, out_dropout=out_p, seed=seed)
    y.backward(g.to(dev))
    min_ = torch.ones(M, K, dtype=torch.float64)
    mout = torch.ones(M, N, dtype=torch.float64)
    if in_p > 0:
        kk, sc = F.dropout_mask(in_p, seed, 2000, M * KP, dev)
        min_ = (kk.reshape(M, KP)[:, :K].double() * sc).cpu()
        assert abs(float(1 - kk.float().mean()) - in_p) < 0.02
    if out_p > 0:
        kk, sc = F.dropout_mask(out_p, seed, 2001, M * NP, dev)
        mout = (kk.reshape(M, NP)[:, :N].double() * sc).cpu()
    xd, Wd, bd = (t.double().requires_grad_() for t in (x, W, b))
    ref = torch.relu((xd * min_) @ Wd.t() + bd) * mout
    ref.backward(g.double())
    tag = "linear drop in %.2f out %.2f" % (in_p, out_p)
    assert _report(tag + " y", y.detach().cpu(), ref.detach()) < OUT_RTOL
    assert _report(tag + " dx", xs.grad.cpu(), xd.grad) < RELU_GRAD_RTOL
    assert _report(tag + " dW", Ws.grad.cpu(), Wd.grad) < RELU_GRAD_RTOL
    assert _report(tag + " db", bs.grad.cpu(), bd.grad) < RELU_GRAD_RTOL
    y2 = F.linear(xs.detach(), Ws.detach(), bs.detach(), act=1, in_dropout=in_p, out_dropout=out_p, seed=seed + 1)
    assert float((y2 - y.detach()).abs().max()) > 1e-3            # another seed, another mask


@pytest.mark.parametrize("K", [5, 43, 301])
def test_linear_with_an_input_width_that_is_no_multiple_of_four(dev, K):
    """nn.Linear accepts any in_features (the reference's own widths are all multiples of 4: 300, 88, 44, 256, 512, 812 ...); the row kernels
    stage 16-byte pieces, so functional.linear pads x and W with zero columns: same product, gradients of the real columns unchanged."""
    F = mta().functional
    M, N = 37, 24
    x = R.gen_normal("oddK:x", (M, K), 3)
    W = R.gen_normal("oddK:W", (N, K), 3) * 0.2
    b = R.gen_normal("oddK:b", (N,), 3) * 0.1
    g = R.gen_normal("oddK:g", (M, N), 3)
    xs, Ws, bs = (t.to(dev).requires_grad_() for t in (x, W, b))
    y = F.linear(xs, Ws, bs, act=2)
    y.backward(g.to(dev))
    xd, Wd, bd = (t.double().requires_grad_() for t in (x, W, b))
    ref = torch.tanh(xd @ Wd.t() + bd)
    ref.backward(g.double())
    assert _report("odd K=%d y" % K, y.detach().cpu(), ref.detach()) < OUT_RTOL
    assert _report("odd K=%d dx" % K, xs.grad.cpu(), xd.grad) < GRAD_RTOL
    assert _report("odd K=%d dW" % K, Ws.grad.cpu(), Wd.grad) < GRAD_RTOL
    assert _report("odd K=%d db" % K, bs.grad.cpu(), bd.grad) < GRAD_RTOL


def _device_kernel_names(step):
    """names of the device kernels one call of `step` launches (torch.profiler); None if the profiler reports no device activity"""
    from torch.profiler import profile, ProfilerActivity
    from torch.autograd import DeviceType
    step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == DeviceType.CUDA]
    return names or None


def _library_kernels(names):
    """kernels that are not ours: ATen element-wise / reduction / copy / cat kernels, rocBLAS / hipBLASLt / MIOpen GEMMs"""
    bad = ("at::", "at_cuda", "elementwise", "Cijk_", "rocblas", "hipblas", "miopen", "MIOpen", "CatArray", "reduce_kernel", "vectorized_")
    return sorted(set(n for n in names if any(b in n for b in bad)))


@pytest.mark.parametrize("which", ["mft", "sft", "raw_sft", "raw_mft"])
def test_train_step_runs_no_library_kernel(dev, which):
    """One train-mode forward + loss + backward of the whole MFT / SFT sequence model launches hand-written HIP kernels only — no ATen
    element-wise, concatenation or copy kernel, no library GEMM (the MFN gate's softmax(att1) * cStar, the shift of c, the concatenations
    and the output dropout, transformer/MFT/multiTransformer.py:210-219,238-246; the SFT embed dropout and decoder glue,
    transformer/SFT/multiTransformer.py:431-433,461-483, used to be torch ops)."""
    MT = mta().multiTransformer
    F = mta().functional
    B, T = 4, 40
    lengths = [40, 33, 20, 5]
    mask = R.prefix_mask(lengths, T).to(dev)
    tgt = (R.gen_uniform("nolib:t", (B, T, 1), 3) * R.prefix_mask(lengths, T)).to(dev)
    raw = which.startswith("raw_")
    if raw:
        # raw windows -> valence: the window encoder (CNN k=2 + max-pool + Highway + Dropout(0.3), modality concatenation, fusion layer:
        # transformer/SFT/models.py:113-142, MFT twin) in front of the sequence model — its Highway combine, dropout and torch.cat were
        # library kernels until round 4
        MM = mta().models
        mods, dims, wl = ["acoustic", "linguistic"], {"acoustic": 88, "linguistic": 300}, {"acoustic": 6, "linguistic": 9}
        if which == "raw_sft":
            model = MM.MultiCNNTransformer(mods, dims, device=dev)
            model.Transformer = MT.NLPTransformer(512, embed_dim=128, h=8, N=2, device=dev)
        else:
            model = MM.MultiCNNTransformerMFT(mods, dims, {"acoustic": 88, "linguistic": 300}, device=dev)
            model.Transformer = MT.MultiTransformer(mods, {"acoustic": 88, "linguistic": 300}, N=2, device=dev)
        model.train()
        x = {m: R.gen_normal("nolib:raw:" + m, (B, T, wl[m], dims[m]), 3).to(dev) for m in mods}
    elif which == "mft":
        model = MT.MultiTransformer(R.MODS_AVL, R.EMBED_AVL, device=dev).train()
        x = {m: R.gen_normal("nolib:" + m, (B, T, R.EMBED_AVL[m]), 3).to(dev) for m in R.MODS_AVL}
    else:
        model = MT.NLPTransformer(512, embed_dim=128, h=8, device=dev).train()
        x = torch.tanh(R.gen_normal("nolib:x", (B, T, 512), 3)).to(dev)
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        F.mse_sum_loss_backward(model(x, lengths, mask) if raw else model(x, mask, lengths), tgt, sum(lengths))

    names = _device_kernel_names(step)
    if names is None:
        pytest.skip("torch.profiler reports no device kernels on this box")
    ours = [n for n in names if "kernel" in n]
    assert len(ours) > 20, names[:10]
    assert _library_kernels(names) == [], "library kernels in a %s train step: %s" % (which, _library_kernels(names))
    for p in params:
        assert p.grad is None or torch.isfinite(p.grad).all()
