"""Train-mode (dropout) tests of the fused encoder path on the GPU.

torch's dropout stream cannot be reproduced in HIP, so train-mode parity is established in two ways:
  * REPLAY (numerical): the exact keep-masks the kernels used are rebuilt with mmt_debug_dropout_mask and the
    oracle is run with those masks as explicit multipliers at the reference's four dropout sites
    (attention probabilities :33, FFN hidden :20, both sublayer outputs :104).  Same tolerances as eval mode.
  * STATISTICAL: drop fraction = p within 4 sigma; kept values scaled by 1/(1-p); masks are deterministic in
    (seed, stream, index) and independent across streams/seeds.
"""
import numpy as np
import pytest
import torch

import oracle
import recipe as R
from test_gpu_parity import OUT_RTOL, RELU_GRAD_RTOL, CCC_MIN, _report, mta
from conftest import grad_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _masks(dev, p, seed, n_layers, B, T, d, h, f):
    F = mta().functional
    Tp, DP, FP = -(-T // 32) * 32, -(-d // 64) * 64, -(-f // 64) * 64
    M = B * T
    out = []
    for l in range(n_layers):
        ka, sa = F.dropout_mask(p, seed, 4 * l + 0, B * h * Tp * Tp, dev, attn_Tp=Tp)
        k0, s0 = F.dropout_mask(p, seed, 4 * l + 1, M * DP, dev)
        kf, sf = F.dropout_mask(p, seed, 4 * l + 2, M * FP, dev)
        k1, s1 = F.dropout_mask(p, seed, 4 * l + 3, M * DP, dev)
        out.append({
            "attn": (ka.reshape(B, h, Tp, Tp)[:, :, :T, :T].double() * sa).cpu(),
            "sub0": (k0.reshape(B, T, DP)[:, :, :d].double() * s0).cpu(),
            "ffn": (kf.reshape(B, T, FP)[:, :, :f].double() * sf).cpu(),
            "sub1": (k1.reshape(B, T, DP)[:, :, :d].double() * s1).cpu(),
        })
    return out


@pytest.mark.parametrize("d,h,n,B,T,lengths,p", [(128, 8, 2, 3, 50, [50, 31, 6], 0.1), (40, 4, 2, 2, 33, [33, 9], 0.25),
                                                 (256, 8, 1, 2, 70, [70, 64], 0.1),
                                                 (128, 8, 1, 2, 300, [300, 170], 0.1)])       # one-kernel attention backward
def test_train_mode_replay_with_kernel_masks(dev, d, h, n, B, T, lengths, p):
    MT = mta().multiTransformer
    F = mta().functional
    enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, R.D_FF, p), p), n)
    p32 = R.gen_params(R.shapes_of(enc.state_dict()), 31)
    enc.load_state_dict(p32)
    enc = enc.to(dev).train()
    x = R.gen_normal("drop:x%d" % d, (B, T, d), 31)
    g = R.gen_normal("drop:g%d" % d, (B, T, d), 31)
    mask = R.prefix_mask(lengths, T)
    seed = 123456789 + d
    flat = torch.cat([q.reshape(-1) for q in enc.flat_parameters()]).detach().requires_grad_()
    xg = x.to(dev).requires_grad_()
    y = F.encoder_stack(xg, mask.to(dev), flat, h, R.D_FF, n, dropout_p=p, seed=seed)
    (y * g.to(dev)).sum().backward()
    drops = _masks(dev, p, seed, n, B, T, d, h, R.D_FF)
    frac = float(1 - (drops[0]["ffn"] > 0).double().mean())
    assert abs(frac - p) < 4 * np.sqrt(p * (1 - p) / drops[0]["ffn"].numel()) + 2e-5

    pd = {k: v.double().clone().requires_grad_() for k, v in p32.items()}
    xd = x.double().clone().requires_grad_()
    ref = oracle.encoder_stack(pd, "", xd, mask.double(), h, drops)
    (ref * g.double()).sum().backward()
    tag = "train d%d p%.2f" % (d, p)
    assert _report(tag + " out", y.detach().cpu(), ref.detach()) < OUT_RTOL
    assert mta().eval_ccc(ref.detach().numpy(), y.detach().cpu().numpy()) >= CCC_MIN
    assert _report(tag + " dx", xg.grad.cpu(), xd.grad) < RELU_GRAD_RTOL
    ref_flat = torch.cat([pd[k].grad.reshape(-1) for k in _flat_names(n)])
    assert _report(tag + " dparams", flat.grad.cpu(), ref_flat) < RELU_GRAD_RTOL
    # and it must differ from the eval-mode result (dropout really happened)
    with torch.no_grad():
        y_eval = F.encoder_stack(x.to(dev), mask.to(dev), flat.detach(), h, R.D_FF, n)
    assert float((y_eval - y.detach()).abs().max()) > 1e-2


def _flat_names(n):
    names = []
    for i in range(n):
        L = "layers.%d." % i
        for j in range(4):
            names += [L + "self_attn.linears.%d.weight" % j, L + "self_attn.linears.%d.bias" % j]
        names += [L + "feed_forward.w_1.weight", L + "feed_forward.w_1.bias", L + "feed_forward.w_2.weight", L + "feed_forward.w_2.bias",
                  L + "sublayer.0.norm.a_2", L + "sublayer.0.norm.b_2", L + "sublayer.1.norm.a_2", L + "sublayer.1.norm.b_2"]
    return names + ["norm.a_2", "norm.b_2"]


def test_mask_statistics_and_determinism(dev):
    F = mta().functional
    n = 1 << 20
    for p in (0.1, 0.2, 0.5):
        k1, s1 = F.dropout_mask(p, 42, 3, n, dev)
        k2, _ = F.dropout_mask(p, 42, 3, n, dev)
        k3, _ = F.dropout_mask(p, 43, 3, n, dev)
        k4, _ = F.dropout_mask(p, 42, 4, n, dev)
        assert torch.equal(k1, k2)                                  # pure function of (p, seed, stream, index)
        frac = 1 - k1.float().mean().item()
        assert abs(frac - p) < 4 * np.sqrt(p * (1 - p) / n) + 2e-5
        assert abs(s1 - 1 / (1 - round(p * 65536) / 65536)) < 1e-6
        for other in (k3, k4):                                      # independent streams: agreement = p^2 + (1-p)^2
            agree = (k1 == other).float().mean().item()
            assert abs(agree - (p * p + (1 - p) * (1 - p))) < 6e-3
        pairs = k1.reshape(-1, 2).float()                           # the two halves of one hash word are uncorrelated
        c = np.corrcoef(pairs[:, 0].cpu().numpy(), pairs[:, 1].cpu().numpy())[0, 1]
        assert abs(c) < 6e-3


def test_module_train_mode_runs_and_reseeds(dev):
    MT = mta().multiTransformer
    enc = MT.Encoder(MT.EncoderLayer(128, MT.MultiHeadedAttention(8, 128), MT.PositionwiseFeedForward(128, 128, 0.1), 0.1), 2).to(dev)
    enc.train()
    x = R.gen_normal("drop:mod", (2, 40, 128), 5).to(dev)
    mask = torch.ones(2, 40, 1, device=dev)
    y1, y2 = enc(x, mask), enc(x, mask)
    assert torch.isfinite(y1).all() and float((y1 - y2).abs().max()) > 1e-3      # a fresh mask per call
    enc.eval()
    assert torch.equal(enc(x, mask), enc(x, mask))


@pytest.mark.parametrize("B,T,d,h,lengths,p", [(2, 300, 128, 8, [300, 170], 0.1),      # d_k 16, one-kernel backward
                                                (2, 70, 256, 8, [70, 33], 0.1),          # d_k 32, two-kernel backward
                                                (3, 33, 40, 4, [33, 20, 1], 0.25),       # d_k 10 (padded to 16), ragged tail tile
                                                (1, 520, 128, 8, [520], 0.1),            # 17 key tiles: two-kernel backward at d_k 16
                                                (2, 70, 256, 4, [70, 41], 0.1),          # d_k 64: two feature blocks
                                                # the one-kernel backward's corner cases in TRAIN mode (attn_bwd_pair.h): 9 key tiles (an odd count: the
                                                # last wave owns one key tile, a blank query tile pads the sweep) with a one-key last tile; 15 tiles with a
                                                # ragged tail; 16 full tiles; a padded head (d_k 10) at 10 tiles
                                                (2, 257, 128, 8, [257, 130], 0.1), (1, 470, 64, 4, [470], 0.1), (1, 512, 128, 8, [512], 0.1),
                                                (2, 300, 40, 4, [300, 41], 0.1)])
def test_standalone_attention_train_mode_replay(dev, B, T, d, h, lengths, p):
    """attention() / MultiHeadedAttention outside the fused stack apply nn.Dropout to p_attn in train mode
    (transformer/MFT/multiTransformer.py:31-33): the stored bit masks (stream 0) are extracted and replayed through the oracle,
    forward and all three input gradients."""
    F = mta().functional
    dk = d // h
    q, k, v = (R.gen_normal("sdpa_drop:%s%d" % (n, d), (B, T, d), 41) for n in "qkv")
    g = R.gen_normal("sdpa_drop:g%d" % d, (B, T, d), 41)
    mask = R.prefix_mask(lengths, T)
    seed = 987654321 + T
    gl = [t.to(dev).requires_grad_() for t in (q, k, v)]
    out = F.sdpa(gl[0], gl[1], gl[2], mask.to(dev), h, dropout_p=p, seed=seed)
    (out * g.to(dev)).sum().backward()
    Tp = -(-T // 32) * 32
    keep, scale = F.dropout_mask(p, seed, 0, B * h * Tp * Tp, dev, attn_Tp=Tp)
    drop = (keep.reshape(B, h, Tp, Tp)[:, :, :T, :T].double() * scale).cpu()
    frac = 1 - float((drop > 0).double().mean())
    assert abs(frac - p) < 4 * np.sqrt(p * (1 - p) / drop.numel()) + 2e-5

    def split(z):
        return z.reshape(B, T, h, dk).permute(0, 2, 1, 3)
    leaves = [t.double().clone().requires_grad_() for t in (q, k, v)]
    ref, _ = oracle.scaled_dot_attention(split(leaves[0]), split(leaves[1]), split(leaves[2]), mask.double().unsqueeze(1), drop)
    ref = ref.permute(0, 2, 1, 3).reshape(B, T, d)
    (ref * g.double()).sum().backward()
    tag = "sdpa train T%d d%d p%.2f" % (T, d, p)
    assert _report(tag + " out", out.detach().cpu(), ref.detach()) < OUT_RTOL
    for name, a, b in zip("qkv", gl, leaves):
        assert _report(tag + " d" + name, a.grad.cpu(), b.grad) < 4e-2, name
    # eval (dropout None / p = 0) differs: dropout really happened
    with torch.no_grad():
        assert float((F.sdpa(gl[0], gl[1], gl[2], mask.to(dev), h) - out).abs().max()) > 1e-3


def test_attention_mask_statistics(dev):
    """the stored attention masks: drop fraction p (exact to 2^-16), rows/columns and streams independent, pure function of the seed"""
    F = mta().functional
    Tp, nbh = 256, 6
    n = nbh * Tp * Tp
    for p in (0.1, 0.25, 0.5):
        k1, s1 = F.dropout_mask(p, 42, 0, n, dev, attn_Tp=Tp)
        k2, _ = F.dropout_mask(p, 42, 0, n, dev, attn_Tp=Tp)
        k3, _ = F.dropout_mask(p, 43, 0, n, dev, attn_Tp=Tp)
        k4, _ = F.dropout_mask(p, 42, 4, n, dev, attn_Tp=Tp)
        assert torch.equal(k1, k2)
        frac = 1 - k1.float().mean().item()
        pq = round(p * 4096) / 4096                                 # the attention stream resolves p to 12 bits (common.h make_drop)
        assert abs(pq - p) < 1.3e-4 and abs(frac - pq) < 4 * np.sqrt(p * (1 - p) / n) + 2e-5
        assert abs(s1 - 1 / (1 - pq)) < 1e-6                        # kept values are scaled by the probability actually used
        for other in (k3, k4):
            agree = (k1 == other).float().mean().item()
            assert abs(agree - (p * p + (1 - p) * (1 - p))) < 6e-3
        m = k1.reshape(nbh, Tp, Tp).float()
        # neighbours along the query axis (bits of one generator word), along the key axis (adjacent words) and across heads
        for a, b in ((m[:, :-1, :], m[:, 1:, :]), (m[:, :, :-1], m[:, :, 1:]), (m[:-1], m[1:])):
            c = np.corrcoef(a.reshape(-1).cpu().numpy(), b.reshape(-1).cpu().numpy())[0, 1]
            assert abs(c) < 6e-3
        # every query row and key column drops about p of its entries
        assert float((1 - m.mean(dim=2)).sub(p).abs().max()) < 6 * np.sqrt(p * (1 - p) / Tp)
        assert float((1 - m.mean(dim=1)).sub(p).abs().max()) < 6 * np.sqrt(p * (1 - p) / Tp)


def test_multi_headed_attention_module_train_mode(dev):
    """MultiHeadedAttention used on its own in train mode (the reference's default dropout 0.1): runs, draws a fresh mask per call,
    passes gradients; eval mode is deterministic"""
    MT = mta().multiTransformer
    mha = MT.MultiHeadedAttention(8, 128).to(dev).train()
    x = R.gen_normal("mha_drop:x", (2, 40, 128), 5).to(dev).requires_grad_()
    mask = R.prefix_mask([40, 13], 40).to(dev)
    y1, y2 = mha(x, x, x, mask), mha(x, x, x, mask)
    assert torch.isfinite(y1).all() and float((y1 - y2).detach().abs().max()) > 1e-4
    y1.sum().backward()
    assert torch.isfinite(x.grad).all() and all(torch.isfinite(p.grad).all() for p in mha.parameters())
    ctx, _ = MT.attention(x.detach().reshape(2, 40, 8, 16).transpose(1, 2), x.detach().reshape(2, 40, 8, 16).transpose(1, 2),
                          x.detach().reshape(2, 40, 8, 16).transpose(1, 2), mask, torch.nn.Dropout(0.1).train())
    assert ctx.shape == (2, 8, 40, 16) and torch.isfinite(ctx).all()
    mha.eval()
    assert torch.equal(mha(x, x, x, mask), mha(x, x, x, mask))


def test_device_resident_seed_fresh_masks_per_graph_replay(dev):
    """A train step captured into a hipGraph must draw NEW masks at every replay (the reference's nn.Dropout does every step:
    transformer/MFT/multiTransformer.py:17,45,101, transformer/SFT/train.py:131-141).  Under capture the encoder takes its seed from
    device memory (mmt_encoder_forward_devseed): the seed state is peeked before each replay, the replays must differ from each other,
    and the masks rebuilt from the peeked value reproduce that replay — forward, input gradient and parameter gradients — through the oracle."""
    MT = mta().multiTransformer
    d, h, n, B, T, lengths, p = 128, 8, 2, 2, 300, [300, 170], 0.1            # T = 300: the one-kernel attention backward
    enc = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, R.D_FF, p), p), n)
    p32 = R.gen_params(R.shapes_of(enc.state_dict()), 47)
    enc.load_state_dict(p32)
    enc = enc.to(dev).train()
    params = list(enc.parameters())
    x = R.gen_normal("devseed:x", (B, T, d), 47)
    g = R.gen_normal("devseed:g", (B, T, d), 47)
    mask = R.prefix_mask(lengths, T)
    xg = x.to(dev).requires_grad_()
    gd, md = g.to(dev), mask.to(dev)
    out = {}

    def step():
        for q in params:
            q.grad = None
        xg.grad = None
        y = enc(xg, md)
        (y * gd).sum().backward()
        out["y"] = y.detach()                                                 # no reference to the step's autograd graph survives the step

    with pytest.raises(RuntimeError, match="warm-up"):                        # a capture needs the seed state to exist beforehand
        fresh = MT.Encoder(MT.EncoderLayer(d, MT.MultiHeadedAttention(h, d), MT.PositionwiseFeedForward(d, R.D_FF, p), p), 1).to(dev).train()
        gtmp = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gtmp):
            fresh(xg.detach(), md)
    torch.cuda.synchronize()
    from multimodal_transformer_amd import graphs
    graph, _ = graphs.capture_step(step, warmup=2)                            # eager side-stream warm-up (by-value seeds, creates the DeviceSeed), then the capture
    results = []
    for _ in range(2):
        seed = mta()._lib.device_seed(enc, 1).peek()                                           # the seed this replay is going to use
        graph.replay()
        torch.cuda.synchronize()
        assert mta()._lib.device_seed(enc, 1).peek() != seed                                   # the launch advanced it
        results.append((seed, out["y"].detach().clone(), xg.grad.detach().clone(),
                        torch.cat([q.grad.reshape(-1) for q in enc.flat_parameters()]).clone()))
    assert float((results[0][1] - results[1][1]).abs().max()) > 1e-2          # different masks in the two replays
    seed, y, dx, dflat = results[1]
    drops = _masks(dev, p, seed, n, B, T, d, h, R.D_FF)
    pd = {k: v.double().clone().requires_grad_() for k, v in p32.items()}
    xd = x.double().clone().requires_grad_()
    ref = oracle.encoder_stack(pd, "", xd, mask.double(), h, drops)
    (ref * g.double()).sum().backward()
    assert _report("devseed out", y.cpu(), ref.detach()) < OUT_RTOL
    assert _report("devseed dx", dx.cpu(), xd.grad) < RELU_GRAD_RTOL
    ref_flat = torch.cat([pd[k].grad.reshape(-1) for k in _flat_names(n)])
    assert _report("devseed dparams", dflat.cpu(), ref_flat) < RELU_GRAD_RTOL
