"""Host batching (CPU): ``batching.generate_train_batches`` / ``make_batch`` against fixtures generated from the reference's
``generateTrainBatch`` (transformer/SFT/train.py:52-106; tests/golden/make_golden_batching.py): chunk order, the stable
sort by length, the cut to the batch's longest sequence, the prefix mask and the sorted lengths are index / integer work and
must be bit-exact, for padded arrays and for ragged per-sequence lists alike."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from multimodal_transformer_amd import batching


@pytest.fixture(scope="module")
def fx():
    return load_golden("batching")


def _inputs(fx):
    mods = sorted(k[3:] for k in fx if k.startswith("in:") and k != "in:target")
    data = {m: fx["in:" + m] for m in mods}
    return mods, data, fx["in:target"], [int(v) for v in fx["lengths"]]


@pytest.mark.parametrize("bs", [3, 25, 1])
@pytest.mark.parametrize("ragged", [False, True])
def test_batches_equal_the_references(fx, bs, ragged):
    mods, data, target, lengths = _inputs(fx)
    if ragged:                                              # the same sequences as unpadded per-sequence arrays
        data = {m: [v[i, :lengths[i]] for i in range(len(lengths))] for m, v in data.items()}
        target = [target[i, :lengths[i]] for i in range(len(lengths))]
    batches = list(batching.generate_train_batches(data, target, lengths, batch_size=bs))
    assert len(batches) == int(fx["bs%d:n" % bs])
    for k, (d, tg, mask, ls) in enumerate(batches):
        pre = "bs%d:%d:" % (bs, k)
        assert ls == [int(v) for v in fx[pre + "lengths"]]
        assert tg.dtype == torch.float32 and mask.dtype == torch.float32
        np.testing.assert_array_equal(mask.numpy(), fx[pre + "mask"])
        np.testing.assert_array_equal(tg.numpy(), fx[pre + "target"])
        for m in mods:
            np.testing.assert_array_equal(d[m].numpy(), fx[pre + m])


def test_prefix_mask_and_stable_sort():
    assert batching.sort_by_length([5, 9, 3, 9, 1, 7, 7, 2]) == [1, 3, 5, 6, 0, 2, 7, 4]      # ties keep their order
    m = batching.prefix_mask([3, 0, 1], 3)
    assert m.shape == (3, 3, 1) and m[:, :, 0].tolist() == [[1, 1, 1], [0, 0, 0], [1, 0, 0]]


def test_empty_and_single():
    d, tg, mask, ls = batching.make_batch({"a": np.zeros((1, 4, 2, 3), np.float32)}, np.zeros((1, 4), np.float32), [2])
    assert d["a"].shape == (1, 2, 2, 3) and tg.shape == (1, 2, 1) and mask.sum() == 2 and ls == [2]
    assert list(batching.generate_train_batches({"a": np.zeros((0, 1, 1, 1), np.float32)}, np.zeros((0, 1), np.float32), [])) == []


def _ragged_dataset(seed=5, n=11):
    rng = np.random.RandomState(seed)
    lengths = [int(v) for v in rng.randint(1, 9, size=n)]
    lengths[min(3, n - 1)] = 8
    data = {"acoustic": [rng.randn(L + 2, 3, 5).astype(np.float32) for L in lengths],          # (longer than lengths[i]: the tail is cut)
            "linguistic": [rng.randn(L + 2, 2, 7).astype(np.float32) for L in lengths]}
    target = [rng.rand(L + 2).astype(np.float32) for L in lengths]
    return data, target, lengths


def test_packed_file_round_trip_and_loader_matches_generate_train_batches(tmp_path):
    """the packed on-disk format: what the loader yields from the memory-mapped file is bit for bit what
    generate_train_batches — the restatement of the reference's generateTrainBatch pinned by tests/golden/batching.npz — builds
    from the same sequences in memory"""
    from multimodal_transformer_amd import batching as Bt
    data, target, lengths = _ragged_dataset()
    path = str(tmp_path / "send.mmtpack")
    h = Bt.pack_dataset(path, data, target, lengths)
    assert h["n"] == len(lengths) and h["total_windows"] == sum(lengths)
    ds = Bt.PackedDataset(path)
    assert ds.lengths == lengths and ds.mods == ["acoustic", "linguistic"] and ds.shape["linguistic"] == (2, 7)
    for i in (0, 3, len(lengths) - 1):
        np.testing.assert_array_equal(ds.windows["acoustic"][ds.start[i]:ds.start[i + 1]], data["acoustic"][i][:lengths[i]])
        np.testing.assert_array_equal(ds.target[ds.start[i]:ds.start[i + 1]], target[i][:lengths[i]])
    cut = {m: [s[:L] for s, L in zip(v, lengths)] for m, v in data.items()}
    ref = list(Bt.generate_train_batches(cut, [t[:L] for t, L in zip(target, lengths)], lengths, batch_size=4))
    got = list(Bt.PackedLoader(ds, batch_size=4))
    assert len(got) == len(ref) == len(Bt.PackedLoader(ds, batch_size=4)) == 3
    for (d0, t0, m0, l0), (d1, t1, m1, l1) in zip(ref, got):
        assert l0 == l1
        assert torch.equal(t0, t1) and torch.equal(m0, m1)
        for mod in d0:
            assert torch.equal(d0[mod], d1[mod]), mod
    # a second epoch over the same loader (the staging buffers are reused) gives the same batches
    again = list(Bt.PackedLoader(ds, batch_size=4, slots=1))
    assert all(torch.equal(a[1], b[1]) and all(torch.equal(a[0][m], b[0][m]) for m in a[0]) for a, b in zip(got, again))


def test_packed_file_rejects_garbage(tmp_path):
    from multimodal_transformer_amd import batching as Bt
    bad = tmp_path / "x.bin"
    bad.write_bytes(b"not a pack file at all")
    with pytest.raises(ValueError):
        Bt.PackedDataset(str(bad))
    data, target, lengths = _ragged_dataset(n=3)
    data["acoustic"][1] = data["acoustic"][1][:, :2]                       # a window of another shape
    with pytest.raises(ValueError):
        Bt.pack_dataset(str(tmp_path / "y.mmtpack"), data, target, lengths)
