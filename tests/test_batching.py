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
