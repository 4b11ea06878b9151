"""Host batching of narrative sequences: the input contract of the sequence models.

Restates what the reference's training loop builds per batch (transformer/SFT/train.py:52-106: ``chunks``,
``generateInputChunkHelper``, ``generateTrainBatch``; the MFT / B2-Trans drivers hold the same code): sequences are taken in
order in chunks of ``batch_size`` (25 in the reference; nothing is shuffled, ``:80``), each chunk is sorted by length, longest
first, with a STABLE sort (ties keep their order, ``:61``), cut to the chunk's longest sequence and given a prefix mask
``mask[b, :length[b]] = 1`` of shape (B, T, 1) (``:99-104``).  Windows are (W, d_raw) blocks per time step; targets one
valence per window.

Unlike the reference, which round-trips nested Python lists through ``torch.tensor`` for every batch (``:68``), the padded
arrays are converted once and each batch is an index gather.
"""
import numpy as np
import torch


def sort_by_length(lengths):
    """Indices that order a chunk longest first, ties in their original order (the reference's stable list sort)."""
    return sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))


def prefix_mask(lengths, T, device=None):
    """(B, T, 1) float32 mask: ones over the first lengths[b] windows (transformer/SFT/train.py:101-104)."""
    t = torch.arange(T).unsqueeze(0)
    m = (t < torch.as_tensor(list(lengths)).unsqueeze(1)).to(torch.float32).unsqueeze(2)
    return m.to(device) if device is not None else m


def _as_padded(x, n, t_max):
    """x: array (n, T, ...) or a list of per-sequence arrays (T_i, ...) -> float32 tensor (n, t_max, ...), zero padded."""
    if isinstance(x, torch.Tensor):
        return x.to(torch.float32)
    if isinstance(x, np.ndarray) and x.dtype != object:
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    seqs = [np.asarray(s, dtype=np.float32) for s in x]
    assert len(seqs) == n
    out = np.zeros((n, t_max) + seqs[0].shape[1:], dtype=np.float32)
    for i, s in enumerate(seqs):
        out[i, :min(len(s), t_max)] = s[:t_max]
    return torch.from_numpy(out)


def make_batch(data, target, lengths, index=None, device=None):
    """One batch in the reference's form.

    data: {modality: (n, T, W, d_raw) array / tensor, or a list of n ragged (T_i, W, d_raw) arrays}; target: (n, T) or ragged
    (T_i,) list; lengths: n ints; index: which sequences make up the batch (default: all).
    -> (data {modality: (B, Tmax, W, d_raw)}, target (B, Tmax, 1), mask (B, Tmax, 1), lengths sorted longest first),
    Tmax = the batch's longest sequence — what ``generateTrainBatch`` yields (``:106``).
    """
    n = len(lengths)
    t_all = int(max(lengths)) if n else 0
    index = list(range(n)) if index is None else list(index)
    ls = [int(lengths[i]) for i in index]
    order = [index[j] for j in sort_by_length(ls)]
    ls_sorted = sorted(ls, reverse=True)
    t_max = ls_sorted[0] if ls_sorted else 0
    sel = torch.as_tensor(order, dtype=torch.long)
    out = {}
    for mod, x in data.items():
        xp = _as_padded(x, n, t_all)
        out[mod] = xp.index_select(0, sel)[:, :t_max].contiguous()
    tg = _as_padded(target, n, t_all).index_select(0, sel)[:, :t_max].unsqueeze(2).contiguous()
    mask = prefix_mask(ls_sorted, t_max)
    if device is not None:
        out = {m: v.to(device, non_blocking=True) for m, v in out.items()}
        tg, mask = tg.to(device, non_blocking=True), mask.to(device, non_blocking=True)
    return out, tg, mask, ls_sorted


def generate_train_batches(data, target, lengths, batch_size=25, device=None):
    """The batches of one epoch, in the reference's order (transformer/SFT/train.py:74-106)."""
    n = len(lengths)
    t_all = int(max(lengths)) if n else 0
    padded = {m: _as_padded(x, n, t_all) for m, x in data.items()}        # converted once, not per batch
    tgt = _as_padded(target, n, t_all)
    for lo in range(0, n, batch_size):
        yield make_batch(padded, tgt, lengths, index=range(lo, min(lo + batch_size, n)), device=device)


def evaluate(model, data, target, lengths, batch_size=1, device=None):
    """Mean and spread of the per-sequence CCC and the loss per window, as ``evaluate`` reports them
    (transformer/SFT/train.py:196-254).  The reference evaluates ONE sequence at a time (``batch_size=1``, ``:210-214``), and that
    is the default here, because it matters: the reference's mask blanks query rows only and never masks keys
    (SFT/multiTransformer.py:29-30), so in a padded batch every shorter sequence attends to the padding windows (whose keys and
    values are non-zero after the embed bias and LayerNorm) and its valence differs from the one-at-a-time result.  A larger
    ``batch_size`` is faster and gives the reference's numbers only when the sequences of a batch have equal length.
    The loss comes from the fused loss kernel (``mmt_mse_sum_forward``), the CCC of every sequence is reduced on the device
    (``mmt_ccc_forward``); sequences shorter than 2 windows have no CCC and are left out of the mean (the reference's NaN)."""
    from . import metrics
    from .functional import mse_sum_loss
    cccs, losses, nwin = [], [], 0
    was_training = model.training
    model.eval()
    with torch.no_grad():
        for d, tg, mask, ls in generate_train_batches(data, target, lengths, batch_size, device):
            out = model(d, ls, mask)
            losses.append(mse_sum_loss(out, tg, 1.0))           # stays on the device: one host read-back at the end
            nwin += sum(ls)
            c = metrics.batched_ccc(out, tg, ls)
            cccs.append((c, ls))
    model.train(was_training)
    loss = float(torch.stack(losses).double().sum()) if losses else 0.0
    vals = [float(v) for c, ls in cccs for v, L in zip(c.tolist(), ls) if L > 1]
    return {"loss": loss / max(nwin, 1), "ccc": float(np.mean(vals)) if vals else float("nan"),
            "ccc_std": float(np.std(vals)) if vals else float("nan"), "max_ccc": max(vals) if vals else float("nan"),
            "per_sequence_ccc": vals}
