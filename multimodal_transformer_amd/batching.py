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


def _to_device(t, device):
    """host -> device through page-locked memory (an asynchronous copy from pageable memory is staged by the runtime and blocks)"""
    if torch.device(device).type == "cuda" and torch.cuda.is_available() and not t.is_pinned():
        t = t.pin_memory()
    return t.to(device, non_blocking=True)


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
        out = {m: _to_device(v, device) for m, v in out.items()}
        tg, mask = _to_device(tg, device), _to_device(mask, device)
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


# ---- packed on-disk format + pinned loader ---------------------------------------------------------------------------
# The reference keeps a data set as nested Python lists and converts every batch with torch.tensor(list) (transformer/SFT/train.py:59-69:
# the host hot spot of its training loop).  A packed file holds the same sequences once, unpadded, as flat fp32 arrays that a loader
# memory-maps; a batch is then a few contiguous row-range copies into a page-locked staging buffer and ONE asynchronous copy per tensor.
#   file  = magic "MMTPACK1" | uint64 header length | JSON header | 64-byte aligned fp32 payload
#   header: {"n", "lengths": [...], "mods": {mod: {"W", "D", "offset"}}, "target_offset", "total_windows"}  (offsets in bytes from file start)
#   payload: per modality [total_windows][W][D], windows of sequence i at rows start[i] .. start[i] + lengths[i]; then targets [total_windows]
_PACK_MAGIC = b"MMTPACK1"


def pack_dataset(path, data, target, lengths):
    """Write {modality: n sequences of (T_i, W, d_raw) windows}, per-window targets and lengths as one packed file."""
    import json
    lengths = [int(v) for v in lengths]
    n, total = len(lengths), int(sum(lengths))
    mods, arrays = {}, []
    for mod, x in data.items():
        seqs = [np.asarray(x[i], dtype=np.float32)[:lengths[i]] for i in range(n)]
        if any(s.ndim != 3 or s.shape[0] != lengths[i] for i, s in enumerate(seqs)):
            raise ValueError("pack_dataset: modality %r needs (T_i, W, d_raw) windows with T_i >= lengths[i]" % mod)
        W, D = seqs[0].shape[1:]
        if any(s.shape[1:] != (W, D) for s in seqs):
            raise ValueError("pack_dataset: modality %r has windows of different shapes" % mod)
        mods[mod] = {"W": int(W), "D": int(D), "offset": 0}
        arrays.append((mod, seqs))
    tgt = [np.asarray(target[i], dtype=np.float32).reshape(-1)[:lengths[i]] for i in range(n)]
    if any(len(t) != lengths[i] for i, t in enumerate(tgt)):
        raise ValueError("pack_dataset: a target is shorter than its sequence")
    header = {"n": n, "lengths": lengths, "mods": mods, "target_offset": 0, "total_windows": total}

    def layout(hlen):
        off = (len(_PACK_MAGIC) + 8 + hlen + 63) // 64 * 64
        for mod in mods:
            mods[mod]["offset"] = off
            off += total * mods[mod]["W"] * mods[mod]["D"] * 4
            off = (off + 63) // 64 * 64
        header["target_offset"] = off
        return off + total * 4
    layout(0)
    hjson = json.dumps(header).encode()
    size = layout(len(hjson) + 64)                       # offsets grow by a few digits at most: reserve, then pad the header with spaces
    hjson = json.dumps(header).encode().ljust(len(hjson) + 64)
    with open(path, "wb") as fh:
        fh.write(_PACK_MAGIC)
        fh.write(np.uint64(len(hjson)).tobytes())
        fh.write(hjson)
        for mod, seqs in arrays:
            fh.seek(mods[mod]["offset"])
            for s_ in seqs:
                fh.write(np.ascontiguousarray(s_).tobytes())
        fh.seek(header["target_offset"])
        for t in tgt:
            fh.write(np.ascontiguousarray(t).tobytes())
        fh.truncate(size)
    return header


class PackedDataset:
    """Memory-mapped view of a packed file: ``windows[mod]`` is (total_windows, W, D), ``target`` (total_windows,), ``start[i]`` the first row of sequence i."""

    def __init__(self, path):
        import json
        with open(path, "rb") as fh:
            if fh.read(len(_PACK_MAGIC)) != _PACK_MAGIC:
                raise ValueError("%s is not a packed data set (bad magic)" % path)
            hlen = int(np.frombuffer(fh.read(8), dtype=np.uint64)[0])
            h = json.loads(fh.read(hlen).decode())
        self.path, self.n, self.lengths, self.total = path, int(h["n"]), [int(v) for v in h["lengths"]], int(h["total_windows"])
        if len(self.lengths) != self.n or sum(self.lengths) != self.total:
            raise ValueError("%s: header lengths do not add up" % path)
        self.start = np.concatenate([[0], np.cumsum(self.lengths)]).astype(np.int64)
        self.mods = list(h["mods"])
        self.shape = {m: (int(v["W"]), int(v["D"])) for m, v in h["mods"].items()}
        self.windows = {m: np.memmap(path, dtype=np.float32, mode="r", offset=int(v["offset"]), shape=(self.total,) + self.shape[m])
                        for m, v in h["mods"].items()}
        self.target = np.memmap(path, dtype=np.float32, mode="r", offset=int(h["target_offset"]), shape=(self.total,))

    def __len__(self):
        return self.n


class PackedLoader:
    """Batches of a PackedDataset in the reference's order and form (``generate_train_batches``: chunks of ``batch_size`` in file order,
    each sorted longest first, padded to its longest sequence, prefix mask), assembled in page-locked staging buffers — two sets,
    used alternately — and copied to the device on a copy stream of its own, so the copy of batch i + 1 overlaps the step on batch i.
    Iterating yields (data {mod: (B,T,W,D)}, target (B,T,1), mask (B,T,1), lengths); with ``device=None`` the tensors stay on the host."""

    def __init__(self, dataset, batch_size=25, device=None, slots=2):
        self.ds, self.batch_size = dataset, int(batch_size)
        self.device = torch.device(device) if device is not None else None
        self.cuda = self.device is not None and self.device.type == "cuda" and torch.cuda.is_available()
        self.slots = max(1, int(slots))
        t_max = max(dataset.lengths) if dataset.n else 0
        B = min(self.batch_size, max(dataset.n, 1))

        def host(*shape):
            return torch.zeros(*shape, dtype=torch.float32, pin_memory=self.cuda)
        self._stage = [{"data": {m: host(B, t_max, *dataset.shape[m]) for m in dataset.mods}, "target": host(B, t_max, 1),
                        "mask": host(B, t_max, 1), "event": None} for _ in range(self.slots)]
        self._copy_stream = torch.cuda.Stream(self.device) if self.cuda else None

    def __len__(self):
        return (self.ds.n + self.batch_size - 1) // self.batch_size

    def _assemble(self, slot, lo, hi):
        ds = self.ds
        ls = [ds.lengths[i] for i in range(lo, hi)]
        order = [lo + j for j in sort_by_length(ls)]
        ls_sorted = [ds.lengths[i] for i in order]
        B, T = len(order), (ls_sorted[0] if ls_sorted else 0)
        st = self._stage[slot]
        if st["event"] is not None:
            st["event"].synchronize()                    # the device copy that last read this staging set has completed
        out = {}
        for m in ds.mods:
            buf = st["data"][m].numpy()                  # a view of the page-locked tensor: the rows below are copied straight from the map
            for b, i in enumerate(order):
                L = ds.lengths[i]
                np.copyto(buf[b, :L], ds.windows[m][ds.start[i]:ds.start[i] + L])
                buf[b, L:T] = 0.0
            out[m] = st["data"][m][:B, :T]
        tg, mk = st["target"].numpy(), st["mask"].numpy()
        for b, i in enumerate(order):
            L = ds.lengths[i]
            np.copyto(tg[b, :L, 0], ds.target[ds.start[i]:ds.start[i] + L])
            tg[b, L:T] = 0.0
            mk[b, :L] = 1.0
            mk[b, L:T] = 0.0
        return out, st["target"][:B, :T], st["mask"][:B, :T], ls_sorted

    def _ship(self, slot, batch):
        out, tg, mk, ls = batch
        if self.device is None:
            return {m: v.clone() for m, v in out.items()}, tg.clone(), mk.clone(), ls
        if not self.cuda:
            return {m: v.to(self.device) for m, v in out.items()}, tg.to(self.device), mk.to(self.device), ls
        with torch.cuda.stream(self._copy_stream):
            # (a strided slice of a pinned buffer is still page-locked memory: the copies below are asynchronous)
            d = {m: v.to(self.device, non_blocking=True) for m, v in out.items()}
            t, k = tg.to(self.device, non_blocking=True), mk.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        self._stage[slot]["event"] = ev
        return d, t, k, ls, ev

    def __iter__(self):
        ranges = [(lo, min(lo + self.batch_size, self.ds.n)) for lo in range(0, self.ds.n, self.batch_size)]
        pending = None
        for j, (lo, hi) in enumerate(ranges):
            shipped = self._ship(j % self.slots, self._assemble(j % self.slots, lo, hi))       # batch j on its way while batch j - 1 is consumed
            if pending is not None:
                yield self._deliver(pending)
            pending = shipped
        if pending is not None:
            yield self._deliver(pending)

    def _deliver(self, shipped):
        if len(shipped) == 5:
            d, t, k, ls, ev = shipped
            torch.cuda.current_stream(self.device).wait_event(ev)      # the consumer's stream waits for the copy; the host does not
            for v in list(d.values()) + [t, k]:
                v.record_stream(torch.cuda.current_stream(self.device))
            return d, t, k, ls
        return shipped
