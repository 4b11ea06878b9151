#!/usr/bin/env python3
"""Generate tests/golden/batching.npz from the REFERENCE's host batching code (build container only).

Run:  MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_batching.py

transformer/SFT/train.py::generateTrainBatch (:74-106, with generateInputChunkHelper :59-69 and chunks :52-55) and
eval_ccc (:42-50) are imported from /root/reference (read-only, never copied) and run on small synthetic padded lists; the
batches they yield — sorted data, targets, prefix masks, sorted lengths — and per-sequence CCCs are stored as data.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe as R  # noqa: E402

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF_DIR = "/root/reference/transformer/SFT"


def load_train():
    sys.path.insert(0, REF_DIR)                       # train.py imports its sibling modules (datasets, models)
    spec = importlib.util.spec_from_file_location("ref_sft_train", os.path.join(REF_DIR, "train.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    tr = load_train()
    n_seq, t_max, dims = 8, 9, {"acoustic": (3, 4), "linguistic": (2, 5)}
    lengths = [5, 9, 3, 9, 1, 7, 7, 2]                # ties (9, 9) and (7, 7): the reference's sort is stable
    data = {m: R.gen_normal("batching:" + m, (n_seq, t_max, w, d), 3).numpy() for m, (w, d) in dims.items()}
    target = R.gen_uniform("batching:target", (n_seq, t_max), 3).numpy()
    for i, L in enumerate(lengths):                   # padded tails are zeros, as constructInput/padInput leave them
        for m in data:
            data[m][i, L:] = 0
        target[i, L:] = 0
    out = {"lengths": np.array(lengths), "t_max": np.array(t_max)}
    for m in data:
        out["in:" + m] = data[m]
    out["in:target"] = target
    lists = {m: v.tolist() for m, v in data.items()}
    for bs in (3, 25, 1):
        for k, (d, tgt, mask, ls) in enumerate(tr.generateTrainBatch(lists, target.tolist(), list(lengths), None, batch_size=bs)):
            pre = "bs%d:%d:" % (bs, k)
            for m in d:
                out[pre + m] = d[m].numpy()
            out[pre + "target"] = tgt.numpy()
            out[pre + "mask"] = mask.numpy()
            out[pre + "lengths"] = np.array(ls)
        out["bs%d:n" % bs] = np.array(k + 1)
    # per-sequence CCC as evaluate() computes it (:196-254): whole (1, T) rows of output vs target of one sequence
    pred = R.gen_uniform("batching:pred", (n_seq, t_max), 3).numpy()
    out["ccc:pred"] = pred
    out["ccc:full"] = np.array([tr.eval_ccc(pred[i, :lengths[i]], target[i, :lengths[i]]) if lengths[i] > 1 else np.nan
                                for i in range(n_seq)], dtype=np.float64)
    path = os.path.join(HERE, "batching.npz")
    np.savez_compressed(path, **out)
    print("batching.npz %.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
