"""Evaluation metric of the reference training loop."""
import numpy as np


def eval_ccc(y_true, y_pred):
    """Concordance correlation coefficient with population (ddof=0) moments, as
    transformer/SFT/train.py:42-50 computes it (np.var, np.cov(bias=True))."""
    yt = np.asarray(y_true, dtype=np.float64).reshape(-1)
    yp = np.asarray(y_pred, dtype=np.float64).reshape(-1)
    dt, dp = yt - yt.mean(), yp - yp.mean()
    return float(2.0 * np.mean(dt * dp) / (np.mean(dt * dt) + np.mean(dp * dp) + (yp.mean() - yt.mean()) ** 2))


def batched_ccc(pred, target, lengths):
    """Per-sequence CCC of a whole batch on the device: pred, target (B, T[, 1]) HIP tensors, lengths B ints -> (B,) float64
    tensor on the same device; sequence b uses its first lengths[b] windows (population moments, as eval_ccc above).
    One kernel (csrc/misc_kernels.h ccc_kernel), no host round trip per sequence as in the reference's evaluate()
    (transformer/SFT/train.py:236-238)."""
    import torch
    from . import _lib
    lib = _lib.load()
    _lib.require_hip(pred, target)
    p_ = pred.detach().reshape(pred.shape[0], -1).contiguous().float()
    t_ = target.detach().reshape(target.shape[0], -1).contiguous().float()
    if p_.shape != t_.shape or len(lengths) != p_.shape[0]:
        raise ValueError("batched_ccc: pred %s, target %s, %d lengths" % (tuple(pred.shape), tuple(target.shape), len(lengths)))
    B, T = p_.shape
    if any(int(L) < 0 or int(L) > T for L in lengths):
        raise ValueError("batched_ccc: a length is outside [0, T]")
    ln = torch.as_tensor([int(L) for L in lengths], dtype=torch.int32).to(p_.device)
    out = torch.empty(B, dtype=torch.float64, device=p_.device)
    _lib.check(lib.mmt_ccc_forward(_lib.ptr(p_), _lib.ptr(t_), _lib.ptr(ln), _lib.ptr(out), B, T, _lib.stream_ptr()))
    return out
