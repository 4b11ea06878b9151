"""Oracle: metric and loss of the reference training loop (TEST INFRASTRUCTURE)."""
import numpy as np
import torch


def eval_ccc(y_true, y_pred):
    """Concordance correlation coefficient — transformer/SFT/train.py:42-50.

    Population moments throughout (np.var default ddof=0; np.cov(..., bias=True)).
    """
    t = np.asarray(y_true, dtype=np.float64).ravel()
    q = np.asarray(y_pred, dtype=np.float64).ravel()
    mt, mq = t.mean(), q.mean()
    vt = ((t - mt) ** 2).mean()
    vq = ((q - mq) ** 2).mean()
    cov = ((t - mt) * (q - mq)).mean()
    return 2.0 * cov / (vt + vq + (mq - mt) ** 2)


def masked_mse_sum_loss(output, target, lengths):
    """Per-batch loss of transformer/SFT/train.py:133-137: ``MSELoss(reduction='sum')``
    (criterion built at :538) divided by the total number of valid windows in the batch."""
    return ((output - target) ** 2).sum() / float(sum(lengths))
