"""Evaluation metric of the reference training loop."""
import numpy as np


def eval_ccc(y_true, y_pred):
    """Concordance correlation coefficient with population (ddof=0) moments, as
    transformer/SFT/train.py:42-50 computes it (np.var, np.cov(bias=True))."""
    yt = np.asarray(y_true, dtype=np.float64).reshape(-1)
    yp = np.asarray(y_pred, dtype=np.float64).reshape(-1)
    dt, dp = yt - yt.mean(), yp - yp.mean()
    return float(2.0 * np.mean(dt * dp) / (np.mean(dt * dt) + np.mean(dp * dp) + (yp.mean() - yt.mean()) ** 2))
