"""Drop-in for the parts of /root/reference/gcd/project_utils/cluster_utils.py the naming path uses:
linear_assignment (:234-275, vendored Munkres) -> scd_munkres (C++), cluster_acc (:39-62)."""
import numpy as np
from ... import ops


def linear_assignment(X):
    return ops.munkres(X)


def cluster_acc(y_true, y_pred, return_ind=False):
    y_true = np.asarray(y_true).astype(int)
    y_pred = np.asarray(y_pred).astype(int)
    assert y_pred.size == y_true.size
    D = max(y_pred.max(), y_true.max()) + 1
    w = np.zeros((D, D), dtype=int)
    np.add.at(w, (y_pred, y_true), 1)
    ind = linear_assignment(w.max() - w)
    acc = sum(w[i, j] for i, j in ind) * 1.0 / y_pred.size
    return (acc, ind, w) if return_ind else acc
