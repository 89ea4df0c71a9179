"""Drop-in for split_cluster_acc_v2 of /root/reference/gcd/project_utils/cluster_and_log_utils.py:29-74."""
import numpy as np
from .cluster_utils import linear_assignment


def split_cluster_acc_v2(y_true, y_pred, mask, return_ind_map=False):
    y_true = np.asarray(y_true).astype(int)
    y_pred = np.asarray(y_pred).astype(int)
    mask = np.asarray(mask).astype(bool)
    old_classes_gt = set(y_true[mask].tolist())
    new_classes_gt = set(y_true[~mask].tolist())
    assert y_pred.size == y_true.size
    D = max(y_pred.max(), y_true.max()) + 1
    w = np.zeros((D, D), dtype=int)
    np.add.at(w, (y_pred, y_true), 1)            # contingency (:47-51)
    ind = linear_assignment(w.max() - w)
    ind_map = {int(j): int(i) for i, j in ind}
    total_acc = sum(w[i, j] for i, j in ind) * 1.0 / y_pred.size
    col = w.sum(axis=0)
    old_acc = sum(w[ind_map[i], i] for i in old_classes_gt) / max(1, sum(col[i] for i in old_classes_gt))
    new_acc = sum(w[ind_map[i], i] for i in new_classes_gt) / max(1, sum(col[i] for i in new_classes_gt))
    if return_ind_map:
        return total_acc, old_acc, new_acc, ind_map
    return total_acc, old_acc, new_acc
