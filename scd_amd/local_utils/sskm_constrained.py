"""Drop-in for /root/reference/local_utils/sskm_constrained.py (K_Means :15-187, pairwise_distance :189-224,
_labels_constrained :226-274) on libscd_hip.so.  Same names, arguments, attributes and error behaviour."""
import numpy as np
import torch

from ..kmeans import ConstrainedEngine, check_random_state  # noqa: F401
from .. import ops


class K_Means(ConstrainedEngine):
    def __init__(self, k=3, tolerance=1e-4, max_iterations=100, size_min=100, size_max=1000, init='k-means++', n_init=10,
                 random_state=None, n_jobs=None, pairwise_batch_size=None, **kw):
        super().__init__(k=k, tolerance=tolerance, max_iterations=max_iterations, size_min=size_min, size_max=size_max,
                         init=init, n_init=n_init, random_state=random_state, n_jobs=n_jobs,
                         pairwise_batch_size=pairwise_batch_size, **kw)


def pairwise_distance(data1, data2, batch_size=None):
    """Squared Euclidean distances [N,M] float32.  Like the reference (:209), the result lives on the CPU when
    `batch_size` is given and on the input device otherwise."""
    data = ops.KMeansData.__new__(ops.KMeansData)          # no E-step operand needed for the exact sweep
    data.x = data1.to(torch.float32).contiguous()
    data.n, data.d = data.x.shape
    out = data.dist(data2.to(data.x.device), sqrt=False)
    return out.cpu() if batch_size is not None else out


def _labels_constrained(X, centers, D_sqrt, size_min, size_max, distances):
    """numpy in / numpy out, overwrites `distances` in place (:226-274)."""
    D = np.asarray(D_sqrt)
    costs = np.around(D * 1000, 0).astype('int32')
    labels, _ = ops.transport_solve(costs, size_min, size_max)
    labels = labels.astype(np.int32)
    distances[:] = D[np.arange(D.shape[0]), labels] ** 2
    return labels, distances.sum()
