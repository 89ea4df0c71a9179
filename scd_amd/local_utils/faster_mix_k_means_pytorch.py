"""Drop-in for /root/reference/gcd/methods/clustering/faster_mix_k_means_pytorch.py (K_Means :47-275,
pairwise_distance :9-44) on libscd_hip.so."""
from ..kmeans import KMeansEngine, check_random_state  # noqa: F401
from .sskm_constrained import pairwise_distance  # noqa: F401


class K_Means(KMeansEngine):
    def __init__(self, k=3, tolerance=1e-4, max_iterations=100, init='k-means++', n_init=10, random_state=None, n_jobs=None,
                 pairwise_batch_size=None, mode=None, **kw):
        super().__init__(k=k, tolerance=tolerance, max_iterations=max_iterations, init=init, n_init=n_init,
                         random_state=random_state, n_jobs=n_jobs, pairwise_batch_size=pairwise_batch_size, **kw)
        self.mode = mode
