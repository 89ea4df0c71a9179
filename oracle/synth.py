"""Seeded synthetic inputs shared by the oracle, the parity tests and bench.py.

TEST/BENCH INFRASTRUCTURE - not a product path.  Generators follow SURVEY.md
section 8(d): clustered unit-norm features (so k-means margins are honest), a
vocabulary matrix W[512, V] whose first K columns sit near the K "true" class
directions, and a labelled/unlabelled split ordered labelled-first as the
reference's MergedDataset does (gcd/data/data_utils.py:27-32).
"""
import numpy as np


def _normalize(x, axis=-1):
    n = np.linalg.norm(x, axis=axis, keepdims=True)
    return x / np.maximum(n, 1e-30)


def clustered_features(n, d, k, seed=13, center_seed=11, noise=0.6, dtype=np.float32):
    """X = normalize(centers[y] + noise/sqrt(d) * randn). Returns (X, y, centers)."""
    rc = np.random.RandomState(center_seed)
    centers = _normalize(rc.randn(k, d))
    r = np.random.RandomState(seed)
    y = r.randint(0, k, size=n)
    x = centers[y] + (noise / np.sqrt(d)) * r.randn(n, d)
    x = _normalize(x)
    return x.astype(dtype), y.astype(np.int64), centers.astype(dtype)


def labelled_split(y, k, prop=0.5, seed=5):
    """Classes < k/2 are 'old'; `prop` of their samples are labelled.

    Returns (perm, mask_lab) with perm ordering labelled rows first
    (reference: gcd/data/get_datasets.py:144-145, data_utils.py:27-32).
    """
    r = np.random.RandomState(seed)
    old = y < (k // 2)
    lab = old & (r.rand(len(y)) < prop)
    perm = np.concatenate([np.nonzero(lab)[0], np.nonzero(~lab)[0]])
    mask_lab = np.zeros(len(y), dtype=bool)
    mask_lab[: int(lab.sum())] = True
    return perm, mask_lab


def vocabulary(v, d_clip, class_dirs, seed=7, jitter=0.15, dtype=np.float16):
    """W[d_clip, V]: column c < K is near class_dirs[c]; the rest are random unit
    distractors.  Column-major-by-name like zeroshot_classifier's output
    (local_utils/clip_lang_util.py:107)."""
    r = np.random.RandomState(seed)
    k = class_dirs.shape[0]
    w = _normalize(r.randn(v, d_clip))
    w[:k] = _normalize(class_dirs + (jitter / np.sqrt(d_clip)) * r.randn(k, d_clip))
    return np.ascontiguousarray(w.T).astype(dtype)


def nouns_list(v):
    return ["name_%05d" % i for i in range(v)]


def blob_case(n, d, k, seed):
    """Labelled-first clustered case used by the k-means goldens (oracle/gen_golden.py)."""
    x, y, _ = clustered_features(n, d, k, seed=seed, center_seed=seed + 100, noise=0.8)
    perm, mask_lab = labelled_split(y, k, prop=0.5, seed=seed + 7)
    return x[perm], y[perm], mask_lab
