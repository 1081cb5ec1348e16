"""Oracle restatement of experiments/evaluation/mmd.py:4-78 (fp64 NumPy).  TEST INFRASTRUCTURE.

tfp.stats.percentile(x, 50, axis=0) (tensorflow-probability 0.20.1, default interpolation "nearest") returns the
sorted value at index round((n - 1) * 0.5) (round-half-to-even, tf.round); restated here on the sorted column.
"""
import numpy as np


def compute_sigma(groundtruth, max_points_for_median=1000):
    """:25-35."""
    g = np.asarray(groundtruth, np.float32)                        # :20 casts to float32
    m = int(min(max_points_for_median, len(g)))
    rows = []
    for i in range(m):
        rows.append(np.square(g[i][None, :] - g[i:m]))             # pairs (i, j >= i), :30-33
    dist = np.sort(np.concatenate(rows, axis=0), axis=0)
    idx = int(np.round((dist.shape[0] - 1) * 0.5))                 # "nearest", half-to-even
    return np.diag(dist[idx].astype(np.float64))


def pair_sum(a, b, kernel):
    """:41-48 / :50-58: sum_i sum_j exp(-(a_i - b_j)^T kernel (a_i - b_j))."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    total = 0.0
    for i in range(a.shape[0]):
        diff = a[i] - b
        total += np.sum(np.exp(-np.sum(diff @ kernel * diff, axis=1)))
    return total


def compute_mmd(groundtruth, model_sample, alpha, sigma=None):
    """:64-78."""
    sigma = compute_sigma(groundtruth) if sigma is None else sigma
    kernel = np.linalg.inv(alpha * sigma)
    n1, n2 = len(groundtruth), len(model_sample)
    return (pair_sum(groundtruth, groundtruth, kernel) / n1 ** 2 + pair_sum(model_sample, model_sample, kernel) / n2 ** 2
            - 2 * pair_sum(groundtruth, model_sample, kernel) / (n1 * n2))
