"""Maximum mean discrepancy (reference: src/gmmvi/experiments/evaluation/mmd.py:4-78).

The bandwidths (median trick, once per groundtruth set) are computed on the host; the three U-statistics are pair
sweeps on the device (csrc/mmd.hip: ``gmmvi_mmd_pair_sum``), replacing the reference's Python loop over the rows with
one [N, D] temporary per step.
"""
import numpy as np

from ... import hip_ops
from ...device import get_context


class MMD:
    """Parameters (mmd.py:19): groundtruth [N, D]; alpha scales the diagonal bandwidth matrix."""

    def __init__(self, groundtruth, alpha, ctx=None):
        self.ctx = ctx if ctx is not None else get_context()
        gt = np.asarray(groundtruth.numpy() if hasattr(groundtruth, "numpy") else groundtruth, np.float32)
        if gt.ndim != 2 or gt.shape[0] == 0:
            raise ValueError("groundtruth must be a non-empty [N, D] array")
        self.groundtruth = self.ctx.asarray(gt)
        self._gt_host = gt
        self.num_groundtruth = int(gt.shape[0])
        self.sigma = self.compute_sigma()                                                   # :22
        self.set_alpha(alpha)

    def compute_sigma(self, max_points_for_median=1000):
        """:25-35: per-dimension median ("nearest" percentile, tfp's default) of the squared coordinate differences
        of all pairs i <= j among the first 1000 points, as a diagonal matrix."""
        m = int(min(max_points_for_median, self.num_groundtruth))
        g = self._gt_host[:m]
        iu, ju = np.triu_indices(m)                                                         # i <= j, row-major as :30-33
        d = g.shape[1]
        med = np.empty(d, np.float32)
        for c in range(d):                                                                  # one column at a time: O(m^2) floats
            dist = np.square(g[iu, c] - g[ju, c])
            med[c] = np.percentile(dist, 50, method="nearest")
        return np.diag(med)

    def _inv_bandwidth(self, alpha):
        """kernel = inv(alpha * sigma) (:41, :52): diagonal."""
        return self.ctx.asarray((1.0 / (np.float32(alpha) * np.diag(self.sigma))).astype(np.float32))

    def compute_ustat(self, sample, alpha):
        """:38-48."""
        s = self.ctx.asarray(sample)
        return hip_ops.mmd_pair_sum(self.ctx, s, s, self._inv_bandwidth(alpha))

    def kernel_mix(self, sample, alpha):
        """:50-58."""
        return hip_ops.mmd_pair_sum(self.ctx, self.groundtruth, self.ctx.asarray(sample), self._inv_bandwidth(alpha))

    def set_alpha(self, alpha):
        """:60-62."""
        self._alpha = alpha
        self.ustat1 = self.compute_ustat(self.groundtruth, alpha)

    def compute_MMD(self, model_sample):
        """:64-78."""
        s = self.ctx.asarray(model_sample)
        num_1, num_2 = self.num_groundtruth, int(s.shape[0])
        return (self.ustat1 / (num_1 ** 2) + self.compute_ustat(s, self._alpha) / (num_2 ** 2)
                - 2 * self.kernel_mix(s, self._alpha) / (num_1 * num_2))
