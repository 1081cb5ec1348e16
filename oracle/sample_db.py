"""Oracle restatement of optimization/sample_db.py:30-228 and of the component-based sample
selector gmmvi_modules/sample_selector.py:69-78,140-219 (and :258-339 for the mixture-based one).
TEST INFRASTRUCTURE.
"""
import numpy as np
from scipy.special import logsumexp

from . import philox


def random_subset(rng, total, size):
    """A uniformly random ``size``-subset of range(total), ascending -- the SET that shuffle(range(total))[:size] is
    (sample_db.py:137-152); its order is irrelevant to the consumer, an arg-max over the candidates.  Oversampled independent
    draws, sorted, duplicates dropped, surplus values removed at random: the distinct values of independent uniform draws form a
    uniformly random subset of their number, and removing uniformly chosen members keeps it one.  Vectorised (NumPy's own
    choice(replace=False) walks a hash set element by element: 1.9 ms against 0.85 ms for 1e5 of 3e6 here) and
    ascending, so the gather that follows reads the database front to back."""
    total, size = int(total), min(int(size), int(total))
    if 4 * size > total or total >= 2 ** 31:                 # dense draws (small databases): NumPy's own algorithm
        return np.sort(rng.choice(total, size=size, replace=False, shuffle=False)).astype(np.int32)
    have = None
    while have is None or have.shape[0] < size:
        missing = size - (0 if have is None else have.shape[0])
        draw = rng.integers(0, total, size=missing + max(64, missing // 16), dtype=np.int32)
        have = np.sort(draw if have is None else np.concatenate([have, draw]))
        keep = np.empty(have.shape[0], bool)
        keep[0] = True
        np.not_equal(have[1:], have[:-1], out=keep[1:])
        have = have[keep]
    if have.shape[0] > size:
        have = np.delete(have, rng.choice(have.shape[0], size=have.shape[0] - size, replace=False, shuffle=False))
    return have


class SampleDB:
    def __init__(self, dim, diagonal_covariances, keep_samples, max_samples=None, dtype=np.float64):
        self.diagonal_covariances = bool(diagonal_covariances)                         # :32
        self._dim = dim
        self.dtype = dtype
        self.keep_samples = keep_samples
        self.max_samples = max_samples
        self.samples = np.zeros((0, dim), dtype)
        self.means = np.zeros((0, dim), dtype)
        cshape = (0, dim) if self.diagonal_covariances else (0, dim, dim)              # :36-41
        self.chols = np.zeros(cshape, dtype)
        self.inv_chols = np.zeros(cshape, dtype)
        self.target_lnpdfs = np.zeros(0, dtype)
        self.target_grads = np.zeros((0, dim), dtype)
        self.mapping = np.zeros(0, np.int32)
        self.num_samples_written = 0

    def remove_every_nth_sample(self, n):
        """sample_db.py:63-79 (tf.unique keeps first-occurrence order)."""
        self.samples = self.samples[::n]
        self.target_lnpdfs = self.target_lnpdfs[::n]
        self.target_grads = self.target_grads[::n]
        self.mapping = self.mapping[::n]
        uniq, first = np.unique(self.mapping, return_index=True)
        used = uniq[np.argsort(first)]                       # order of first appearance, like tf.unique
        remap = np.full(int(self.mapping.max()) + 1 if self.mapping.size else 0, -1, np.int64)
        remap[used] = np.arange(len(used))
        self.mapping = remap[self.mapping].astype(np.int32)
        self.means = self.means[used]
        self.chols = self.chols[used]
        self.inv_chols = self.inv_chols[used]

    def add_samples(self, samples, means, chols, target_lnpdfs, target_grads, mapping):
        """sample_db.py:81-135."""
        dt = self.dtype
        samples = np.asarray(samples, dt)
        if self.max_samples is not None and samples.shape[0] + self.samples.shape[0] > self.max_samples:
            self.remove_every_nth_sample(2)
        self.num_samples_written += samples.shape[0]
        if self.diagonal_covariances:
            inv = 1.0 / np.asarray(chols, dt)                                          # :119 / :130
        else:
            inv = np.linalg.inv(np.asarray(chols, dt))                                 # :121 / :132
        if self.keep_samples:
            self.mapping = np.concatenate([self.mapping, np.asarray(mapping, np.int32) + self.chols.shape[0]])
            self.means = np.concatenate([self.means, np.asarray(means, dt)])
            self.chols = np.concatenate([self.chols, np.asarray(chols, dt)])
            self.inv_chols = np.concatenate([self.inv_chols, inv])
            self.samples = np.concatenate([self.samples, samples])
            self.target_lnpdfs = np.concatenate([self.target_lnpdfs, np.asarray(target_lnpdfs, dt)])
            self.target_grads = np.concatenate([self.target_grads, np.asarray(target_grads, dt)])
        else:
            self.mapping = np.asarray(mapping, np.int32).copy()
            self.means = np.asarray(means, dt).copy()
            self.chols = np.asarray(chols, dt).copy()
            self.inv_chols = inv
            self.samples = samples.copy()
            self.target_lnpdfs = np.asarray(target_lnpdfs, dt).copy()
            self.target_grads = np.asarray(target_grads, dt).copy()

    def get_random_sample(self, n, rng):
        """sample_db.py:137-152 (tf.random.shuffle + slice -> random_subset below: the same set law, NumPy generator)."""
        idx = random_subset(rng, self.samples.shape[0], n)
        return self.samples[idx], self.target_lnpdfs[idx]

    def gaussian_log_pdf(self, mean, chol, inv_chol, x):
        """sample_db.py:154-162: uses inv_chol @ (mean - x)^T (diagonal branch :155-158: elementwise)."""
        if self.diagonal_covariances:
            const = -0.5 * self._dim * np.log(2 * np.pi) - np.sum(np.log(chol))
            return const - 0.5 * np.sum(np.square(inv_chol[:, None] * (mean[None, :] - x).T), axis=0)
        const = -0.5 * self._dim * np.log(2 * np.pi) - np.sum(np.log(np.diag(chol)))
        return const - 0.5 * np.sum(np.square(inv_chol @ (mean - x).T), axis=0)

    def evaluate_background(self, weights, means, chols, inv_chols, samples):
        """sample_db.py:164-192: sequential two-way LSE over the background components."""
        lw = np.log(weights)
        out = self.gaussian_log_pdf(means[0], chols[0], inv_chols[0], samples) + lw[0]
        for i in range(1, len(weights)):
            out = logsumexp(np.stack([out, self.gaussian_log_pdf(means[i], chols[i], inv_chols[i], samples) + lw[i]]),
                            axis=0)
        return out

    def get_newest_samples(self, n):
        """sample_db.py:194-228."""
        d = self._dim
        if self.samples.shape[0] == 0 or n == 0:
            return (np.zeros(0, self.dtype), np.zeros((0, d), self.dtype), np.zeros(0, np.int32),
                    np.zeros(0, self.dtype), np.zeros((0, d), self.dtype))
        start = max(0, self.samples.shape[0] - n)
        xs = self.samples[start:]
        mapping = self.mapping[start:]
        uniq, first, counts = np.unique(mapping, return_index=True, return_counts=True)
        order = np.argsort(first)                                        # tf.unique_with_counts order
        active, counts = uniq[order], counts[order].astype(self.dtype)
        w = counts / counts.sum()
        bg = self.evaluate_background(w, self.means[active], self.chols[active], self.inv_chols[active], xs)
        return bg, xs, mapping, self.target_lnpdfs[start:], self.target_grads[start:]


class VipsSampleSelector:
    """gmmvi_modules/sample_selector.py:103-219 ("M", component-based)."""

    def __init__(self, target, model, sample_db, desired_samples_per_component, ratio_reused_samples_to_desired,
                 seed=0):
        self.target_distribution = target
        self.model = model
        self.sample_db = sample_db
        self.desired_samples_per_component = desired_samples_per_component
        self.reused_samples_per_component = int(np.floor(ratio_reused_samples_to_desired * desired_samples_per_component))
        self.seed = seed
        self.eps_override = None     # optional callable(n, d) -> eps for host-provided normals

    def get_effective_samples(self, model_densities, oldsamples_pdf):
        """:140-158."""
        lw = model_densities - oldsamples_pdf[None, :]
        lw = lw - logsumexp(lw, axis=1, keepdims=True)
        w = np.exp(lw)
        return 1.0 / np.sum(w * w, axis=1)

    def _draw_eps(self, n):
        d = self.model.num_dimensions
        if self.eps_override is not None:
            return self.eps_override(n, d)
        return philox.normals(self.seed, self.sample_db.num_samples_written, n, d, philox.STREAM_COMPONENT_NORMALS)

    def sample_where_needed(self, samples, oldsamples_pdf):
        """:160-202."""
        if samples.shape[0] == 0:
            n_eff = np.zeros(self.model.num_components, np.int64)
        else:
            n_eff = np.floor(self.get_effective_samples(self.model.component_log_densities(samples),
                                                        oldsamples_pdf)).astype(np.int64)
        n_add = np.maximum(1, self.desired_samples_per_component - n_eff)
        eps = self._draw_eps(int(n_add.sum()))
        new_samples, mapping = self.model.sample_from_components_no_shuffle(n_add, eps)
        lp, grad = self.target_distribution.log_density_and_grad(new_samples)
        return new_samples, lp, grad, mapping

    def select_samples(self):
        """:204-219."""
        n_reuse = self.reused_samples_per_component * self.model.num_components
        old_pdf, samples, _, _, _ = self.sample_db.get_newest_samples(n_reuse)
        n_reused = samples.shape[0]
        new_samples, new_lp, new_grad, mapping = self.sample_where_needed(samples, old_pdf)
        self.sample_db.add_samples(new_samples, self.model.means, self.model.chol_cov, new_lp, new_grad, mapping)
        bg, samples, mapping, lp, grad = self.sample_db.get_newest_samples(n_reused + new_samples.shape[0])
        return samples, mapping, bg, lp, grad


class LinSampleSelector(VipsSampleSelector):
    """gmmvi_modules/sample_selector.py:221-339 ("P", mixture-based)."""

    def select_samples(self):
        n_reuse = self.reused_samples_per_component * self.model.num_components
        old_pdf, old_samples, _, _, _ = self.sample_db.get_newest_samples(n_reuse)
        n_reused = old_samples.shape[0]
        if n_reused == 0:
            n_eff = 0
        else:
            lw = self.model.log_density(old_samples) - old_pdf            # :273-277 on a [N] vector, axis=1 -> whole vector
            lw = lw - logsumexp(lw)
            n_eff = int(np.floor(1.0 / np.sum(np.exp(lw) ** 2)))
        n_add = max(1, self.desired_samples_per_component - n_eff)
        first = self.sample_db.num_samples_written
        new_samples, mapping = self.model.sample(n_add, self.seed, first)
        lp, grad = self.target_distribution.log_density_and_grad(new_samples)
        self.sample_db.add_samples(new_samples, self.model.means, self.model.chol_cov, lp, grad, mapping)
        bg, samples, mapping, lp, grad = self.sample_db.get_newest_samples(n_reused + new_samples.shape[0])
        return samples, mapping, bg, lp, grad
