"""CPU oracle for the gmmvi per-iteration hot path.  TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy restatement (fp64 by default, fp32 on request) of the
reference's algorithm for the path named in BASELINE.json's north_star.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and there only as the checker or the reported CPU baseline -- never from
the product package ``gmmvi_amd``.

PARITY UNPINNED: the reference (TensorFlow 2 / TensorFlow-Probability) cannot be
imported in the build container (``ModuleNotFoundError: tensorflow``; nothing was
denied) and ships no tests, golden vectors or fixtures for this path (SURVEY.md §4,
§8c).  The restatement is therefore pinned only by (i) SciPy cross-checks
(``multivariate_normal``, ``multivariate_t``, ``solve_triangular``, ``logsumexp``),
(ii) central finite differences of every gradient, (iii) closed-form identities
(Gaussian KL, Stein identities on a Gaussian target), and (iv) end-to-end known
answers (single-Gaussian target => exact mean/covariance, ELBO -> log Z).  See
``tests/test_oracle_*.py``.

Every function cites the reference file:line (relative to /root/reference/src/gmmvi)
whose arithmetic it restates.
"""

from . import philox, gmm, targets, sample_db, stein, more, updaters, weights, stepsizes, adaptation, train, mmd  # noqa: F401
