"""Target-distribution interface (reference: src/gmmvi/experiments/target_distributions/lnpdf.py:6-127).

User targets subclass ``LNPDF`` exactly as with the reference.  Arrays crossing this boundary are ``DeviceArray``s
(``.numpy()`` like a tf.Tensor) on the way in; NumPy arrays or anything with ``.numpy()`` on the way out.  There is
no automatic differentiation in this build: a first-order estimator (Stein) needs ``log_density_and_grad``; the
built-in targets implement it with analytic-gradient kernels.
"""


class LNPDF:
    def __init__(self, use_log_density_and_grad: bool = False, safe_for_tf_graph: bool = True):
        self._use_log_density_and_grad = use_log_density_and_grad
        self._safe_for_tf_graph = safe_for_tf_graph

    def log_density(self, x):
        raise NotImplementedError

    def log_density_and_grad(self, x):
        raise NotImplementedError(
            "this target provides no gradient: implement log_density_and_grad (the MI355X build has no autodiff; "
            "the reference's GradientTape fallback, sample_selector.py:74-77, does not exist here)")

    def get_num_dimensions(self) -> int:
        raise NotImplementedError

    def expensive_metrics(self, model, samples) -> dict:
        return dict()

    def can_sample(self) -> bool:
        return False

    @property
    def use_log_density_and_grad(self) -> bool:
        return self._use_log_density_and_grad

    @property
    def safe_for_tf_graph(self) -> bool:
        return self._safe_for_tf_graph

    def sample(self, n: int):
        raise NotImplementedError
