"""GMMVI orchestrator (reference: src/gmmvi/optimization/gmmvi.py:16-174).

Same constructor, ``build_from_config`` and ``train_iter`` as the reference; the module objects are the device-backed
mirrors in ``gmmvi_modules``.  ``train_iter`` issues the whole iteration asynchronously on the context's HIP stream
(no host synchronisation on the SAMTRON path with a zero reuse ratio); results become visible to the host through
``.numpy()`` on any returned array.
"""
from ..models.gmm_wrapper import GmmWrapper
from .gmmvi_modules.component_stepsize_adaptation import ComponentStepsizeAdaptation
from .gmmvi_modules.component_adaptation import ComponentAdaptation
from .gmmvi_modules.ng_based_component_updater import NgBasedComponentUpdater
from .gmmvi_modules.ng_estimator import NgEstimator
from .gmmvi_modules.sample_selector import SampleSelector
from .gmmvi_modules.weight_stepsize_adaptation import WeightStepsizeAdaptation
from .gmmvi_modules.weight_updater import WeightUpdater
from .sample_db import SampleDB


class _Counter:
    """num_updates with the tf.Variable-like surface the reference callers touch (gmmvi.py:87, :174)."""
    def __init__(self):
        self.value = 0
    def assign_add(self, n):
        self.value += int(n)
    def numpy(self):
        return self.value
    def __int__(self):
        return self.value
    def __index__(self):
        return self.value
    def __eq__(self, other):
        return self.value == int(other)
    def __gt__(self, other):
        return self.value > int(other)
    def __mod__(self, other):
        return self.value % int(other)
    def __repr__(self):
        return str(self.value)


class GMMVI:
    def __init__(self, model: GmmWrapper, sample_db: SampleDB, temperature, sample_selector: SampleSelector,
                 num_component_adapter: ComponentAdaptation, component_stepsize_adapter: ComponentStepsizeAdaptation,
                 ng_estimator: NgEstimator, ng_based_updater: NgBasedComponentUpdater,
                 weight_stepsize_adapter: WeightStepsizeAdaptation, weight_updater: WeightUpdater):
        self.temperature = temperature
        self.model = model
        self.num_dimensions = self.model.num_dimensions
        self.sample_db = sample_db
        self.sample_selector = sample_selector
        self.num_component_adapter = num_component_adapter
        self.component_stepsize_adapter = component_stepsize_adapter
        self.ng_estimator = ng_estimator
        self.ng_based_updater = ng_based_updater
        self.weight_stepsize_adapter = weight_stepsize_adapter
        self.weight_updater = weight_updater
        self.num_updates = _Counter()
        from .fused import SamtronFastPath
        self._fast_path = SamtronFastPath(self)

    # the plug-in modules of an iteration: (constructor argument of GMMVI, factory(config, model, sample_db=, target_distribution=));
    # gmmvi.py:105-144 builds the same objects one statement at a time
    _MODULES = (
        ("ng_estimator", lambda c, m, **kw: NgEstimator.build_from_config(c, c['temperature'], m)),
        ("ng_based_updater", lambda c, m, **kw: NgBasedComponentUpdater.build_from_config(c, m)),
        ("num_component_adapter", lambda c, m, **kw: ComponentAdaptation.build_from_config(
            c, m, kw["sample_db"], target_distribution=kw["target_distribution"],
            prior_mean=c["model_initialization"]["prior_mean"], initial_cov=c["model_initialization"]["initial_cov"])),
        ("component_stepsize_adapter", lambda c, m, **kw: ComponentStepsizeAdaptation.build_from_config(c, m)),
        ("sample_selector", lambda c, m, **kw: SampleSelector.build_from_config(c, m, kw["sample_db"],
                                                                               kw["target_distribution"])),
        ("weight_updater", lambda c, m, **kw: WeightUpdater.build_from_config(c, m)),
        ("weight_stepsize_adapter", lambda c, m, **kw: WeightStepsizeAdaptation.build_from_config(c, m)),
    )

    @staticmethod
    def build_from_config(config: dict, target_distribution, model: GmmWrapper):
        """gmmvi.py:105-144: one module per entry of ``_MODULES``, all on the same model and sample database."""
        sample_db = SampleDB.build_from_config(config, model.num_dimensions)
        parts = {name: factory(config, model, sample_db=sample_db, target_distribution=target_distribution)
                 for name, factory in GMMVI._MODULES}
        return GMMVI(model, sample_db, config['temperature'], **parts)

    def train_iter(self):
        """gmmvi.py:146-161.  Built-in SAMTRON-style module sets take the single-call fast path (optimization/fused.py:
        the same kernels in the same order, one host call); everything else runs module by module."""
        if self._fast_path.eligible():
            self._fast_path.step()
        else:
            samples, mapping, sample_dist_densities, target_lnpdfs, target_lnpdf_grads = \
                self.sample_selector.select_samples()
            self._run_updates(samples, mapping, sample_dist_densities, target_lnpdfs, target_lnpdf_grads)
        self.num_component_adapter.adapt_number_of_components(self.num_updates)

    def _run_updates(self, samples, mapping, sample_dist_densities, target_lnpdfs, target_lnpdf_grads):
        """gmmvi.py:163-174 -- the ordering contract of the hot path: stepsizes, natural-gradient estimate, component
        update, then weight stepsize and weight update."""
        new_component_stepsizes = self.component_stepsize_adapter.update_stepsize(self.model.stepsizes)
        self.model.update_stepsizes(new_component_stepsizes)
        # only_use_own_samples needs max(mapping) (ng_estimator.py:244): read it off the DB's host mirror of the mapping
        # (a mixture-based selector may leave the newest DB component without a draw, so "number of DB components - 1"
        # is not it)
        self.model._mapping_max_hint = None
        if getattr(self.ng_estimator, "_only_use_own_samples", False) and hasattr(self.sample_db, "newest_mapping_host"):
            host_mapping = self.sample_db.newest_mapping_host(samples.shape[0])
            if host_mapping.size:
                self.model._mapping_max_hint = int(host_mapping.max())
        expected_hessian_neg, expected_grad_neg = self.ng_estimator.get_expected_hessian_and_grad(
            samples, mapping, sample_dist_densities, target_lnpdfs, target_lnpdf_grads)
        self.ng_based_updater.apply_NG_update(expected_hessian_neg, expected_grad_neg, self.model.stepsizes)
        weight_stepsize = self.weight_stepsize_adapter.update_stepsize()
        self.weight_updater.update_weights(samples, sample_dist_densities, target_lnpdfs, weight_stepsize)
        self.num_updates.assign_add(1)
