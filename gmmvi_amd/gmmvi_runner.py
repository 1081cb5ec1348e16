"""Runner around ``GMMVI.train_iter()``: seeding, wall-clock bookkeeping, metric evaluation and model dumps.

Drop-in for the reference's ``gmmvi.gmmvi_runner.GmmviRunner`` (src/gmmvi/gmmvi_runner.py:23-200): the constructor
signature, ``build_from_config``, the five public methods, the metric keys (:111-116, :134-135, :142), the checkpoint
line (:169-171) and the npz keys (:187-200) are the contract; everything behind them is organised for a device-resident
model: one seeding helper, an evaluator that keeps the 2000 evaluation samples on the GPU, and a dump writer that owns the
output directory.
"""
import os
import random
import time

import numpy as np

from .experiments.setup_experiment import init_experiment
from .optimization.gmmvi import GMMVI

_NUM_EVAL_SAMPLES = 2000          # gmmvi_runner.py:131
_CHECKPOINT_LINE = "Checkpoint {:3d} | FEVALS: {:10d} | avg. sample logpdf: {:05.05f} | ELBO: {:05.05f}"


def _seed_everything(seed):
    """What tf.keras.utils.set_random_seed does upstream (:38), minus TensorFlow: python and numpy generators here, the
    device Philox streams are keyed by the caller with the same seed."""
    random.seed(seed)
    np.random.seed(seed)


def _host(value):
    return value.numpy() if hasattr(value, "numpy") else np.asarray(value)


class _GmmDumper:
    """Writes ``gmm_dump_<n>.npz`` / ``final_gmm_dump.npz`` below ``<dump_gmm_path>/<creation time>/`` (:56-61, :177-200)."""

    KEYS = ("weights", "means", "covs", "timestamps", "fevals")

    def __init__(self, root):
        self.directory = os.path.join(root, str(time.time()))
        os.makedirs(self.directory, exist_ok=True)

    @staticmethod
    def wanted(iteration):
        return iteration < 100 or iteration % 50 == 0

    def write(self, name, gmmvi):
        model = gmmvi.model
        payload = dict(zip(self.KEYS, (np.exp(_host(model.log_weights)), _host(model.means), _host(model.covs), time.time(),
                                      gmmvi.sample_db.num_samples_written.numpy())))
        np.savez(os.path.join(self.directory, name), **payload)


class GmmviRunner:
    def __init__(self, config, log_metrics_interval):
        config.setdefault("seed", config.get("start_seed"))
        _seed_everything(config["seed"])
        self.config = config
        self.log_metrics_interval = log_metrics_interval
        self.wall_times = []

        target_distribution, initial_model = init_experiment(config)
        initial_model.model.seed = int(config["seed"])
        self.gmmvi = GMMVI.build_from_config(config, target_distribution, initial_model)

        self.mmd = self._build_mmd(config.get("mmd_evaluation_config"))
        self._dumper = _GmmDumper(config["dump_gmm_path"]) if "dump_gmm_path" in config else None
        self.dump_gmms = self._dumper is not None
        if self.dump_gmms:
            self.dump_gmm_path = self._dumper.directory

    @staticmethod
    def _build_mmd(mmd_config):
        """:45-54 (no graph to warm up here: the MMD kernel is a plain launch)."""
        if mmd_config is None:
            return None
        from .experiments.evaluation.mmd import MMD
        here = os.path.dirname(os.path.realpath(__file__))
        return MMD(np.load(os.path.join(here, mmd_config["sample_dir"])), mmd_config["alpha"])

    @staticmethod
    def build_from_config(config: dict):
        """:63-81: ``config['gmmvi_runner_config']`` holds the runner's own arguments."""
        return GmmviRunner(config=config, **config["gmmvi_runner_config"])

    # ---- metrics ------------------------------------------------------------------------------------------------
    def get_samples_and_entropy(self, num_samples):
        """:83-100 -> (samples drawn from the model, Monte-Carlo entropy estimate)."""
        samples = self.gmmvi.model.sample(num_samples)[0]
        log_q = _host(self.gmmvi.model.log_density(samples))
        return samples, -float(log_q.mean())

    def get_cheap_metrics(self):
        """:102-117: quantities that cost nothing to read after an iteration."""
        db, model = self.gmmvi.sample_db, self.gmmvi.model
        return {
            "num_samples": db.num_samples_written.numpy(),
            "num_components": model.num_components,
            "max_weight": float(np.max(model.weights)),
            "num_db_samples": db.samples.shape[0],
            "num_db_components": db.means.shape[0],
        }

    def get_expensive_metrics(self):
        """:119-144: ELBO = E_q[log p~] + temperature * H(q) on fresh samples, plus the target's own metrics and the MMD."""
        samples, entropy = self.get_samples_and_entropy(_NUM_EVAL_SAMPLES)
        selector = self.gmmvi.sample_selector
        target_density = float(_host(selector.target_uld(samples)).mean())
        metrics = {
            "-elbo": -(target_density + self.gmmvi.temperature * entropy),
            "entropy": entropy,
            "target_density": target_density,
            "algo_time": np.sum(self.wall_times),
        }
        metrics.update(selector.target_distribution.expensive_metrics(self.gmmvi.model, samples))
        if self.mmd is not None:
            metrics["MMD:"] = self.mmd.compute_MMD(samples)
        return metrics

    # ---- one iteration ----------------------------------------------------------------------------------------------
    def _timed_train_iter(self):
        """The reference's train_iter returns when the step is done; here the launches are asynchronous, so the stream is
        drained before the clock is read."""
        start = time.time()
        self.gmmvi.train_iter()
        self.gmmvi.model.ctx.sync()
        elapsed = time.time() - start
        self.wall_times.append(elapsed)
        return elapsed

    def iterate_and_log(self, n: int) -> dict:
        """:146-175."""
        output = {"walltime": self._timed_train_iter()}
        output.update(self.get_cheap_metrics())
        if n % self.log_metrics_interval == 0:
            evaluated = self.get_expensive_metrics()
            print(_CHECKPOINT_LINE.format(n, output["num_samples"], evaluated["target_density"], -evaluated["-elbo"]))
            print(f"{self.gmmvi.model.num_components} components\n")
            output.update(evaluated)
        return output

    # ---- dumps ----------------------------------------------------------------------------------------------------------
    def log_to_disk(self, n: int):
        """:177-190."""
        if self._dumper is not None and _GmmDumper.wanted(n):
            self._dumper.write("gmm_dump_" + str("%01d" % n) + ".npz", self.gmmvi)

    def finalize(self):
        """:192-200."""
        if self._dumper is not None:
            self._dumper.write("final_gmm_dump.npz", self.gmmvi)
