"""GmmviRunner (reference: src/gmmvi/gmmvi_runner.py:23-200): seeds, timing, metrics and npz dumps around
GMMVI.train_iter(); same constructor, ``build_from_config``, ``iterate_and_log``, ``log_to_disk`` and ``finalize``."""
import os
import random
from time import time

import numpy as np

from .optimization.gmmvi import GMMVI
from .experiments.setup_experiment import init_experiment


class GmmviRunner:
    def __init__(self, config, log_metrics_interval):
        if "seed" not in config.keys():
            config["seed"] = config["start_seed"]
        # tf.keras.utils.set_random_seed(seed) seeds python, numpy and TF (gmmvi_runner.py:38); the device streams
        # are Philox keyed by the same seed
        random.seed(config["seed"])
        np.random.seed(config["seed"])
        self.wall_times = []
        self.config = config
        self.log_metrics_interval = log_metrics_interval
        target_distribution, initial_model = init_experiment(self.config)
        initial_model.model.seed = int(config["seed"])
        self.gmmvi = GMMVI.build_from_config(self.config, target_distribution, initial_model)
        if "mmd_evaluation_config" in config.keys():                                       # gmmvi_runner.py:45-54
            from .experiments.evaluation.mmd import MMD
            dir_path = os.path.dirname(os.path.realpath(__file__))
            samples = np.load(os.path.join(dir_path, config['mmd_evaluation_config']['sample_dir']))
            self.mmd = MMD(samples, config['mmd_evaluation_config']["alpha"])
        else:
            self.mmd = None
        if "dump_gmm_path" not in self.config:
            self.dump_gmms = False
        else:
            self.dump_gmms = True
            self.dump_gmm_path = os.path.join(self.config["dump_gmm_path"], str(time()))
            os.makedirs(self.dump_gmm_path, exist_ok=True)

    @staticmethod
    def build_from_config(config: dict):
        """gmmvi_runner.py:63-81."""
        return GmmviRunner(config=config, **config['gmmvi_runner_config'])

    def get_samples_and_entropy(self, num_samples):
        """gmmvi_runner.py:83-100."""
        test_samples = self.gmmvi.model.sample(num_samples)[0]
        entropy = -float(np.mean(self.gmmvi.model.log_density(test_samples).numpy()))
        return test_samples, entropy

    def get_cheap_metrics(self):
        """gmmvi_runner.py:102-117."""
        return {"num_samples": self.gmmvi.sample_db.num_samples_written.numpy(),
                "num_components": self.gmmvi.model.num_components,
                "max_weight": float(np.max(self.gmmvi.model.weights)),
                "num_db_samples": self.gmmvi.sample_db.samples.shape[0],
                "num_db_components": self.gmmvi.sample_db.means.shape[0]}

    def get_expensive_metrics(self):
        """gmmvi_runner.py:119-144: ELBO = E_q[log p~] + temperature * H(q) on 2000 fresh samples."""
        expensive_metrics = dict()
        test_samples, entropy = self.get_samples_and_entropy(2000)
        lp = self.gmmvi.sample_selector.target_uld(test_samples)
        mean_reward = float(np.mean(np.asarray(lp.numpy() if hasattr(lp, "numpy") else lp)))
        elbo = mean_reward + self.gmmvi.temperature * entropy
        expensive_metrics.update({"-elbo": -elbo, "entropy": entropy, "target_density": mean_reward,
                                  "algo_time": np.sum(self.wall_times)})
        expensive_metrics.update(
            self.gmmvi.sample_selector.target_distribution.expensive_metrics(self.gmmvi.model, test_samples))
        if self.mmd is not None:                                                           # gmmvi_runner.py:140-142
            expensive_metrics.update({"MMD:": self.mmd.compute_MMD(test_samples)})
        return expensive_metrics

    def iterate_and_log(self, n: int) -> dict:
        """gmmvi_runner.py:146-175."""
        output_dict = {}
        ts1 = time()
        self.gmmvi.train_iter()
        self.gmmvi.model.ctx.sync()          # the reference's train_iter returns when the step is done
        ts2 = time()
        output_dict.update({"walltime": ts2 - ts1})
        self.wall_times.append(ts2 - ts1)
        output_dict.update(self.get_cheap_metrics())
        if n % self.log_metrics_interval == 0:
            eval_dict = self.get_expensive_metrics()
            print("Checkpoint {:3d} | FEVALS: {:10d} | avg. sample logpdf: {:05.05f} | ELBO: {:05.05f}".format(
                n, output_dict["num_samples"], eval_dict["target_density"], -eval_dict["-elbo"]))
            print(f"{self.gmmvi.model.num_components} components\n")
            output_dict.update(eval_dict)
        return output_dict

    def log_to_disk(self, n: int):
        """gmmvi_runner.py:177-190."""
        if self.dump_gmms and (n < 100 or n % 50 == 0):
            m = self.gmmvi.model
            np.savez(self.dump_gmm_path + '/gmm_dump_' + str("%01d" % n) + '.npz',
                     weights=np.exp(m.log_weights.numpy()), means=m.means.numpy(), covs=m.covs, timestamps=time(),
                     fevals=self.gmmvi.sample_db.num_samples_written.numpy())

    def finalize(self):
        """gmmvi_runner.py:192-200."""
        if self.dump_gmms:
            m = self.gmmvi.model
            np.savez(self.dump_gmm_path + '/final_gmm_dump.npz', weights=m.weights, means=m.means.numpy(),
                     covs=m.covs, timestamps=time(), fevals=self.gmmvi.sample_db.num_samples_written.numpy())
