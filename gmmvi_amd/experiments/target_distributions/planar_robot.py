"""Planar n-link robot target (reference: src/gmmvi/experiments/target_distributions/planar_robot.py:11-138)."""
import numpy as np

from ... import hip_ops
from ...device import get_context
from .lnpdf import LNPDF


class PlanarRobot(LNPDF):
    def __init__(self, num_links, num_goals, prior_std=2e-1, likelihood_std=1e-2):
        super().__init__(use_log_density_and_grad=True)
        self.ctx = get_context()
        self._num_dimensions = num_links
        prior_stds = prior_std * np.ones(num_links)
        prior_stds[0] = 1.0                                                      # planar_robot.py:32-33
        self.prior_stds = prior_stds.astype(np.float32)
        self.link_lengths = np.ones(num_links)
        self._num_goals = num_goals
        if num_goals == 1:
            self.goals = np.array([[7., 0.]], np.float32)
        elif num_goals == 4:
            self.goals = np.array([[7., 0.], [-7., 0.], [0., 7.], [0., -7.]], np.float32)   # :40
        else:
            raise ValueError
        self.likelihood_std = float(likelihood_std)
        self._prior_dev = self.ctx.asarray(self.prior_stds)
        self._goals_dev = self.ctx.asarray(self.goals)

    def get_num_dimensions(self):
        return self._num_dimensions

    def _fast_path_target(self):
        """Descriptor for the single-call iteration (optimization/fused.py)."""
        return {"kind": 1, "prior_std": self._prior_dev.ptr, "goals": self._goals_dev.ptr,
                "G": int(self.goals.shape[0]), "lik_std": self.likelihood_std}

    def forward_kinematics(self, theta):
        """planar_robot.py:58-64 (host; metrics only)."""
        theta = np.asarray(theta.numpy() if hasattr(theta, "numpy") else theta, np.float64)
        c = np.cumsum(theta, axis=1)
        return np.stack([np.cos(c).sum(1), np.sin(c).sum(1)], axis=1)

    def log_density(self, theta):
        return hip_ops.target_planar(self.ctx, self._prior_dev, self._goals_dev, self.likelihood_std,
                                     self.ctx.asarray(theta), want_grad=False)[0]

    def log_density_and_grad(self, theta):
        return hip_ops.target_planar(self.ctx, self._prior_dev, self._goals_dev, self.likelihood_std,
                                     self.ctx.asarray(theta), want_grad=True)

    def expensive_metrics(self, model, samples) -> dict:
        """planar_robot.py:68-132: per goal, count clusters of first-joint angles among good components."""
        out = dict()
        means = model.means.numpy().astype(np.float64)
        fk = self.forward_kinematics(means)
        good = self.log_density(means.astype(np.float32)).numpy() > -7.0
        for goal in self.goals:
            close = np.linalg.norm(fk - goal[None, :], axis=1) < 0.05
            idx = np.where(close & good)[0]
            if idx.size == 0:
                n_modes = 0
            else:
                first = np.sort(means[idx, 0])
                n_modes = 1 + int(np.sum((first[1:] - first[:-1]) > 0.4))
            out[f"num_detected_modes_{goal}"] = n_modes
        return out


def make_single_goal():
    return PlanarRobot(10, 1)


def make_four_goal():
    return PlanarRobot(10, 4)
