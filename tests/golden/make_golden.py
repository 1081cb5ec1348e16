"""Generates tests/golden/*.npz from the CPU oracle (the reference cannot run here and ships no vectors: SURVEY.md 8c).
Each file stores the INPUTS (target parameters, initial mixture, seed, hyper-parameters) and the oracle's per-iteration
OUTPUTS of SAMTRON train_iter() -- data only.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from helpers import samtron_config, make_oracle  # noqa: E402

CASES = [  # (name, target kind, D, K, samples/component, seed, iterations)
    ("stm_K3_D4", "stm", 4, 3, 32, 11, 20),
    ("gmm_K3_D4", "gmm", 4, 3, 32, 11, 20),
    ("planar_K4_D10", "planar", 10, 4, 50, 11, 20),
    ("stm_K8_D20", "stm", 20, 8, 64, 11, 20),
]


def run_case(kind, d, k, s, seed, iters):
    cfg = samtron_config(s)
    o = make_oracle(kind, d, k, s, seed, cfg)
    t = o.target
    out = {"kind": kind, "d": d, "k": k, "s": s, "seed": seed, "iters": iters,
           "init_means": o.model.means.copy(), "init_covs": o.model.covs.copy(),
           "init_weights": o.model.weights.copy()}
    if kind != "planar":
        out.update(target_weights=t.weights, target_means=t.means, target_covs=t.covs)
    rec = {key: [] for key in ("means", "chols", "log_weights", "stepsizes", "last_etas", "success", "n_probes",
                               "rewards", "elr", "weight_stepsize", "h_neg", "g_neg")}
    for _ in range(iters):
        info = o.train_iter()
        rec["means"].append(o.model.means.copy()); rec["chols"].append(o.model.chol_cov.copy())
        rec["log_weights"].append(o.model.log_weights.copy()); rec["stepsizes"].append(o.model.stepsizes.copy())
        rec["last_etas"].append(o.model.last_log_etas.copy()); rec["success"].append(info["success"].copy())
        rec["n_probes"].append(info["n_probes"].copy()); rec["rewards"].append(o.model.reward_history[:, -1].copy())
        rec["elr"].append(info["expected_log_ratios"].copy()); rec["weight_stepsize"].append(info["weight_stepsize"])
        rec["h_neg"].append(info["h_neg"].copy()); rec["g_neg"].append(info["g_neg"].copy())
    out.update({key: np.array(v) for key, v in rec.items()})
    out["elbo"] = np.array(o.elbo(4000, seed=5))
    return out


if __name__ == "__main__":
    for name, kind, d, k, s, seed, iters in CASES:
        np.savez_compressed(os.path.join(HERE, f"samtron_{name}.npz"), **run_case(kind, d, k, s, seed, iters))
        print("wrote", name)
