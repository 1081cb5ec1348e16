"""Developer tool: which small adaptive configuration makes added components come alive AND deletes some (test design for
tests/test_hip_sharded.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import samtron_config, make_oracle, make_device
from gmmvi_amd.device import get_context
from gmmvi_amd.sharded import HipOps, LocalExchange
from gmmvi_amd.sharded_adaptive import ShardedAdaptiveGMMVI

for kind, d, k, s, seed, ad_over, iters in [
        ("gmm", 4, 2, 60, 11, {}, 45), ("gmm", 4, 2, 60, 5, {"thresholds_for_add_heuristic": [500., 100., 20.]}, 45),
        ("stm", 6, 2, 60, 3, {"thresholds_for_add_heuristic": [500., 100., 20.]}, 45),
        ("gmm", 6, 3, 60, 7, {"thresholds_for_add_heuristic": [1000., 200., 50.], "add_iters": 3, "del_iters": 9}, 60)]:
    ad = {"del_iters": 6, "add_iters": 2, "max_components": 14, "thresholds_for_add_heuristic": [50., 20., 10.],
          "min_weight_for_del_heuristic": 1e-3, "num_database_samples": 300, "num_prior_samples": 0}
    ad.update(ad_over)
    cfg = samtron_config(s, adaptive=ad)
    o = make_oracle(kind, d, k, s, seed, cfg)
    g = make_device(kind, d, k, s, seed, cfg, o)
    cfg = dict(cfg, model_initialization=dict(cfg["model_initialization"], prior_mean=0.0,
                                              initial_cov=g.num_component_adapter.prior_var.tolist()))
    sh = ShardedAdaptiveGMMVI(HipOps(get_context(), g.sample_selector.target_distribution), LocalExchange(), d,
                              g.model.means.numpy(), g.model.chol_cov.numpy(), s, seed, cfg, history_length=400)
    deleted = 0
    for it in range(iters):
        before = set(sh.unique_component_ids.tolist())
        sh.train_iter()
        deleted += len(before - set(sh.unique_component_ids.tolist()))
    w = np.exp(sh.log_weights.numpy())
    print(kind, d, k, seed, ad_over, "-> K", sh.num_components, "deleted", deleted, "ids", sh.unique_component_ids.tolist(),
          "weights", np.round(w, 4).tolist())
