"""Developer tool: what differs between device and oracle for a component in the iterations right after it was added."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import LONG_CASES, make_long_oracle, make_long_device
case = dict(LONG_CASES["c4"], k=int(os.environ.get("TK", 20)))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 16
o = make_long_oracle(case)
g = make_long_device(case, o)
if len(sys.argv) > 2 and sys.argv[2] == "modular":
    g.ng_based_updater.want_info = True
np.set_printoptions(linewidth=220, precision=5)
for it in range(1, iters + 1):
    k_before = o.model.num_components
    info = o.train_iter(); g.train_iter()
    # state after the updates of this iteration but BEFORE the add is gone; compare the components that existed during the iteration
    om, gm = o.model, g.model
    kk = k_before
    dm = np.abs(gm.means.numpy()[:kk] - om.means[:kk]).max(axis=1) / (1e-3 + np.abs(om.means[:kk]).max(axis=1))
    dc = np.abs(gm.chol_cov.numpy()[:kk] - om.chol_cov[:kk]).reshape(kk, -1).max(axis=1) / np.abs(om.chol_cov[:kk]).reshape(kk, -1).max(axis=1)
    de = np.abs(gm.last_log_etas.numpy()[:kk] - om.last_log_etas[:kk]) / (1e-6 + np.abs(om.last_log_etas[:kk]))
    rg, ro = gm.reward_slot(0).numpy()[:kk], om.reward_history[:kk, -1]
    dr = np.abs(rg - ro) / (1 + np.abs(ro))
    new = slice(max(0, kk - 3), kk)
    print(f"it {it:2d} K {kk}: max rel dev over OLD comps: mean {dm[:-3].max():.1e} chol {dc[:-3].max():.1e} eta {de[:-3].max():.1e} reward {dr[:-3].max():.1e} | "
          f"newest 3: mean {dm[new]} chol {dc[new]} eta {de[new]} reward {dr[new]}  eta_o {om.last_log_etas[new]} eta_g {gm.last_log_etas.numpy()[new]}"
          f" succ_o {info['success'][new].astype(int)} nprobe_o {info['n_probes'][new]}", flush=True)
