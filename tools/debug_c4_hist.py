"""Developer tool: compares the reward / weight history windows of the device with the oracle's, column by column."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import LONG_CASES, make_long_oracle, make_long_device
case = dict(LONG_CASES["c4"])
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 25
o = make_long_oracle(case)
g = make_long_device(case, o)
if len(sys.argv) > 2 and sys.argv[2] == "modular":
    g.ng_based_updater.want_info = True
oa = o.num_component_adapter
win = oa.kernel.size + oa.del_iters
np.set_printoptions(linewidth=250, precision=4, suppress=False)
for it in range(1, iters + 1):
    o.train_iter(); g.train_iter()
    if it >= iters - 2:
        rg, wg = g.model.reward_window(win).astype(np.float64), g.model.weight_window(win).astype(np.float64)
        ro, wo = o.model.reward_history[:, -win:], o.model.weight_history[:, -win:]
        print(f"it {it}: K {o.model.num_components} / {g.model.num_components}; window {win} columns (oldest first)")
        valid = (ro > -1e37) & (rg > -1e37)
        print("  sentinel pattern equal:", np.array_equal(ro > -1e37, rg > -1e37))
        dr = np.where(valid, np.abs(ro - rg) / (1 + np.abs(ro)), 0)
        print("  reward: max rel diff per column:", dr.max(axis=0))
        dw = np.abs(wo - wg) / np.maximum(wo, 1e-30)
        print("  weight: max rel diff per column:", dw.max(axis=0))
        k = int(np.argmax(dw.max(axis=1)))
        print(f"  worst weight row {k}: oracle {wo[k]}\n                     device {wg[k]}")
        k = int(np.argmax(dr.max(axis=1)))
        print(f"  worst reward row {k}: oracle {ro[k]}\n                     device {rg[k]}")
