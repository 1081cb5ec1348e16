"""Developer tool: runs the long_c4 case (examples/6 regime) on the oracle and on the device side by side and prints, per
iteration, the deletion decisions and how close each criterion of component_adaptation.py:261-300 was to its threshold."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import LONG_CASES, make_long_oracle, make_long_device
from scipy.special import logsumexp

case = LONG_CASES["c4"]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 70
o = make_long_oracle(case)
g = make_long_device(case, o)
F32MAX = float(np.finfo(np.float32).max)


def criteria(rh, wh, ks, kern, del_iters):
    cur = np.mean(rh[:, -ks:] * kern[None, :], axis=1)
    old = np.mean(rh[:, -ks - del_iters:-del_iters] * kern[None, :], axis=1)
    old = old - np.max(cur); cur = cur - np.max(cur)
    with np.errstate(all="ignore"):
        imp = (cur - old) / np.abs(old)
        win = rh[:, -ks - del_iters:]
        greedy = np.max(np.exp(win - logsumexp(win, axis=0, keepdims=True)), axis=1)
    actual = np.max(wh[:, -ks - del_iters:-1], axis=1)
    full = np.all(rh[:, -ks - del_iters:] != -F32MAX, axis=1)     # the whole window holds real rewards
    return imp, np.maximum(actual, greedy), rh[:, -del_iters] != -F32MAX, full


oa, ga = o.num_component_adapter, g.num_component_adapter
for it in range(1, iters + 1):
    o.train_iter(); g.train_iter()
    ko, kg = o.model.num_components, g.model.num_components
    same = np.array_equal(o.model.unique_component_ids, g.model.unique_component_ids)
    if it >= case["adaptive"]["del_iters"]:
        ks = oa.kernel.size
        io, wo, ao, fo = criteria(o.model.reward_history, o.model.weight_history, ks, oa.kernel, oa.del_iters)
        win = ks + ga.del_iters
        rg, wg = g.model.reward_window(win).astype(np.float64), g.model.weight_window(win).astype(np.float64)
        ig, wgm, ag, fg = criteria(rg, wg, ks, ga.kernel.astype(np.float64), ga.del_iters)
        if same:
            near_imp = np.nanmin(np.abs(io - 0.4)[ao & (wo < 1e-6)]) if np.any(ao & (wo < 1e-6)) else np.inf
            near_w = np.nanmin(np.abs(wo / 1e-6 - 1)[ao & (io <= 0.4)]) if np.any(ao & (io <= 0.4)) else np.inf
            dmax = np.nanmax(np.abs(io - ig)[fo]) if fo.any() else 0
            wrel = np.nanmax((np.abs(wo - wgm) / np.maximum(wo, 1e-30))[fo]) if fo.any() else 0
            print(f"it {it:3d} K {ko} / {kg}: next-call margins: improvement {near_imp:.2e}, weight (relative) {near_w:.2e}; "
                  f"oracle - device: improvement {dmax:.2e}, max weight rel {wrel:.2e}", flush=True)
    if not same:
        print(f"it {it}: ids differ: only oracle {sorted(set(o.model.unique_component_ids) - set(g.model.unique_component_ids))}, "
              f"only device {sorted(set(g.model.unique_component_ids) - set(o.model.unique_component_ids))}")
        break
