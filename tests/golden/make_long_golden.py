"""Generates the LONG-HORIZON fixtures tests/golden/long_*.npz from the fp64 CPU oracle (the reference cannot run here and
ships no vectors: SURVEY.md 8c; these files are oracle output, data only -- "parity unpinned" upstream as DESIGN.md 1 says).

  long_c2.npz   BASELINE configs[1] shape: 20-D Student-t mixture target, K = 50 fixed, 100 samples / component, 120 iterations
  long_c1.npz   BASELINE configs[0] = examples/5_samtron_20D_student-T.py:13-30: K = 45 adaptive (add every 60, delete after
                100 iterations), 200 samples / component, reuse ratio 0, 260 iterations (adds at 60/120/180/240, deletions
                from iteration 101)
  long_c1_full.npz  the same run at the example's stated length, examples/5_samtron_20D_student-T.py:30: 1501 iterations (K 45 -> 70
                by 25 adds, no deletion: the rule never fires on this target; checkpoints every 50 iterations; ~65 minutes)
  long_c4.npz   BASELINE configs[3]'s example as examples/6_samtron_planar4.py:19-26 runs it: planar-4 target, 100 initial
                components, a component added EVERY iteration, deletions from iteration 11 (del_iters 10), 100 samples /
                component, weight stepsize 5, 140 iterations; the script asserts that the oracle deleted >= 5 components;
                additionally `id_trace` (the unique component ids after every iteration, -1 padded)
  long_ns.npz   the north-star workload exactly as bench.py builds it (bench.spec("ns"): K = 100, D = 20, N = 10 000), 60
                iterations: what bench.py's matched_elbo leg compares the device trajectory with (no oracle iterations
                inside the bench run)
  pair_c5.npz   the per-GPU shard shape of BASELINE configs[4] that bench.py --workload c5 composes, cut to K = 8 components:
                single-Gaussian target D = 300 (gmm.py:148-162 law), 312 samples / component, 2 iterations

Stored per case: the INPUTS (target parameters, initial mixture, seeds, hyper-parameters are in tests/helpers.py LONG_CASES)
and, every `every` iterations, the oracle's ELBO on a FIXED evaluation set of 20 000 Philox draws (gmmvi_runner.py:131-133
definition), the standard error of that estimate, the number of components; per iteration the number of components and the
count of accepted component updates; at the end the mixture.

Run:  python tests/golden/make_long_golden.py [c2 c1 pair_c5]      (about 15 minutes on 8 cores)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from helpers import LONG_CASES, make_long_oracle, score_elbo, samtron_config, make_oracle  # noqa: E402


def run_long(case, dtype=np.float64, verbose=True):
    o = make_long_oracle(case, dtype=dtype)
    t = o.target
    out = {"init_means": o.model.means.astype(np.float32), "init_covs": o.model.model.covs.astype(np.float32)}
    if hasattr(t, "weights"):                        # (the planar-robot target has no parameters beyond its construction)
        out.update(target_weights=t.weights, target_means=t.means, target_covs=t.covs)
    cp_it, cp_elbo, cp_sigma, cp_k, k_trace, n_success, id_trace, n_deleted = [], [], [], [], [], [], [], 0
    t0 = time.time()
    for it in range(1, case["iters"] + 1):
        ids_before = set(o.model.unique_component_ids)
        info = o.train_iter()
        k_trace.append(o.model.num_components)
        id_trace.append(np.array(o.model.unique_component_ids))
        n_deleted += len(ids_before - set(o.model.unique_component_ids))
        n_success.append(int(np.sum(info["success"])))
        if it % case["every"] == 0:
            m = o.model.model
            e, sg = score_elbo(t, m.log_weights, m.means, m.chol_cov)
            cp_it.append(it); cp_elbo.append(e); cp_sigma.append(sg); cp_k.append(m.num_components)
            if verbose:
                print(f"  it {it:4d}  K {m.num_components:3d}  elbo {e:12.4f} +- {sg:.4f}  ({time.time() - t0:.0f} s)", flush=True)
    m = o.model.model
    out.update(checkpoint_iters=np.array(cp_it), checkpoint_elbo=np.array(cp_elbo), checkpoint_sigma=np.array(cp_sigma),
               checkpoint_k=np.array(cp_k), k_trace=np.array(k_trace), n_success=np.array(n_success),
               final_log_weights=m.log_weights.copy(), final_means=m.means.copy(), final_chols=m.chol_cov.copy(),
               final_component_ids=np.array(o.model.unique_component_ids), n_deleted=np.array(n_deleted))
    width = max(len(v) for v in id_trace)
    out["id_trace"] = np.stack([np.pad(v, (0, width - len(v)), constant_values=-1) for v in id_trace]).astype(np.int32)
    if verbose:
        print(f"  components deleted over the run: {n_deleted}", flush=True)
    if case.get("min_deleted"):
        assert n_deleted >= case["min_deleted"], f"the oracle deleted only {n_deleted} components"
    return out


def run_bench_workload(workload, iters, every):
    """ELBO checkpoints of the fp64 oracle on a bench.py workload (same construction as bench.make_oracle)."""
    import bench
    w = bench.spec(workload, 1)
    o = bench.make_oracle(w)
    t = o.target
    out = {"init_means": w["means"], "target_means": t.means, "workload": workload}
    cp_it, cp_elbo, cp_sigma = [], [], []
    t0 = time.time()
    for it in range(1, iters + 1):
        o.train_iter()
        if it % every == 0:
            m = o.model.model
            e, sg = score_elbo(t, m.log_weights, m.means, m.chol_cov)
            cp_it.append(it); cp_elbo.append(e); cp_sigma.append(sg)
            print(f"  it {it:4d}  elbo {e:12.4f} +- {sg:.4f}  ({time.time() - t0:.0f} s)", flush=True)
    m = o.model.model
    out.update(checkpoint_iters=np.array(cp_it), checkpoint_elbo=np.array(cp_elbo), checkpoint_sigma=np.array(cp_sigma),
               final_log_weights=m.log_weights.copy(), final_means=m.means.copy())
    return out


def run_pair_c5():
    """D = 300, K = 8, S = 312: the state after each of 2 iterations (what tests/helpers.run_pair compares)."""
    cfg = samtron_config(312, initial_stepsize=0.1)
    o = make_oracle("gauss", 300, 8, 312, 7, cfg)
    t = o.target
    out = {"init_means": o.model.means.astype(np.float32), "init_covs": o.model.model.covs.astype(np.float32),
           "target_weights": t.weights, "target_means": t.means, "target_covs": t.covs}
    rec = {key: [] for key in ("means", "chols", "log_weights", "stepsizes", "success", "n_probes", "rewards", "elr")}
    for _ in range(2):
        info = o.train_iter()
        rec["means"].append(o.model.means.copy()); rec["chols"].append(o.model.chol_cov.copy())
        rec["log_weights"].append(o.model.log_weights.copy()); rec["stepsizes"].append(o.model.stepsizes.copy())
        rec["success"].append(info["success"].copy()); rec["n_probes"].append(info["n_probes"].copy())
        rec["rewards"].append(o.model.reward_history[:, -1].copy()); rec["elr"].append(info["expected_log_ratios"].copy())
    out.update({key: np.array(v) for key, v in rec.items()})
    for key in ("chols", "init_covs", "target_covs"):
        out[key] = np.asarray(out[key], np.float32)        # 8 x 300 x 300 per iteration: keep the file small
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c1", "ns"]
    for name in which:
        print("case", name, flush=True)
        res = (run_pair_c5() if name == "pair_c5" else run_bench_workload("ns", 60, 10) if name == "ns"
               else run_long(LONG_CASES[name]))
        path = os.path.join(HERE, ("" if name == "pair_c5" else "long_") + f"{name}.npz")
        np.savez_compressed(path, **res)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB", flush=True)
