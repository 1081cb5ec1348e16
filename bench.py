#!/usr/bin/env python3
"""bench.py -- gmmvi hot path on MI355X: `python bench.py --gpus N --steps K --warmup W`.

A "step" is one GMMVI.train_iter() (sample selection + target evaluation + background density -> model density and
gradient -> Stein natural-gradient estimate -> KL-constrained component update -> weight update) on synthetic inputs.
Metric (BASELINE.json): samples*components/sec (= N_samples * K_components / t_iter, whole job) with train_iter/sec
beside it.  Workload at 1 GPU: the north-star shape, 20-D Student-t mixture target, K = 100 components, 100 samples per
component (N = 10 000 samples/iter), SAMTRON design choices with a fixed number of components and reuse ratio 0.
With --gpus N > 1 the components are sharded over the ranks (100 per GPU, weak scaling at a fixed ~10 000 samples per
iteration: desired samples per component = ceil(100 / N)); one process per GPU, RCCL exchanges (gmmvi_amd/sharded.py).

One JSON line on stdout (rank 0).  Extra objects: "roofline" (dominant kernel, algorithmic FLOPs / HIP-event time),
"cpu_baseline" (the NumPy oracle = CPU port of the reference algorithm, timed on this host on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_FP32_TFLOPS = 157.3      # MI355X fp32 vector == f32-input MFMA rate (MI355X_MICROARCH.md, chip-level parameters)
PEAK_FP64_MFMA_TFLOPS = 78.6  # MI355X fp64 matrix rate (half the f32 MFMA rate)
# Blocked path (D > 50), default route: every f32 contraction runs as SIX bf16 x bf16 partial products of 3-way split operands
# on the bf16 matrix cores (2516.6 TFLOP/s dense = 256 CUs x 4 x 1024 FLOP/clk x 2.4 GHz) with f32 accumulation; results carry
# f32-contraction accuracy (tests/test_hip_blocked.py::test_split_operand_route_is_as_accurate_as_the_f32_route).  Its kernels
# are priced against the route's own ceiling in f32-equivalent FLOPs; GMMVI_BLOCKED_F32=1 selects the f32 MFMA route (157.3).
PEAK_SPLIT_BF16_TFLOPS = 2516.6 / 6.0


def kernel_peak(name):
    if name == "more_gram":
        return PEAK_FP64_MFMA_TFLOPS
    if name.startswith("blocked_") and os.environ.get("GMMVI_BLOCKED_F32", "0") in ("", "0"):
        return PEAK_SPLIT_BF16_TFLOPS
    return PEAK_FP32_TFLOPS
PEAK_HBM_GBS = 8000.0


def kernel_bound(name, d):
    """Which unit bounds a kernel: the scalar-fed density sweeps (padded D <= 24: csrc/density.hip mixture_eval_kernel /
    mixture_eval_pk_kernel) are vector-ALU kernels (v_fma_f32 / v_pk_fma_f32 fed from scalar registers), every other
    FLOP-carrying kernel contracts on the matrix cores."""
    return "valu" if name.startswith("sweep_") and d <= 24 else "mfma"

WORKLOADS = {
    # id: (target kind, D, K per GPU, samples per component at 1 GPU)
    "ns": ("stm", 20, 100, 100),       # north star: K=100, D=20, N=10k
    "c2": ("stm", 20, 50, 100),        # BASELINE configs[1]: K=50, N=5000
    "c3": ("gmm", 50, 100, 100),       # BASELINE configs[2]: GMM target D=50, K=100, N=10k
    "c4": ("planar", 10, 200, 100),    # BASELINE configs[3] on one GPU: planar-4, K=200, N=20k
    # the north-star shape with the reference's DEFAULT selector (component-based.yml:3-4: reuse ratio 2): ~3 N active samples,
    # data-dependent numbers of new samples (one [K] read-back per iteration), separate background sweep over ~3 K snapshots
    "ns_reuse2": ("stm", 20, 100, 100, "Stein", 2.0),
    "ns_more": ("stm", 20, 100, 100, "MORE"),   # north-star shape with the MORE estimator ("ZAMTRON") instead of Stein
    "c3_more": ("gmm", 50, 100, 100, "MORE"),   # C3 shape with MORE: F = 1326 features per component (tiled Gram + blocked fp64 Cholesky)
    # BASELINE configs[4] per GPU: D=300 single-Gaussian target (make_target_with_scale, gmm.py:148-162), K=512 over 8 GPUs
    # = 64 components per GPU, N = 20k samples/iter; blocked path (csrc/blocked.hip).  No CPU baseline: the oracle's
    # [K,N,D] fp64 temporaries make one iteration take minutes ("C5 infeasible on CPU")
    "c5": ("gauss300", 300, 64, 312),
    "c5_k8": ("gauss300", 300, 8, 312),      # the c5 shard cut to 8 components: what tests/test_hip_blocked.py checks against the oracle
    "c5_full": ("gauss300", 300, 512, 39),   # all of BASELINE configs[4] on ONE GPU (component chunks: Z of 512 components > scratch budget)
    # dimension sweep at the C3 shape (crossover of the register-resident and the blocked kernels, GMMVI_BLOCKED_ABOVE)
    "d32": ("gmm", 32, 100, 100), "d40": ("gmm", 40, 100, 100), "d63": ("gmm", 63, 100, 100),
    "tiny": ("stm", 4, 4, 16),         # host-overhead probe (kernels are empty; time = launch path)
    # BASELINE configs[3]'s example AS IT IS RUN (examples/6_samtron_planar4.py:19-26): 100 initial components, a component
    # added EVERY iteration, deletions from iteration 11 on, weight stepsize 5 -- K changes every iteration: what the packing,
    # ring growth and eligibility checks of an adaptive run cost (timed behind the warm-up iterations; value = mean N K / t)
    "c4_adaptive": ("planar", 10, 100, 100, "Stein", 0.0,
                    {"del_iters": 10, "add_iters": 1, "max_components": 1000,
                     "thresholds_for_add_heuristic": [5000., 1000., 500., 200., 100., 50.],
                     "min_weight_for_del_heuristic": 1e-6, "num_database_samples": 100000, "num_prior_samples": 0}),
}
# Strong-scaling workloads: the BASELINE configurations that NAME a GPU count, as stated -- K is the TOTAL number of
# components, split evenly over the ranks (K % gpus == 0), the samples per component do not change with the rank count.
STRONG_WORKLOADS = {
    "c4_sharded": ("planar", 10, 200, 100),      # BASELINE configs[3]: planar-4, K = 200 over 4 GPUs = 50 / GPU, N = 20 000
    "c5_sharded": ("gauss300", 300, 512, 39),    # BASELINE configs[4]: D = 300, K = 512 over 8 GPUs = 64 / GPU, N = 19 968
}
WORKLOADS.update(STRONG_WORKLOADS)


def kernel_flops(name, n, k, d):
    """Algorithmic FLOPs of one launch: SURVEY.md 8(d)'s per-pair figures (pairs P = n*k) -- the work the reference's
    algorithm needs for that step, whatever route the kernel takes."""
    p = float(n) * k
    return {
        # density sweeps (csrc/density.hip names its launches: the single-call iteration tags them, the plug-in path by shape)
        "sweep_values": p * (d * d + 4 * d),              # forward substitution + square-sum + LSE
        "sweep_post": p * (d * d + 4 * d),                # the post-update sweep of the weight step (log values only)
        "sweep_grad": p * (2 * d * d + 8 * d),            # + backward substitution + responsibility-weighted gradient
        "sweep_target": p * (2 * d * d + 8 * d),          # the target evaluation (mixture targets: log value + gradient)
        "sweep_dual": p * (2 * d * d + 8 * d),            # model log q + gradient + background in one pass over the components
        # sample reuse (the reference's default selector): the background mixture over the window's snapshot components (log
        # values only) and the model sweep with the gradient are separate launches of the single-call iteration
        "sweep_background": p * (d * d + 4 * d),
        "sweep_model": p * (2 * d * d + 8 * d),
        "stein_partial": p * (4 * d * d + 6 * d),         # y per sample (2D^2) + rank-1 accumulate (2D^2)
        # MORE: lower triangle of the (F+1)x(F+1) Gram matrix of [phi; reward], F = D(D+1)/2 + D + 1 (2 flop per MAC)
        "more_gram": p * ((d * (d + 1) // 2 + d + 2) * (d * (d + 1) // 2 + d + 3) + d * d),
        # blocked path (D > 64): triangular whitening Z = (X - mu) L^-T, gradient sum_k r Z L^-1, Stein sum e [g;1][z;1]^T
        "blocked_forward": p * (d * d + 2 * d),
        "blocked_grad": p * (d * d + d),
        "blocked_stein_accumulate": p * 2 * (d + 1) * (d + 1),
    }.get(name)


def kernel_flops_executed(name, n, k, d):
    """FLOPs the kernel actually executes where that is LESS than the algorithmic figure (reported beside it, never instead):
    the Stein estimate in moment form accumulates [g;1][x - mu;1]^T (2 (D+1)^2 + 3D per pair) and applies Sigma^-1 once
    per component afterwards -- the per-sample substitution is gone."""
    p = float(n) * k
    return {"stein_partial": p * (2 * (d + 1) * (d + 1) + 3 * d)}.get(name, kernel_flops(name, n, k, d))


def spec(workload, n_gpus, seed=0):
    """Host-only description of a workload (no GPU needed: the CPU baseline and the tests use it too): oracle target,
    initial mixture, config."""
    from helpers import samtron_config
    from oracle import targets as otargets
    kind, d, k_per_gpu, s1 = WORKLOADS[workload][:4]
    extra = WORKLOADS[workload][4:]
    estimator = extra[0] if len(extra) > 0 else "Stein"
    reuse = float(extra[1]) if len(extra) > 1 else 0.0
    adaptive = extra[2] if len(extra) > 2 else None
    if workload in STRONG_WORKLOADS:
        if k_per_gpu % n_gpus:
            raise SystemExit(f"bench.py: workload {workload} splits K = {k_per_gpu} components evenly; --gpus {n_gpus} does not divide it")
        k_total, s = k_per_gpu, s1
    else:
        k_total = k_per_gpu * n_gpus
        s = int(np.ceil(s1 / n_gpus))
    # the target's parameters come from the PRODUCT's own constructors (the reference's laws on the global NumPy RNG,
    # gmmvi_amd/experiments/target_distributions/{student_t_mixture,gmm}.py: host arrays, no GPU); the oracle target is built
    # from those arrays and serves the CPU baseline and the matched-ELBO scorer only
    from gmmvi_amd.experiments.target_distributions import student_t_mixture as p_stm, gmm as p_gmm
    state = np.random.get_state()
    np.random.seed(seed)
    try:
        if kind == "stm":
            ot = otargets.StudentTMixtureTarget(*p_stm.make_target_parameters(d, False), alpha=2)
            prior_scale, initial_cov = 100.0, 300.0                    # stm20.yml:9-14
        elif kind == "gmm":
            ot = otargets.GmmTarget(*p_gmm.make_target_parameters(d))
            prior_scale, initial_cov = 31.63, 1000.0                   # gmm20.yml:7-12
        elif kind == "gauss300":
            ot = otargets.GmmTarget(*p_gmm.make_target_with_scale_parameters(d, 1, 1.0))
            prior_scale, initial_cov = 100.0, 300.0                    # stm300.yml:9-14
        else:
            ot = otargets.PlanarRobotTarget(d, 4)                      # (no random parameters: planar_robot.py:137-138)
            prior_scale, initial_cov = [1.0] + [0.2] * (d - 1), [0.0625] + [0.0025] * (d - 1)   # planar_robot_4.yml
    finally:
        np.random.set_state(state)
    init_rng = np.random.default_rng(seed + 1)
    means = (np.asarray(prior_scale) * init_rng.standard_normal((k_total, d))).astype(np.float32)
    covs = np.broadcast_to((np.asarray(initial_cov) * np.eye(d)).astype(np.float32), (k_total, d, d))
    cfg = samtron_config(s, initial_stepsize=0.1, estimator=estimator, reuse_ratio=reuse, adaptive=adaptive,
                         wstep=5.0 if adaptive else 1.0)                  # examples/6...:24 (the rule caps it at max_stepsize)
    cfg["model_initialization"].update(prior_mean=0.0, initial_cov=initial_cov)
    return dict(kind=kind, d=d, k_total=k_total, s=s, n_total=k_total * s, cfg=cfg, oracle_target=ot,
                means=means, covs=np.ascontiguousarray(covs), seed=seed + 2)


def build(workload, n_gpus, rank, seed=0):
    """spec() plus the device-side target and the classes of the product (needs the GPU)."""
    import gmmvi_amd  # noqa: F401
    from gmmvi_amd.models.full_cov_gmm import FullCovGMM
    from gmmvi_amd.models.gmm_wrapper import GmmWrapper
    from gmmvi_amd.optimization.gmmvi import GMMVI
    from gmmvi_amd.experiments.target_distributions.gmm import GMM_LNPDF
    from gmmvi_amd.experiments.target_distributions.student_t_mixture import StudentTMixture_LNPDF
    from gmmvi_amd.experiments.target_distributions.planar_robot import PlanarRobot
    w = spec(workload, n_gpus, seed)
    ot = w["oracle_target"]
    if w["kind"] == "stm":
        tgt = StudentTMixture_LNPDF(ot.weights, ot.means, ot.covs, alpha=2)
    elif w["kind"] in ("gmm", "gauss300"):
        tgt = GMM_LNPDF(ot.weights, ot.means, ot.covs)
    else:
        tgt = PlanarRobot(w["d"], 4)
    w.update(target=tgt, FullCovGMM=FullCovGMM, GmmWrapper=GmmWrapper, GMMVI=GMMVI)
    return w


def make_gmmvi(w, n_gpus, rank):
    if n_gpus == 1:
        model = w["FullCovGMM"](np.ones(w["k_total"]) / w["k_total"], w["means"], w["covs"])
        model.seed = w["seed"]
        wrapper = w["GmmWrapper"](model, 0.1, 1e-12, 10000)        # setup_experiment.py:40-41 history length
        return w["GMMVI"].build_from_config(w["cfg"], w["target"], wrapper)
    if w["cfg"]["num_component_adapter_type"] == "adaptive":     # K changes: partition by component id (sharded_adaptive.py)
        from gmmvi_amd.sharded_adaptive import ShardedAdaptiveGMMVI
        return ShardedAdaptiveGMMVI.build(w, n_gpus, rank)
    from gmmvi_amd.sharded import ShardedGMMVI
    return ShardedGMMVI.build(w, n_gpus, rank)


def make_oracle(w, dtype=np.float64):
    """The CPU restatement on the workload; ``dtype=np.float32`` runs it in the reference's own arithmetic."""
    from oracle import train as otrain, gmm as ogmm
    cfg = w["cfg"]
    model = ogmm.FullCovGMM(np.ones(w["k_total"]) / w["k_total"], w["means"], w["covs"], dtype=dtype)
    return otrain.OracleGMMVI(w["oracle_target"], model, seed=w["seed"], ng_estimator=cfg["ng_estimator_type"],
                              desired_samples_per_component=w["s"],
                              ratio_reused_samples_to_desired=cfg["sample_selector_config"]["ratio_reused_samples_to_desired"],
                              component_stepsize_config=cfg["component_stepsize_adapter_config"],
                              weight_stepsize_config=cfg["weight_stepsize_adapter_config"],
                              adaptive=(dict(cfg["num_component_adapter_config"], prior_mean=0.0,
                                             initial_cov=cfg["model_initialization"]["initial_cov"])
                                        if cfg["num_component_adapter_type"] == "adaptive" else None))


def time_cpu_baseline(w, seconds, warmup=3, max_iters=20, min_iters=3):
    """BASELINE.md section 3 / SURVEY.md 8(d): the CPU restatement in fp32 (the reference computes in fp32) on this host's
    cores, ``warmup`` untimed iterations, then timed iterations until ``max_iters`` or the time budget (at least
    ``min_iters``); median iteration time.  -> (median seconds, timed iterations, BLAS threads)."""
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    o = make_oracle(w, dtype=np.float32)
    t_begin = time.perf_counter()
    for _ in range(warmup):
        o.train_iter()
    ts = []
    while len(ts) < max_iters and (len(ts) < min_iters or time.perf_counter() - t_begin < seconds):
        t1 = time.perf_counter()
        o.train_iter()
        ts.append(time.perf_counter() - t1)
    return float(np.median(ts)), len(ts), int(threads)


# ---- self-launch: `python bench.py --gpus N` without a launcher starts its own ranks --------------------------------------
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n_ranks, argv, child=None, timeout=900.0, log_dir=None):
    """Start ``n_ranks`` fresh processes of this script (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT in their environment, as torch.distributed.run would set them) and relay rank 0's stdout.
    The calling process has not touched the GPU and never does (children are new processes, nothing is re-exec'ed).
    ALL children are polled: the first rank that exits non-zero ends the job -- the others (possibly parked inside an RCCL
    call that can no longer complete) are terminated and that rank's exit code is returned, its stderr relayed; ``timeout``
    seconds without completion return 124.  Every rank writes stdout / stderr to its own file (rank{r}.out / rank{r}.err in
    ``log_dir``, default a fresh temporary directory), so a failing rank's message is not lost among the others.
    -> exit code: 0 when every rank exited 0."""
    import subprocess
    import tempfile
    cmd = [sys.executable, child or os.path.abspath(__file__)] + list(argv)
    port = _free_port()
    log_dir = log_dir or tempfile.mkdtemp(prefix="gmmvi_bench_ranks_")
    procs, files = [], []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        fo = open(os.path.join(log_dir, f"rank{r}.out"), "wb")
        fe = open(os.path.join(log_dir, f"rank{r}.err"), "wb")
        files += [fo, fe]
        procs.append(subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe))
    rc, failed = 0, None
    t_end = None if timeout is None else time.time() + timeout
    try:
        pending = set(range(n_ranks))
        while pending and rc == 0:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc, failed = code, r
                    break
            if pending and rc == 0:
                if t_end is not None and time.time() > t_end:
                    rc = 124
                    break
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for f in files:
            f.close()

    def read(r, ext):
        with open(os.path.join(log_dir, f"rank{r}.{ext}"), "rb") as f:
            return f.read().decode(errors="replace")
    if rc == 0:
        sys.stdout.write(read(0, "out"))
        sys.stdout.flush()
    else:
        what = f"timed out after {timeout:.0f} s" if failed is None else f"rank {failed} exited with code {rc}"
        sys.stderr.write(f"bench.py: {what}; the other ranks were terminated; per-rank logs in {log_dir}\n")
        for r in ([failed] if failed is not None else range(n_ranks)):
            tail = read(r, "err")[-4000:]
            if tail:
                sys.stderr.write(f"---- rank {r} stderr (tail) ----\n{tail}\n")
    return rc


def matched_elbo(workload, w):
    """"At matched ELBO": a fresh device run from the workload's seed against the fp64 oracle's trajectory.  Where a
    committed oracle fixture exists for the workload (tests/golden/long_<workload>.npz, made by
    tests/golden/make_long_golden.py: ELBO of the oracle every 10 iterations on a fixed set of 20 000 Philox draws) the
    device runs the fixture's full horizon (60 iterations at the north star) and is scored at every checkpoint by the same
    fp64 scorer -- no oracle iterations inside the bench run; otherwise oracle and device run 6 iterations side by side.
    Tolerance: |dELBO| <= 3 sigma_MC + 1e-2 nats."""
    from helpers import score_elbo
    g = make_gmmvi(build(workload, 1, 0), 1, 0)
    path = os.path.join(ROOT, "tests", "golden", f"long_{workload}.npz")
    if os.path.exists(path):
        fx = np.load(path)
        if np.array_equal(fx["init_means"], w["means"]):        # the fixture belongs to this very workload construction
            cps = {int(i): j for j, i in enumerate(fx["checkpoint_iters"])}
            rows, ok = [], True
            for it in range(1, int(fx["checkpoint_iters"][-1]) + 1):
                g.train_iter()
                if it in cps:
                    j = cps[it]
                    e, _ = score_elbo(w["oracle_target"], g.model.log_weights.numpy(), g.model.means.numpy(),
                                      g.model.chol_cov.numpy())
                    tol = 3.0 * float(fx["checkpoint_sigma"][j]) + 1e-2
                    rows.append({"iter": it, "gpu_fp32": e, "cpu_fp64": float(fx["checkpoint_elbo"][j]),
                                 "abs_diff": abs(e - float(fx["checkpoint_elbo"][j])), "tol": tol})
                    ok = ok and rows[-1]["abs_diff"] <= tol
            return {"iters": rows[-1]["iter"], "gpu_fp32": rows[-1]["gpu_fp32"], "cpu_fp64": rows[-1]["cpu_fp64"],
                    "abs_diff": rows[-1]["abs_diff"], "max_abs_diff": max(r["abs_diff"] for r in rows),
                    "within_tolerance": bool(ok), "checkpoints": rows,
                    "oracle": f"committed fixture tests/golden/long_{workload}.npz (fp64 oracle, same seeds)"}
    iters = 6
    o = make_oracle(w)
    for _ in range(iters):
        o.train_iter()
        g.train_iter()
    om = o.model.model
    e_cpu, sg = score_elbo(w["oracle_target"], om.log_weights, om.means, om.chol_cov)
    e_gpu, _ = score_elbo(w["oracle_target"], g.model.log_weights.numpy(), g.model.means.numpy(), g.model.chol_cov.numpy())
    return {"iters": iters, "gpu_fp32": e_gpu, "cpu_fp64": e_cpu, "abs_diff": abs(e_gpu - e_cpu),
            "within_tolerance": bool(abs(e_gpu - e_cpu) <= 3 * sg + 1e-2), "oracle": "fp64 oracle run side by side"}


def parse_profile(ctx):
    import ctypes
    buf = ctypes.create_string_buffer(1 << 16)
    ctx.check(ctx.lib.gmmvi_profile_report(ctx.handle, buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms, pairs = line.split()
        out[name] = (int(cnt), float(ms), float(pairs))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ns", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=80.0)
    args = ap.parse_args()
    n_gpus = args.gpus
    if "WORLD_SIZE" not in os.environ and n_gpus > 1:
        # no launcher: start the ranks ourselves, BEFORE anything here touches the GPU
        raise SystemExit(spawn_ranks(n_gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != n_gpus:
        n_gpus = world                                   # under a launcher its world size wins

    from gmmvi_amd.device import get_context
    ctx = get_context()
    w = build(args.workload, n_gpus, rank)
    algo = make_gmmvi(w, n_gpus, rank)
    exchange = getattr(algo, "exchange", None)

    def barrier():
        ctx.sync()
        if exchange is not None:
            exchange.barrier()
            ctx.sync()

    adaptive_k = w["cfg"]["num_component_adapter_type"] == "adaptive"
    pairs_done = [0.0]

    def timed(steps):
        barrier()
        pairs_done[0] = 0.0
        t_start = time.perf_counter()
        for _ in range(steps):
            if adaptive_k:               # K changes between iterations: (samples, components) pairs of this one
                pairs_done[0] += float(algo.model.num_components if n_gpus == 1 else algo.num_components) ** 2 * w["s"]
            algo.train_iter()
        if hasattr(algo, "flush"):
            algo.flush()             # sharded path: the last weight step rides with the next exchange; close it in the timed region
        barrier()
        dt = time.perf_counter() - t_start
        return exchange.max_scalar(dt) if exchange is not None else dt

    for _ in range(args.warmup):
        algo.train_iter()
    elapsed = timed(args.steps)
    ms_per_step = 1e3 * elapsed / args.steps
    pairs_per_step = pairs_done[0] / args.steps if adaptive_k else float(w["n_total"]) * w["k_total"]
    k_span = (None if not adaptive_k else
              f"K {w['k_total']} at the start, {int(algo.model.num_components if n_gpus == 1 else algo.num_components)} after "
              f"{args.warmup + args.steps} iterations")
    # the same figure at the two ends of a run: over the FIRST iterations of a fresh process (what a short --steps measures: the
    # sample database is still growing, caches are cold) and in the steady state behind them
    first_n = args.warmup + args.steps
    steady_steps = 200
    if first_n < 225:
        for _ in range(225 - first_n):   # bring the run to the same age before the steady-state window whatever --steps was
            algo.train_iter()
    steady_ms = 1e3 * timed(steady_steps) / steady_steps

    # ---- roofline leg: the same steps again with per-kernel HIP events on the compute stream ---------------------
    ctx.check(ctx.lib.gmmvi_profile_enable(ctx.handle, 1))
    prof_steps = min(args.steps, 50)
    for _ in range(prof_steps):
        algo.train_iter()
    prof = parse_profile(ctx)
    ctx.check(ctx.lib.gmmvi_profile_enable(ctx.handle, 0))
    kernels = {name: {"launches_per_step": c / prof_steps, "avg_us": 1e3 * ms / c, **({"pairs_per_launch": pr / c} if pr else {})}
               for name, (c, ms, pr) in prof.items()}
    k_local = w["k_total"] // n_gpus
    step_us = sum(v["avg_us"] * v["launches_per_step"] for v in kernels.values())
    for v in kernels.values():
        v["share_of_step"] = v["avg_us"] * v["launches_per_step"] / step_us
    dominant = max(kernels, key=lambda nme: kernels[nme]["share_of_step"])
    # every launch name stands for ONE kind of launch (the dual sweep, the target evaluation and the post-update sweep have
    # their own names); kernels that report the pairs they processed are priced on those pairs, the others on N x K
    def launch_flops(nme, fn=kernel_flops):
        if "pairs_per_launch" in kernels[nme]:
            return fn(nme, kernels[nme]["pairs_per_launch"], 1, w["d"])
        return fn(nme, w["n_total"], k_local, w["d"])
    priced = [nme for nme in kernels if kernel_flops(nme, 1, 1, 1) is not None]
    # roofline kernel: the TIME-dominant launch of the step (largest share of the step time) among the kernels that carry
    # algorithmic FLOPs -- normally dominant_kernel itself; a latency-chain kernel without a per-pair FLOP figure (the
    # KL-constrained update, one workgroup per component) cannot be priced and is then named in roofline.note
    roof_name = max(priced, key=lambda nme: kernels[nme]["share_of_step"])
    fl = launch_flops(roof_name)
    achieved = fl / (kernels[roof_name]["avg_us"] * 1e-6) / 1e12
    kernel_roofline = {}
    for nme in priced:
        pk = kernel_peak(nme)
        t = kernels[nme]["avg_us"] * 1e-6
        kernel_roofline[nme] = {"avg_us": kernels[nme]["avg_us"], "launches_per_step": kernels[nme]["launches_per_step"],
                                "share_of_step": kernels[nme]["share_of_step"],
                                "bound": kernel_bound(nme, w["d"]), "peak": pk,
                                "flops_alg_per_launch": launch_flops(nme), "frac": launch_flops(nme) / t / 1e12 / pk,
                                "frac_executed": launch_flops(nme, kernel_flops_executed) / t / 1e12 / pk,
                                # every kernel also against the plain f32 peak (the blocked_* kernels run on the bf16 matrix
                                # cores through split operands and are priced against that route's own ceiling above)
                                "frac_of_f32_peak": launch_flops(nme) / t / 1e12 / PEAK_FP32_TFLOPS}
    roof_peak = kernel_peak(roof_name)
    d, n_tot, k_tot = w["d"], w["n_total"], w["k_total"]
    f_alg_iter = float(n_tot) * k_tot * (8 * d * d + 12 * d)                   # SURVEY.md 8d (probes reported apart)
    b_alg_iter = 4.0 * (3 * n_tot * d + 3 * n_tot + 2 * k_tot * (d * d + d + 1))

    # HBM bytes per launch of the roofline kernel from the newest committed PMC summary (profiles/rNN_traffic.json; collected
    # with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled as MI355X_MICROARCH.md prescribes)
    traffic = None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                traffic = json.load(f).get(args.workload, {}).get(roof_name, {}).get("hbm_bytes")
        except (OSError, ValueError):
            traffic = None
        if traffic is not None:
            break
    result = {
        "metric": "samples_components_per_sec", "value": pairs_per_step / (elapsed / args.steps),
        "unit": "samples*components/s", "train_iter_per_sec": args.steps / elapsed,
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "ms_per_step_first": {"iterations": f"{args.warmup + 1}..{args.warmup + args.steps} of a fresh process", "ms": ms_per_step},
        "ms_per_step_steady": {"iterations": f"{max(first_n, 225) + 1}..{max(first_n, 225) + steady_steps}", "ms": steady_ms},
        "higher_is_better": True, "scaling": "strong" if args.workload in STRONG_WORKLOADS else "weak", "vs_baseline": None, "dtype": "f64" if roof_name == "more_gram" else "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {w['kind']} target D={d}, K={k_tot} components "
                               f"({k_local}/GPU), {w['s']} samples/component, N={n_tot} samples/iter, SAMTRON "
                               f"({w['cfg']['ng_estimator_type']}, {'adaptive' if w['cfg']['num_component_adapter_type'] == 'adaptive' else 'fixed'} K, reuse ratio "
                               f"{w['cfg']['sample_selector_config']['ratio_reused_samples_to_desired']:g}, KL trust regions, "
                               f"improvement-based stepsizes)",
                   "K": k_tot, "D": d, "N": n_tot, "parallelism": f"component-shard x{n_gpus}",
                   # which code path issued the iterations: the launches are the same on the first two
                   "path": ("single C call per iteration (gmmvi_train_iter_samtron)" if getattr(getattr(algo, "_fast_path", None), "eligible", lambda: False)()
                            else "four C calls per iteration with an all-gather between them (gmmvi_train_iter_sharded_phase)"
                            if getattr(algo, "_fast", None) is not None
                            else "module-by-module launches, components partitioned by id (gmmvi_amd/sharded_adaptive.py)"
                            if adaptive_k and n_gpus > 1
                            else "module-by-module plug-in calls"),
                   **({"adaptive": k_span + "; value = mean N K of the timed iterations / time"} if k_span else {})},
        "roofline": {"kernel": roof_name, "bound": kernel_bound(roof_name, d), "achieved": achieved, "peak": roof_peak,
                     "unit": "TFLOP/s", "frac": achieved / roof_peak, "traffic": traffic,
                     "avg_us": kernels[roof_name]["avg_us"], "flops_per_launch": fl,
                     "flops_executed_per_launch": launch_flops(roof_name, kernel_flops_executed),
                     "frac_of_f32_peak": achieved / PEAK_FP32_TFLOPS,
                     "selection": "the launch with the largest share of the step time among the FLOP-carrying kernels"
                                  + ("" if roof_name == dominant else f" (dominant_kernel {dominant} is a latency chain without a "
                                     "per-pair FLOP figure)") + "; kernel_roofline prices every FLOP-carrying kernel the same way",
                     "share_of_step": kernels[roof_name]["share_of_step"],
                     "note": "achieved = algorithmic FLOPs of the launch (pairs x per-pair figure of bench.kernel_flops, SURVEY.md "
                             "8(d)) / its mean HIP-event duration; peak = fp32 vector == f32 MFMA rate (blocked_* kernels on the split-operand "
                             "route: bf16 dense MFMA peak / 6 partial products, in f32-equivalent FLOPs). The Stein kernel "
                             "(moment form) executes fewer FLOPs than its algorithmic figure (kernel_roofline.stein_partial."
                             "frac_executed). Algorithmic HBM "
                             "bytes per iteration are tiny (see iter_roofline): the north star's >=50% HBM roofline is "
                             "unreachable on algorithmic bytes (SURVEY.md 8d)."},
        "kernel_roofline": kernel_roofline,
        "iter_roofline": {"flops_alg": f_alg_iter, "bytes_alg": b_alg_iter,
                          "frac_fp32_peak": f_alg_iter / (elapsed / args.steps) / (PEAK_FP32_TFLOPS * 1e12 * n_gpus),
                          "frac_hbm_peak": b_alg_iter / (elapsed / args.steps) / (PEAK_HBM_GBS * 1e9 * n_gpus)},
        "kernels": kernels, "dominant_kernel": dominant,
    }

    if any(nme.startswith("blocked_") for nme in kernels):
        result["arithmetic"] = ("f32 values throughout; blocked contractions (D > 50) " +
                                ("on v_mfma_f32_32x32x2_f32 (GMMVI_BLOCKED_F32=1)" if kernel_peak("blocked_forward") == PEAK_FP32_TFLOPS else
                                 "as 6 bf16 x bf16 partial products of 3-way split f32 operands with f32 accumulation (error of an f32 "
                                 "contraction: tests/test_hip_blocked.py); GMMVI_BLOCKED_F32=1 selects the f32 MFMA route"))
    # ---- CPU baseline + matched-ELBO check (rank 0, 1 GPU only) -------------------------------------------------
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline and w["d"] > 64:
        result["cpu_baseline"] = None
        result["cpu_baseline_note"] = ("not timed: one fp64 NumPy iteration of this workload materialises several [K,N,D] "
                                       "arrays (3 GB each) and takes minutes; parity at D = 300 is covered by "
                                       "tests/test_hip_blocked.py on smaller K, N")
    elif rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        t_cpu, cpu_iters, threads = time_cpu_baseline(w, args.cpu_seconds)
        result["cpu_baseline"] = {"value": n_tot * k_tot / t_cpu, "unit": "samples*components/s",
                                  "train_iter_per_sec": 1.0 / t_cpu, "cores": threads, "kind": "port", "dtype": "f32",
                                  "sample": f"3 warm-up + {cpu_iters} timed train_iter() of the same workload (fp32 "
                                            f"NumPy/SciPy restatement of the reference's algorithm, the reference's own "
                                            f"arithmetic), median iteration time"}
        result["matched_elbo"] = matched_elbo(args.workload, w)
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
