"""profiles/rNN_traffic.json from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summaries that tools/prof_bench.sh leaves
(tools/pmc_summary.py text format): per bench kernel name the HBM bytes of one launch,
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE counts half of a coalesced 16-byte-per-lane stream on gfx950
(MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact for such streams; narrower accesses are not separately calibrated.
usage: python tools/make_traffic_json.py out.json workload=fetch.txt,write.txt [workload=...]"""
import json, re, sys

NAMES = [  # (substring of the kernel instance name, bench.py kernel name); first match wins
    ("stein_moment_kernel", "stein_partial"), ("stein_finalize_kernel", "stein_finalize"),
    # density sweeps by their bench.py launch names (the single-call iteration: dual sweep = model gradient instance, post-update
    # sweep = model log-value instance, target = the other family / the per-wave matrix-core instance at D = 50)
    ("mixture_eval_kernel<20, 1, true", "sweep_target"), ("mixture_eval_kernel<10, 1, true", "sweep_target"),
    ("mixture_eval_mfma_ws_kernel<50, 0, true", "sweep_dual"), ("mixture_eval_mfma_kernel<50, 0, true", "sweep_target"), ("mixture_eval_mfma_kernel<50, 0, false", "sweep_post"),
    ("mixture_eval_pk_kernel<20, 0, true", "sweep_dual"), ("mixture_eval_pk_kernel<20, 0, false", "sweep_post"),
    ("mixture_eval_pk_kernel<10, 0, true", "sweep_dual"), ("mixture_eval_pk_kernel<10, 0, false", "sweep_post"),
    ("elr_riders_kernel", "expected_log_ratios"),
    ("mixture_eval_kernel<20, 0, true", "sweep_dual"), ("mixture_eval_kernel<20, 0, false", "sweep_post"),
    ("mixture_eval_kernel<10, 0, true", "sweep_dual"), ("mixture_eval_kernel<10, 0, false", "sweep_post"),
    ("update_kl_fast_kernel", "update_kl"), ("combine_partials_kernel", "mixture_combine"),
    ("sample_components_kernel", "sample_components"), ("elr_kernel", "expected_log_ratios"),
    ("update_weights_kernel", "update_weights"),
    ("bgemm_ws_kernel<10, 1, 4>", "blocked_stein_accumulate"), ("bgemm_ws_kernel<10, 0, 1>", "blocked_forward"),
    ("bgemm_ws_kernel<10, 0, 2>", "blocked_grad"),
    ("bgemm_kernel<5, 1, 1, 4>", "blocked_stein_accumulate"), ("bgemm_kernel<5, 0, 0, 1>", "blocked_forward"),
    ("bgemm_kernel<5, 0, 1, 2>", "blocked_grad"), ("blk_tridiag", "blocked_tridiag"),
]


def read(path, counter):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
        else:
            m = re.search(counter + r"=([0-9.e+]+)", line)
            if m and name:
                out[name] = float(m.group(1))
    return out


def main():
    result = {"_comment": __doc__.split("usage:")[0].strip()}
    for arg in sys.argv[2:]:
        wl, files = arg.split("=")
        fpath, wpath = files.split(",")
        fetch, write = read(fpath, "FETCH_SIZE"), read(wpath, "WRITE_SIZE")
        entry = {}
        for inst in sorted(set(fetch) | set(write)):
            for sub, bench_name in NAMES:
                if sub in inst and bench_name not in entry:
                    f, w = fetch.get(inst, 0.0), write.get(inst, 0.0)
                    entry[bench_name] = {"kernel_instance": inst, "fetch_size_kb": f, "write_size_kb": w,
                                         "hbm_bytes": (2 * f + w) * 1024}
                    break
        result[wl] = entry
    json.dump(result, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
