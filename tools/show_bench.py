"""Developer tool: one line per bench.py JSON file (ms per step and the per-kernel HIP-event averages)."""
import json, sys
for path in sys.argv[1:]:
    d = json.load(open(path))
    ks = ", ".join(f"{k} {v['avg_us']:.1f}" for k, v in d["kernels"].items())
    print(f"{path}: {d['ms_per_step'] * 1e3:.1f} us/step, {d['train_iter_per_sec']:.0f} it/s; {ks}")
