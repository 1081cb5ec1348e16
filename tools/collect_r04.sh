#!/bin/bash
# Developer tool: the bench lines of every workload for profiles/r04_* (run on the GPU box).
O=gpurun_out/$1; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_ns_driver_cmd.json 2> $O/err.txt
python3 bench.py > $O/bench_ns.json 2>> $O/err.txt
for wl in c2 c3 c4 d32 d40 ns_reuse2 c4_adaptive; do python3 bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2>> $O/err.txt; done
python3 bench.py --workload c5 --steps 20 --warmup 3 > $O/bench_c5.json 2>> $O/err.txt
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("bench_")[-1][:-5], round(d["ms_per_step"] * 1e3, 1), "us; steady", round(d["ms_per_step_steady"]["ms"] * 1e3, 1), d["roofline"]["kernel"], d["roofline"]["bound"], round(d["roofline"]["frac"], 3), (d.get("matched_elbo") or {}).get("within_tolerance"))
    except Exception as e:
        print(f, "ERR", e)
PY
