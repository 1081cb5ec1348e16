#!/bin/bash
# Developer tool: north-star iteration (bench.py, 200 steps) under a list of environment settings; prints ms per step and the
# per-kernel HIP-event times.  usage: tools/iter_cfgs.sh out.txt "A=1 B=2" "A=3" ...
out=$1; shift
: > $out
for cfg in "$@"; do
  env $cfg python bench.py --steps 200 --warmup 20 --no-cpu-baseline ${WORKLOAD:+--workload $WORKLOAD} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$cfg]', round(d['ms_per_step']*1e3,1), 'us', {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})" >> $out
done
cat $out
