#!/bin/bash
# Developer tool: the three big blocked contractions of the C5 shard with parts of bgemm_ws_tile switched off (experiment build:
# tools/build_bg_stamps.sh; GMMVI_BG_DEBUG bits there): timing only, the results are wrong.
export GMMVI_HIP_LIB=gmmvi_amd/libgmmvi_hip_bgstamps.so
for dbg in 0 1 2 4 8 3 6 12 15 256; do
  GMMVI_BG_DEBUG=$dbg python3 bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('debug $dbg:', {n:round(k[n]['avg_us']) for n in ('blocked_forward','blocked_grad','blocked_stein_accumulate') if n in k})"
done
