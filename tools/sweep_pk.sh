#!/bin/bash
# Developer tool: geometry sweep of the packed density kernel at the north-star shape (run on the GPU box).
out=${1:-gpurun_out/sweep_pk.txt}
: > $out
GMMVI_ME_PK=0 python tools/bench_sweep.py >> $out 2>&1
for ky in 2 3 4 5 6 7 8 10 13; do
  GMMVI_ME_PK=1 GMMVI_ME_PK_KY=$ky python tools/bench_sweep.py >> $out 2>&1
done
for nw in 4 5 6 7; do
  GMMVI_ME_PK=1 GMMVI_ME_PK_NW=$nw python tools/bench_sweep.py >> $out 2>&1
done
