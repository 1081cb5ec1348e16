# usage (on the GPU box): bash tools/collect_round.sh <outdir> -- the bench lines of every workload (with CPU baseline and matched
# ELBO where bench.py runs them), the C5 shard also on the f32 matrix-core route
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O; cd $GRAFT_REPO_ROOT
for w in ns c2 c3 c4 d32 d40 d63 ns_reuse2 c5; do
    timeout -k 10 400 python3 bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err || echo "FAILED $w" >> $O/failed.txt
    echo "$w done"
done
GMMVI_BLOCKED_F32=1 timeout -k 10 300 python3 bench.py --workload c5 --no-cpu-baseline > $O/bench_c5_f32route.json 2> $O/bench_c5_f32route.err || echo "FAILED c5 f32" >> $O/failed.txt
echo "all done"
