# usage (GPU box): bash tools/pmc_l2.sh <outdir> [workload] -- L2 (TCC) hit / miss / request sums per kernel
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/$1; W=${2:-c5}
mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/l2_$W -- python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline > $O/l2_$W.log 2>&1
python3 tools/pmc_summary.py --sum $O/l2_$W > $O/l2_$W.txt; find $O/l2_$W -name "*counter_collection.csv" -delete
