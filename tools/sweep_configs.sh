# usage (on the GPU box): bash tools/sweep_configs.sh <outfile> -- times tools/bench_sweep.py under a list of sweep geometries
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $(dirname $O); : > $O
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout -k 10 120 python3 tools/bench_sweep.py >> $O 2>&1 || echo "FAILED: $*" >> $O; }
run GMMVI_X=0
run GMMVI_ME_KY=5 GMMVI_ME_NW=5
run GMMVI_ME_KY=6 GMMVI_ME_NW=4
run GMMVI_ME_KY=4 GMMVI_ME_NW=5
run GMMVI_ME_KY=3 GMMVI_ME_NW=12
run GMMVI_ME_KY=6 GMMVI_ME_NW=6
run GMMVI_ME_KY=13 GMMVI_ME_NW=2
run GMMVI_ME_KY=10 GMMVI_ME_NW=2
run GMMVI_ME_KY=10 GMMVI_ME_NW=3
cat $O
