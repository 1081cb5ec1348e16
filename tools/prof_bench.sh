# usage (on the GPU box): bash tools/prof_bench.sh <outdir> [workload] -- kernel trace + two SQ counter passes + FETCH/WRITE passes
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/$1; W=${2:-ns}
mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$W -- python3 bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline > $O/kt_$W.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/pmc1_$W -- python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > $O/pmc1_$W.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc2_$W -- python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > $O/pmc2_$W.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$W -- python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > $O/pmc3_$W.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$W -- python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline > $O/pmc4_$W.log 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
for p in pmc1_$W pmc2_$W pmc_fetch_$W pmc_write_$W; do python3 tools/pmc_summary.py $O/$p > $O/$p.txt; find $O/$p -name "*counter_collection.csv" -delete; done
find $O/kt_$W -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$W.csv
