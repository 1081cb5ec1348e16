# usage (GPU box): bash tools/pmc_c5.sh <outdir> [workload] -- two SQ counter passes, per-kernel SUMS (ratios are then size-weighted)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/$1; W=${2:-c5}
mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/pmc1_$W -- python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc1_$W.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/pmc2_$W -- python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc2_$W.log 2>&1
for p in pmc1_$W pmc2_$W; do python3 tools/pmc_summary.py --sum $O/$p > $O/$p.txt; find $O/$p -name "*counter_collection.csv" -delete; done
