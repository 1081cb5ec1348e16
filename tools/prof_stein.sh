# usage (on the GPU box): bash tools/prof_stein.sh <outdir> [workload]
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/$1; W=${2:-ns}
mkdir -p $O; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python3 tools/stein_probe.py $W 100 > $O/probe_$W.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$W -- python3 tools/stein_probe.py $W 50 > $O/kt_$W.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/pmc1_$W -- python3 tools/stein_probe.py $W 10 > $O/pmc1_$W.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/pmc2_$W -- python3 tools/stein_probe.py $W 10 > $O/pmc2_$W.log 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
for p in pmc1_$W pmc2_$W; do python3 tools/pmc_summary.py $O/$p > $O/$p.txt; find $O/$p -name "*counter_collection.csv" -delete; done
cat $O/probe_$W.txt; find $O/kt_$W -name "*kernel_stats.csv" | head -1 | xargs -I{} head -8 {}
grep -A1 "stein" $O/pmc1_$W.txt $O/pmc2_$W.txt
