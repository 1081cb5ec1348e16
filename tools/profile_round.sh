set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s6
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
python bench.py > $O/bench_ns.json 2> $O/bench_ns.err
python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench_ns.err
python bench.py --workload c3 --no-cpu-baseline > $O/bench_c3.json 2>> $O/bench_ns.err
python bench.py --workload c4 --no-cpu-baseline > $O/bench_c4.json 2>> $O/bench_ns.err
python bench.py --workload c5 --steps 20 --warmup 3 > $O/bench_c5.json 2>> $O/bench_ns.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ns -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/kt_ns.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c5 -- python3 bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $O/kt_c5.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_ns -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_ns -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_c5 -- python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_c5 -- python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc4.log 2>&1
# keep only the summaries (the merge back is capped at 64 MiB)
find $O -name "*kernel_trace.csv" -size +8M -delete
for d in pmc_fetch_ns pmc_write_ns pmc_fetch_c5 pmc_write_c5; do python tools/pmc_summary.py $O/$d > $O/$d.txt; find $O/$d -name "*counter_collection.csv" -size +20M -delete; done
tail -3 $O/pytest_gpu.log
cut -c1-200 $O/bench_ns.json
