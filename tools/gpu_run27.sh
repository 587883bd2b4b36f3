#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 0.25e6 0.5e6; do
  CB_SPLITK_COST=$c python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_27_$c.log 2>&1; echo "cost $c: $(tail -1 gpurun_out/bench_27_$c.log | cut -c60-140)"
done
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > gpurun_out/pytest_all_27.log; tail -2 gpurun_out/pytest_all_27.log
python bench.py > gpurun_out/bench_27.log 2>&1; tail -1 gpurun_out/bench_27.log | cut -c1-300
REPO=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof27 -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/rocprof_27.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/pmc27f -o t --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/pmc_27f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/pmc27w -o t --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/pmc_27w.log 2>&1
cd $REPO
python tools/pmc_traffic.py gpurun_out/pmc27f gpurun_out/pmc27w gpurun_out/pmc_traffic_27.json | head -5
find gpurun_out -name "*.csv" -size +8M -delete
