#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_all_30.log 2>&1
tail -6 gpurun_out/pytest_all_30.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_30.log 2>&1; tail -1 gpurun_out/smoke_30.log
python bench.py > gpurun_out/bench_30.log 2>&1; tail -1 gpurun_out/bench_30.log | cut -c1-300
REPO=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof30 -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/rocprof_30.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/pmc30f -o t --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/pmc_30f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/pmc30w -o t --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/pmc_30w.log 2>&1
cd $REPO
python tools/pmc_traffic.py gpurun_out/pmc30f gpurun_out/pmc30w gpurun_out/pmc_traffic_30.json | head -4
find gpurun_out -name "*.csv" -size +8M -delete
