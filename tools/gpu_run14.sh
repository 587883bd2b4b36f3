#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/pytest_ops_14.log
tail -4 gpurun_out/pytest_ops_14.log
python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/pytest_models_14.log
tail -4 gpurun_out/pytest_models_14.log
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_14.log 2>&1
tail -1 gpurun_out/bench_14.log | cut -c1-300
python tools/bench_kernels.py --iters 10 > gpurun_out/kernels_bench_14.log 2>&1
cp gpurun_out/kernels_bench.json gpurun_out/kernels_bench_14.json
REPO=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof14 -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/rocprof_14.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/pmc14f -o t --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/pmc_14f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/pmc14w -o t --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $REPO/gpurun_out/pmc_14w.log 2>&1
cd $REPO; ls gpurun_out/prof14 gpurun_out/pmc14f
python tools/pmc_traffic.py gpurun_out/pmc14f gpurun_out/pmc14w gpurun_out/pmc_traffic_14.json
find gpurun_out -name "*.csv" -size +8M -delete
