#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/pytest_ops_11.log
tail -4 gpurun_out/pytest_ops_11.log
python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "mini" 2>&1 | tail -25 > gpurun_out/pytest_models_11.log
tail -4 gpurun_out/pytest_models_11.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_11.log 2>&1
tail -1 gpurun_out/bench_11.log | cut -c1-300
python tools/bench_kernels.py --iters 10 > gpurun_out/kernels_bench_11.log 2>&1
cp gpurun_out/kernels_bench.json gpurun_out/kernels_bench_11.json
REPO=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof11 -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/rocprof_11.log 2>&1
cd $REPO; ls gpurun_out/prof11
find gpurun_out -name "*.csv" -size +8M -delete
