#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -p no:cacheprovider -x ) > gpurun_out/pytest_all_16.log 2>&1
tail -6 gpurun_out/pytest_all_16.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_16.log 2>&1
tail -1 gpurun_out/bench_16.log | cut -c1-300
python tools/bench_kernels.py --iters 10 > gpurun_out/kernels_bench_16.log 2>&1
cp gpurun_out/kernels_bench.json gpurun_out/kernels_bench_16.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_16.log 2>&1; tail -1 gpurun_out/smoke_16.log
