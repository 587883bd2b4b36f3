#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_all_25.log 2>&1
tail -6 gpurun_out/pytest_all_25.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_25.log 2>&1; tail -1 gpurun_out/bench_25.log | cut -c1-200
CB_GN_COMPACT=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_25_off.log 2>&1; tail -1 gpurun_out/bench_25_off.log | cut -c1-200
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_25b.log 2>&1; tail -1 gpurun_out/bench_25b.log | cut -c1-200
