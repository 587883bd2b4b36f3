#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_all_22.log 2>&1
tail -8 gpurun_out/pytest_all_22.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_22.log 2>&1; tail -1 gpurun_out/bench_22.log | cut -c1-200
CB_GEMM_OCC4=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_22_occ0.log 2>&1; tail -1 gpurun_out/bench_22_occ0.log | cut -c1-200
python tools/bench_kernels.py --iters 10 > gpurun_out/kernels_bench_22.log 2>&1; cp gpurun_out/kernels_bench.json gpurun_out/kernels_bench_22.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_22b.log 2>&1; tail -1 gpurun_out/bench_22b.log | cut -c1-200
