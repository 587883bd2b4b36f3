#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "not sd14" 2>&1 | tail -4 > gpurun_out/pytest_21.log; tail -2 gpurun_out/pytest_21.log
CB_GEMM_OCC4=1 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "not sd14" 2>&1 | tail -4 > gpurun_out/pytest_21_occ4.log; tail -2 gpurun_out/pytest_21_occ4.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_21.log 2>&1; tail -1 gpurun_out/bench_21.log | cut -c1-200
CB_GEMM_OCC4=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_21_occ4.log 2>&1; tail -1 gpurun_out/bench_21_occ4.log | cut -c1-200
python tools/bench_kernels.py --iters 10 > gpurun_out/kernels_bench_21.log 2>&1; cp gpurun_out/kernels_bench.json gpurun_out/kernels_bench_21.json
CB_GEMM_OCC4=1 python tools/bench_kernels.py --iters 10 > gpurun_out/kernels_bench_21_occ4.log 2>&1; cp gpurun_out/kernels_bench.json gpurun_out/kernels_bench_21_occ4.json
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_21b.log 2>&1; tail -1 gpurun_out/bench_21b.log | cut -c1-200
