#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_all_23.log 2>&1
tail -6 gpurun_out/pytest_all_23.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_23.log 2>&1; tail -1 gpurun_out/bench_23.log | cut -c1-200
CB_GEMM_FUSED_REDUCE=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_23_nofuse.log 2>&1; tail -1 gpurun_out/bench_23_nofuse.log | cut -c1-200
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_23b.log 2>&1; tail -1 gpurun_out/bench_23b.log | cut -c1-200
python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "not sd14" 2>&1 | tail -3
python tools/bench_kernels.py --iters 10 > gpurun_out/kernels_bench_23.log 2>&1; cp gpurun_out/kernels_bench.json gpurun_out/kernels_bench_23.json
