#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "decoder or ddim or sampler" 2>&1 | tail -40 > gpurun_out/pytest_sampler_15.log
tail -12 gpurun_out/pytest_sampler_15.log
python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/pytest_ops_15.log
tail -2 gpurun_out/pytest_ops_15.log
python tools/bench_sampler.py --n 1 > gpurun_out/sampler_15_n1.log 2>&1; tail -1 gpurun_out/sampler_15_n1.log
python tools/bench_sampler.py --n 4 --reps 1 > gpurun_out/sampler_15_n4.log 2>&1; tail -1 gpurun_out/sampler_15_n4.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_15.log 2>&1
tail -1 gpurun_out/bench_15.log | cut -c1-400
