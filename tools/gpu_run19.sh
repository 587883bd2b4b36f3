#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_all_19.log 2>&1
tail -8 gpurun_out/pytest_all_19.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_19.log 2>&1
tail -1 gpurun_out/bench_19.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_19.log 2>&1; tail -1 gpurun_out/smoke_19.log
