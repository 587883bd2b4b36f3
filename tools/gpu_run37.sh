#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_all_37.log 2>&1
tail -6 gpurun_out/pytest_all_37.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_37.log 2>&1; tail -1 gpurun_out/smoke_36.log
python bench.py > gpurun_out/bench_37.log 2>&1; tail -1 gpurun_out/bench_37.log | cut -c1-200
