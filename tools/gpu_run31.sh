#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider -k "prefetch or train_step_mini" 2>&1 | tail -6 > gpurun_out/pytest_31.log; tail -3 gpurun_out/pytest_31.log
python bench.py --no-cpu-baseline > gpurun_out/bench_31.log 2>&1; tail -1 gpurun_out/bench_31.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['pipelined'])"
tail -3 gpurun_out/bench_31.log | cut -c1-300
