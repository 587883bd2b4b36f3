#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider -k "head or image_mode or glue" 2>&1 | tail -4 > gpurun_out/pytest_28.log; tail -2 gpurun_out/pytest_28.log
for c in 1e6 0.25e6 0.1e6 0.03e6 0.25e6 1e6; do
  CB_SPLITK_COST=$c python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_28_$c.log 2>&1; echo "cost $c: $(tail -1 gpurun_out/bench_28_$c.log | cut -c60-140)"
done
