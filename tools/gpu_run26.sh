#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 1e6 4e6 8e6 16e6 1e6; do
  CB_SPLITK_COST=$c python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_26_$c.log 2>&1; echo "cost $c: $(tail -1 gpurun_out/bench_26_$c.log | cut -c60-140)"
done
