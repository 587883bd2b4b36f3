#!/bin/bash
# bench + per-kernel microbench + rocprofv3 kernel trace (one gpurun call)
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k glue 2>&1 | tail -5 > gpurun_out/pytest_glue.log
( time python bench.py --steps 10 --warmup 2 --cpu-steps 2 ) > gpurun_out/bench_1.log 2>&1
tail -3 gpurun_out/bench_1.log
python tools/bench_kernels.py --iters 10 > gpurun_out/kernels_bench.log 2>&1
tail -5 gpurun_out/kernels_bench.log
REPO=$(pwd)
cd /tmp && rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/rocprof_1.log 2>&1
cd $REPO
ls -R gpurun_out/prof | head -30
find gpurun_out/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" > gpurun_out/kernel_stats_head.csv
# keep the merged output small: drop the raw trace
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
