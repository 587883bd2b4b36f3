set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/cb_bench.log 2>&1; grep '^{' gpurun_out/cb_bench.log | tail -1 > gpurun_out/BENCH_r03_runN.json; cut -c1-160 gpurun_out/BENCH_r03_runN.json
