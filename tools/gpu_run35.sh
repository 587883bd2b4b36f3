#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
CB_CONV_KORD=1 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "not sd14_vs" 2>&1 | tail -5 > gpurun_out/pytest_35_kord.log; tail -3 gpurun_out/pytest_35_kord.log
run() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined > gpurun_out/bench_35_tmp.log 2>&1; echo "$* : $(tail -1 gpurun_out/bench_35_tmp.log | cut -c72-110)"; }
run CB_CONV_KORD=0
run CB_CONV_KORD=1
run CB_CONV_KORD=0
run CB_CONV_KORD=1
REPO=$(pwd)
cd /tmp
CB_CONV_KORD=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/gpurun_out/pmc35f -o t --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined > $REPO/gpurun_out/pmc_35f.log 2>&1
CB_CONV_KORD=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $REPO/gpurun_out/pmc35w -o t --output-format csv -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pipelined > $REPO/gpurun_out/pmc_35w.log 2>&1
cd $REPO
python tools/pmc_traffic.py gpurun_out/pmc35f gpurun_out/pmc35w gpurun_out/pmc_traffic_35.json | head -4
find gpurun_out -name "*.csv" -size +8M -delete
