#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined > gpurun_out/bench_34_tmp.log 2>&1; echo "$* : $(tail -1 gpurun_out/bench_34_tmp.log | cut -c72-110)"; }
run A=base
run CB_GN_BLOCKS=256
run CB_GN_BLOCKS=512
run CB_GN_BLOCKS=2048
run CB_GN_BLOCKS=4096
run A=base
