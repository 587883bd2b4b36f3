#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_29_tmp.log 2>&1; echo "$* : $(tail -1 gpurun_out/bench_29_tmp.log | cut -c72-110)"; }
run A=base
run CB_PLAN_TRAFFIC=0.2
run CB_PLAN_TRAFFIC=0.8
run CB_PLAN_BLOCK_OVH=2
run CB_PLAN_BLOCK_OVH=5
run CB_PLAN_MINK=2
run CB_PLAN_MINK=8
run CB_PLAN_EFF=1.0,0.8,0.8,0.7
run CB_PLAN_EFF=1.0,0.8,0.8,0.45
run CB_PLAN_EFF=1.0,0.9,0.9,0.58
run CB_PLAN_EFF=1.0,0.7,0.7,0.58
run A=base
