#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider -k "groupnorm or train_step_mini or unet_mini" 2>&1 | tail -3
python bench.py > gpurun_out/bench_38.log 2>&1; tail -1 gpurun_out/bench_38.log | cut -c1-200
