#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/exp_overlap.py > gpurun_out/exp_overlap_20.log 2>&1; tail -2 gpurun_out/exp_overlap_20.log
python -m pytest tests -m gpu -q -p no:cacheprovider -k "glue or txt2img or sampler or extract" 2>&1 | tail -5 > gpurun_out/pytest_20.log; tail -3 gpurun_out/pytest_20.log
