#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/bench_sampler.py --n 8 --steps 10 --reps 1 > gpurun_out/sampler_39_n8.log 2>&1; tail -1 gpurun_out/sampler_39_n8.log
python -m pytest tests -m gpu -q -p no:cacheprovider -k "sampler or decoder or txt2img" 2>&1 | tail -3
