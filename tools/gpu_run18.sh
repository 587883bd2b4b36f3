#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "not sd14" 2>&1 | tail -15 > gpurun_out/pytest_18.log
tail -4 gpurun_out/pytest_18.log
python tools/run_config5.py --identities 2 --train-steps 10 --n-samples 2 --ddim-steps 5 > gpurun_out/config5_small_18.log 2>&1; tail -2 gpurun_out/config5_small_18.log | cut -c1-600
python tools/run_config5.py --identities 1 --train-steps 800 > gpurun_out/config5_18.log 2>&1; tail -1 gpurun_out/config5_18.log | cut -c1-600
