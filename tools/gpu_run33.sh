#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_all_33.log 2>&1
tail -8 gpurun_out/pytest_all_33.log
python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "prefetch or host_training" 2>&1 | tail -3
