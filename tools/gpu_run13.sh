#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "sd14" --durations=5 ) > gpurun_out/pytest_sd14_13.log 2>&1
tail -12 gpurun_out/pytest_sd14_13.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_13.log 2>&1; tail -2 gpurun_out/smoke_13.log
