#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_all_32.log 2>&1
tail -8 gpurun_out/pytest_all_32.log
