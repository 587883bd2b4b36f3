#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/pytest_all_40.log 2>&1
tail -6 gpurun_out/pytest_all_40.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
