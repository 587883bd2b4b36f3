#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 280 python -m celebbasis_amd.main --base tests/golden/aigc_id_params.json -t --synthetic -n cli_smoke --max_steps 6 --pipeline -l gpurun_out/logs data.params.batch_size=1 > gpurun_out/main_cli_41.log 2>&1
tail -4 gpurun_out/main_cli_41.log; ls -la gpurun_out/logs/cli_smoke/checkpoints 2>/dev/null
