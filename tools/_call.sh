set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --no-header ) > gpurun_out/c6_pytest.log 2>&1
tail -6 gpurun_out/c6_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c6_smoke.log 2>&1; tail -1 gpurun_out/c6_smoke.log
timeout 600 python bench.py > gpurun_out/c6_bench.log 2>&1; grep '^{' gpurun_out/c6_bench.log | tail -1 > gpurun_out/BENCH_r03_run1.json; cut -c1-200 gpurun_out/BENCH_r03_run1.json
bash tools/gpurun.sh trace r03b > gpurun_out/c6_trace.log 2>&1; tail -2 gpurun_out/c6_trace.log
bash tools/gpurun.sh trace3 r03b > gpurun_out/c6_trace3.log 2>&1; tail -2 gpurun_out/c6_trace3.log
bash tools/gpurun.sh pmc r03b > gpurun_out/c6_pmc.log 2>&1; tail -2 gpurun_out/c6_pmc.log
bash tools/gpurun.sh pmc3 r03b > gpurun_out/c6_pmc3.log 2>&1; tail -2 gpurun_out/c6_pmc3.log
timeout 300 python -m celebbasis_amd.main --base tests/golden/aigc_id_params.json -t --synthetic --max_steps 220 --report gpurun_out/MAIN_r03_device.json data.params.batch_size=1 -l gpurun_out/logs_a > gpurun_out/c6_main_dev.log 2>&1; tail -1 gpurun_out/c6_main_dev.log | cut -c1-300
timeout 300 python -m celebbasis_amd.main --base tests/golden/aigc_id_params.json -t --synthetic --max_steps 220 --pipeline --report gpurun_out/MAIN_r03_device_pipeline.json data.params.batch_size=1 -l gpurun_out/logs_c > gpurun_out/c6_main_pipe.log 2>&1; tail -1 gpurun_out/c6_main_pipe.log | cut -c1-300
ls -la gpurun_out/r03b*
