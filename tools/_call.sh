set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --no-header ) > gpurun_out/c10_pytest.log 2>&1
tail -6 gpurun_out/c10_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c10_smoke.log 2>&1; tail -1 gpurun_out/c10_smoke.log
bash tools/gpurun.sh trace r03c > gpurun_out/c10_trace.log 2>&1; tail -2 gpurun_out/c10_trace.log
bash tools/gpurun.sh trace3 r03c > gpurun_out/c10_trace3.log 2>&1; tail -2 gpurun_out/c10_trace3.log
bash tools/gpurun.sh pmc r03c > gpurun_out/c10_pmc.log 2>&1; tail -2 gpurun_out/c10_pmc.log
bash tools/gpurun.sh pmc3 r03c > gpurun_out/c10_pmc3.log 2>&1; tail -2 gpurun_out/c10_pmc3.log
mkdir -p profiles_tmp && cp gpurun_out/r03c_* profiles/ && timeout 600 python bench.py > gpurun_out/c10_bench.log 2>&1; grep '^{' gpurun_out/c10_bench.log | tail -1 > gpurun_out/BENCH_r03_run1.json; cut -c1-200 gpurun_out/BENCH_r03_run1.json
ls -la gpurun_out/r03c*
