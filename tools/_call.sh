set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --no-header ) > gpurun_out/c4_pytest.log 2>&1
tail -8 gpurun_out/c4_pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c4_smoke.log 2>&1; tail -2 gpurun_out/c4_smoke.log
timeout 600 python bench.py > gpurun_out/c4_bench.log 2>&1; grep '^{' gpurun_out/c4_bench.log | tail -1 > gpurun_out/BENCH_r03_run1.json; cut -c1-300 gpurun_out/BENCH_r03_run1.json
bash tools/gpurun.sh trace r03a > gpurun_out/c4_trace.log 2>&1; tail -3 gpurun_out/c4_trace.log
bash tools/gpurun.sh trace3 r03a > gpurun_out/c4_trace3.log 2>&1; tail -3 gpurun_out/c4_trace3.log
bash tools/gpurun.sh pmc r03a > gpurun_out/c4_pmc.log 2>&1; tail -3 gpurun_out/c4_pmc.log
bash tools/gpurun.sh pmc3 r03a > gpurun_out/c4_pmc3.log 2>&1; tail -3 gpurun_out/c4_pmc3.log
ls -la gpurun_out/r03a* 
