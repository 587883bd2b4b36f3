set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python tests/horizon_cli.py f32 40 planA > gpurun_out/c3_hzA.log 2>&1; tail -1 gpurun_out/c3_hzA.log | cut -c1-300
CB_PLAN_TABLE=0 timeout 400 python tests/horizon_cli.py f32 40 planB > gpurun_out/c3_hzB.log 2>&1; tail -1 gpurun_out/c3_hzB.log | cut -c1-300
for pr in 1 0 -1; do
  CB_PF_PRIORITY=$pr timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/c3_bench_prio_$pr.log 2>&1
  echo "PF_PRIORITY=$pr: $(grep '^{' gpurun_out/c3_bench_prio_$pr.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["pipelined"], d["config3"]["ms_per_step"], d["config3"]["pipelined"])')"
done
