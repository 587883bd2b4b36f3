set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/_b16.py bf16 > gpurun_out/c9_b16_bf16.log 2>&1; tail -2 gpurun_out/c9_b16_bf16.log | cut -c1-300
timeout 300 python tools/_b16.py f32 > gpurun_out/c9_b16_f32.log 2>&1; tail -2 gpurun_out/c9_b16_f32.log | cut -c1-300
for pr in -1 0; do
  CB_AUX_PRIORITY=$pr timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/c9_bench_aux_$pr.log 2>&1
  echo "AUX_PRIORITY=$pr: $(grep '^{' gpurun_out/c9_bench_aux_$pr.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["phase_ms"], d["pipelined"]["ms_per_step"], "| c3", d["config3"]["ms_per_step"], d["config3"]["phase_ms"], d["config3"]["pipelined"]["ms_per_step"])')"
done
for pr in -1 0; do
  CB_AUX_PRIORITY=$pr timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/c9_bench_aux_${pr}_b.log 2>&1
  echo "AUX_PRIORITY=$pr (2nd): $(grep '^{' gpurun_out/c9_bench_aux_${pr}_b.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["pipelined"]["ms_per_step"], "| c3", d["config3"]["ms_per_step"], d["config3"]["pipelined"]["ms_per_step"])')"
done
