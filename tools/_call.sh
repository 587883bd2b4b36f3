set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16_ops.py -q -p no:cacheprovider --no-header -k "groupnorm or glue or mse or loss or norm" ) > gpurun_out/c5_ops.log 2>&1
tail -4 gpurun_out/c5_ops.log
( timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_bf16_models.py -q -p no:cacheprovider --no-header -k "mini and not 40_steps" ) > gpurun_out/c5_models.log 2>&1
tail -4 gpurun_out/c5_models.log
timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/c5_bench.log 2>&1
grep '^{' gpurun_out/c5_bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["kernel_launches_per_step"], d["pipelined"]["ms_per_step"], "| config3", d["config3"]["ms_per_step"], d["config3"]["samples_per_s"], d["config3"]["pipelined"]["ms_per_step"]); print([ (e["kernel"], e["launches"], e["ms_events"]) for e in d["roofline"]["hbm_bound"]["events"][:8]]); print([ (e["kernel"], e["launches"], e["ms_events"]) for e in d["config3"]["roofline"]["hbm_bound"]["events"][:8]])'
