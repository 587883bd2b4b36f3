set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_bf16_ops.py -q -x -p no:cacheprovider --no-header -k "attention" ) > gpurun_out/c2_attn.log 2>&1
tail -3 gpurun_out/c2_attn.log
for v in "" "2,2,2" "1,1,1" "4,2,2" "4,4,1"; do
  CB_F2_QB="$v" timeout 200 python tools/bench_flash.py --dtype bf16 --batch 4 --only "self 64" > gpurun_out/c2_flash_qb_$v.log 2>&1
  echo "QB=$v: $(grep shape gpurun_out/c2_flash_qb_$v.log | cut -c1-200)"
done
( timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_ops.py -q -p no:cacheprovider --no-header -k "mini_vs_oracle or b1_vs_oracle or 10_steps or feeder" ) > gpurun_out/c2_pytest.log 2>&1
tail -6 gpurun_out/c2_pytest.log
timeout 400 python tests/horizon_cli.py bf16 20 base > gpurun_out/c2_hz_base.log 2>&1; tail -1 gpurun_out/c2_hz_base.log | cut -c1-600
CB_BF16_KV_F32=1 timeout 400 python tests/horizon_cli.py bf16 20 kv32 > gpurun_out/c2_hz_kv32.log 2>&1; tail -1 gpurun_out/c2_hz_kv32.log | cut -c1-600
for st in 2 3 4; do
  CB_G64_STAGES=$st timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --no-config3 > gpurun_out/c2_bench_g64_$st.log 2>&1
  echo "G64_STAGES=$st fp32: $(grep '^{' gpurun_out/c2_bench_g64_$st.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["kernel_launches_per_step"])')"
done
for st in 2 3; do
  CB_G64_STAGES=$st timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --trace-leg config3 > gpurun_out/c2_bench3_g64_$st.log 2>&1
  echo "G64_STAGES=$st config3: $(grep '^{' gpurun_out/c2_bench3_g64_$st.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); c=d["config3"]; print(c["ms_per_step"], c["samples_per_s"], c["phase_ms"])')"
done
