set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/c8_bench.log 2>&1; grep '^{' gpurun_out/c8_bench.log | tail -1 > gpurun_out/BENCH_r03_runN.json; cut -c1-160 gpurun_out/BENCH_r03_runN.json
timeout 300 python tools/bench_sampler.py --n 8 > gpurun_out/c8_sampler_f32.log 2>&1; grep '^{' gpurun_out/c8_sampler_f32.log | tail -1 > gpurun_out/SAMPLER_r03_n8.json; cut -c1-300 gpurun_out/SAMPLER_r03_n8.json
timeout 300 python tools/bench_sampler.py --n 8 --dtype bf16 > gpurun_out/c8_sampler_bf16.log 2>&1; grep '^{' gpurun_out/c8_sampler_bf16.log | tail -1 > gpurun_out/SAMPLER_r03_n8_bf16.json; cut -c1-300 gpurun_out/SAMPLER_r03_n8_bf16.json
timeout 400 python tools/run_config5.py --identities 1 > gpurun_out/c8_c5_f32.log 2>&1; grep '^{' gpurun_out/c8_c5_f32.log | tail -1 > gpurun_out/CONFIG5_r03_1gpu.json; cut -c1-400 gpurun_out/CONFIG5_r03_1gpu.json
timeout 400 python tools/run_config5.py --identities 1 --precision bf16 > gpurun_out/c8_c5_bf16.log 2>&1; grep '^{' gpurun_out/c8_c5_bf16.log | tail -1 > gpurun_out/CONFIG5_r03_1gpu_bf16.json; cut -c1-400 gpurun_out/CONFIG5_r03_1gpu_bf16.json
