#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/pytest_ops_7.log
tail -4 gpurun_out/pytest_ops_7.log
python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "mini" 2>&1 | tail -25 > gpurun_out/pytest_models_7.log
tail -4 gpurun_out/pytest_models_7.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_7.log 2>&1
tail -1 gpurun_out/bench_7.log | cut -c1-300
python tools/bench_kernels.py --iters 10 > gpurun_out/kernels_bench_7.log 2>&1
cp gpurun_out/kernels_bench.json gpurun_out/kernels_bench_7.json
REPO=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof7 -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/rocprof_7.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $REPO/gpurun_out/pmc7a -o conv --output-format csv -- python $REPO/tools/pmc_conv.py > $REPO/gpurun_out/pmc_7a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU -d $REPO/gpurun_out/pmc7b -o conv --output-format csv -- python $REPO/tools/pmc_conv.py > $REPO/gpurun_out/pmc_7b.log 2>&1
cd $REPO; ls gpurun_out/prof7
find gpurun_out -name "*.csv" -size +8M -delete
