#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "not sd14" 2>&1 | tail -6 > gpurun_out/pytest_17.log
tail -3 gpurun_out/pytest_17.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_17.log 2>&1
tail -1 gpurun_out/bench_17.log | cut -c1-300
REPO=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $REPO/gpurun_out/pmc17a -o conv --output-format csv -- python $REPO/tools/pmc_conv.py > $REPO/gpurun_out/pmc_17a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_WAIT_INST_LDS -d $REPO/gpurun_out/pmc17b -o conv --output-format csv -- python $REPO/tools/pmc_conv.py > $REPO/gpurun_out/pmc_17b.log 2>&1
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof17 -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/rocprof_17.log 2>&1
cd $REPO
find gpurun_out -name "*.csv" -size +8M -delete
