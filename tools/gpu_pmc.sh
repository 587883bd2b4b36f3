#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $REPO/gpurun_out/pmc6a -o conv --output-format csv -- python $REPO/tools/pmc_conv.py > $REPO/gpurun_out/pmc_6a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU -d $REPO/gpurun_out/pmc6b -o conv --output-format csv -- python $REPO/tools/pmc_conv.py > $REPO/gpurun_out/pmc_6b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_WAVE32_LDS SQ_LEVEL_WAVES -d $REPO/gpurun_out/pmc6c -o conv --output-format csv -- python $REPO/tools/pmc_conv.py > $REPO/gpurun_out/pmc_6c.log 2>&1
cd $REPO
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
find gpurun_out -name "*.csv" -size +8M -delete
