#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
KB_ITERS=20 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/p1 -o pmc --output-format csv -- python3 $R/tools/kbench.py conv3 > $O/r03i_p1.log 2>&1
KB_ITERS=20 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA -d /tmp/p2 -o pmc --output-format csv -- python3 $R/tools/kbench.py conv3 > $O/r03i_p2.log 2>&1
find /tmp/p1 -name "*counter_collection.csv" -exec cp {} $O/r03i_p1.csv \;
find /tmp/p2 -name "*counter_collection.csv" -exec cp {} $O/r03i_p2.csv \;
ls -la $O/r03i*
