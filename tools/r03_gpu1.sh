#!/bin/bash
# round-3 GPU call 1: new/changed parity tests, conv 3x3 A/B (round-2 padded halo rows vs 384-float rows), LDS-conflict PMC, quick bench
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd $R
( timeout 1500 python -m pytest tests/test_gpu_r3.py tests/test_gpu_r2.py -x -q -m gpu -s -k "two_ranks or inference or streaming or config3" 2>&1 | tail -40 ) > $O/r03a_tests_new.log 2>&1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_r2.py tests/test_isa_hygiene.py -q -m gpu -k "not config3 and not config4" 2>&1 | tail -15 ) > $O/r03a_tests_all.log 2>&1
for lib in libbmc_hip.so libbmc_hip_r02conv.so; do
  echo "== $lib" >> $O/r03a_kbench.log
  BMC_HIP_LIB=$R/bmcnet-esr_amd/csrc/$lib KB_ITERS=200 python tools/kbench.py conv3 >> $O/r03a_kbench.log 2>&1
  BMC_HIP_LIB=$R/bmcnet-esr_amd/csrc/$lib KB_ITERS=200 KB_B=16 python tools/kbench.py conv3 >> $O/r03a_kbench.log 2>&1
done
cd /tmp
for lib in libbmc_hip.so libbmc_hip_r02conv.so; do
  export BMC_HIP_LIB=$R/bmcnet-esr_amd/csrc/$lib
  KB_ITERS=20 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d /tmp/pmc_$lib -o pmc --output-format csv -- python3 $R/tools/kbench.py conv3 > $O/r03a_pmc_$lib.log 2>&1
  find /tmp/pmc_$lib -name "*counter_collection.csv" -exec cp {} $O/r03a_pmc_${lib}_counters.csv \;
done
unset BMC_HIP_LIB
cd $R
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-bf16x6 > $O/r03a_bench.json 2> $O/r03a_bench.err
tail -3 $O/r03a_tests_new.log $O/r03a_tests_all.log; cat $O/r03a_kbench.log; head -c 600 $O/r03a_bench.json
