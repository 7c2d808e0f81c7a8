#!/bin/bash
# PMC passes over one bench step per arithmetic mode -> gpurun_out/r03_pmc_summary.json (copy to profiles/ afterwards).
set -u
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in fp32 bf16x6; do
  for p in "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    rocprofv3 --kernel-trace --pmc ${p#*:} -d /tmp/pmcw/pmc_${m}_${p%%:*} -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --math $m --no-cpu-baseline --no-bf16x6 --also none > $O/pmc_${m}_${p%%:*}.log 2>&1
  done
done
cd $R
python3 tools/pmc_summary.py /tmp/pmcw r03 > $O/pmc_summary.log 2>&1
cp /tmp/pmcw/r03_pmc_summary.json $O/ 2>/dev/null
ls -la $O | tail -5
