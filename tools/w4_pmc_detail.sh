#!/bin/bash
# Detailed PMC passes over the ISOLATED F(4x4) convolution launch (tools/time_wino4.py, 8 x 180x240 x 128 -> 128): where the wave
# cycles go (instruction classes, waits), average vector-memory / LDS latencies, L1 / L2 behaviour.  Counters that this chip's
# rocprofv3 does not list are dropped from a pass.   usage (GPU box): bash tools/w4_pmc_detail.sh <tag> [lib]
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; TAG=$1; LIB=${2:-$R/bmcnet-esr_amd/csrc/libbmc_hip.so}
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > /tmp/counters.txt 2>&1
have() { grep -qw "$1" /tmp/counters.txt; }
declare -a PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_VMEM SQ_INSTS_WAVE32_LDS"
 "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum"
 "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
)
: > $O/${TAG}_pmc_detail.log
i=0
for p in "${PASSES[@]}"; do
  i=$((i+1)); sel=""
  for c in $p; do have $c && sel="$sel $c" || echo "pass $i: no counter $c" >> $O/${TAG}_pmc_detail.log; done
  [ -z "$sel" ] && continue
  rm -rf /tmp/w4d_$i
  W4_ONLY=1 KB_ITERS=20 BMC_HIP_LIB=$LIB timeout 300 rocprofv3 --kernel-trace --pmc $sel -d /tmp/w4d_$i -o p -- python3 $R/tools/time_wino4.py 8 180 240 > /tmp/w4d_$i.log 2>&1
  db=$(find /tmp/w4d_$i -name '*.db' | head -1)
  if [ -n "$db" ]; then echo "pass $i: $(python3 $R/tools/pmc_agg.py $db wino4_conv)" >> $O/${TAG}_pmc_detail.log; else echo "pass $i FAILED: $(tail -3 /tmp/w4d_$i.log)" >> $O/${TAG}_pmc_detail.log; fi
done
cat $O/${TAG}_pmc_detail.log
