#!/bin/bash
# PMC passes on the isolated F(4x4) weight-gradient launch (tools/time_wgrad.py), one library per argument (default: the product)
R=${GRAFT_REPO_ROOT:-$PWD}; export TMPDIR=/tmp; cd /tmp
for n in ${W4G_ABLS:-0}; do
  lib=$R/bmcnet-esr_amd/csrc/libbmc_hip_w4gabl$n.so; [ $n = 0 ] && lib=$R/bmcnet-esr_amd/csrc/libbmc_hip.so
  i=0
  for p in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum" "FETCH_SIZE"; do
    i=$((i+1))
    TW_B=${TW_B:-8} TW_ONLY=winograd4 BMC_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc $p -d /tmp/w4gpmc_${n}_$i -o p -- python3 $R/tools/time_wgrad.py > /tmp/w4gpmc_${n}_$i.log 2>&1
    echo "abl $n pass $i: $(python3 $R/tools/pmc_agg.py $(find /tmp/w4gpmc_${n}_$i -name '*.db' | head -1) wino4_wgrad_kernel 2>&1 | tail -1)"
  done
done
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TD|TCC)_[A-Z0-9_]+" | sort -u | tr '\n' ' ' | head -c 6000
