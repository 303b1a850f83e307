set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES" \
         "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum" "TCC_BUSY_sum TCC_CYCLE_sum" \
         "TCC_HIT_sum TCC_MISS_sum" "TCC_SRC_FIFO_FULL_sum TCC_IB_STALL_sum" "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmcf_$i -o pmc --output-format csv -- python3 $R/tools/fwd_variants.py > $R/gpurun_out/pmcf_$i.log 2>&1
done
ls $R/gpurun_out | grep -c pmcf
