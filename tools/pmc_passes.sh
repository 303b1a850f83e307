# rocprofv3 counter passes over bench.py (one counter group per pass); usage: bash tools/pmc_passes.sh [set]
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
SET=${1:-core}
EXTRA=${2:-}                      # extra bench flags, e.g. "--hidden 512 --points 125000" (output dirs then carry the tag $3)
TAGX=${3:-}
if [ "$SET" = core ]; then
CSETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
        "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
        "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
else
CSETS=("TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
        "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
        "TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum" "TCC_BUSY_sum TCC_CYCLE_sum" \
        "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
        "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM" \
        "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" \
        "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" \
        "TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum" "TCC_BUBBLE_sum TCC_NORMAL_WRITEBACK_sum")
fi
i=0
for C in "${CSETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_${SET}${TAGX}_$i -o pmc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config3 $EXTRA > $R/gpurun_out/pmc_${SET}${TAGX}_$i.log 2>&1
  grep -c . $R/gpurun_out/pmc_${SET}${TAGX}_$i/pmc_counter_collection.csv || grep -i 'unable' $R/gpurun_out/pmc_${SET}${TAGX}_$i.log | cut -c1-200
done
