#!/bin/bash
# Counter passes for the fused kernel (run on the GPU box from the repo root).
# Each pass is its own rocprofv3 run: counters are never combined with trace domains.
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$REPO/$OUT/$name" -- python3 "$REPO/tools/run_forward.py" --steps 2 --rounds 1 > "$REPO/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$REPO/$OUT/$name.log"; return 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD && \
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS && \
pass sq3 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS && \
pass tcc1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum && \
pass tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum && \
pass tcc3 TCC_HIT_sum TCC_MISS_sum && \
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
echo "pmc done"
