#!/bin/bash
# Round 6: where does the dense kernel's time go?  (a) the development build's AOC_DENSE_DEBUG bits (WRONG results by design: parts switched off) at R = 6,
# bench pools; (b) SQ counters of the release kernel alone.  Output: gpurun_out/r06a_dense/
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06a_dense
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for dbg in 0 2 46 302 558 814 256 512; do
  echo "== AOC_DENSE_DEBUG=$dbg  (1 no publish, 2 no rescoring, 4 no row DMA, 8 no step barrier, 32 no bound DMA, 256 no decision, 512 no fragment reads)"
  AOC_LIB_VARIANT=dev AOC_DENSE_DEBUG=$dbg python tools/bench_dense.py 6 2>&1 | grep -E "^split|rescored"
done
} > "$out/debug_bits2.txt" 2>&1
{
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS" "SQ_WAIT_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  echo "--pmc $set"
  tools/pmc_kernel.sh dense_prune "$set" python tools/bench_dense.py 6
done
} > "$out/pmc_sq.txt" 2>&1
tail -5 "$out/debug_bits2.txt"
