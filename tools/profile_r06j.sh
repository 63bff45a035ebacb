#!/bin/bash
# Round 6: SQ counters of dense_prune_q4_kernel (libaoc_hip_q4.so) alone at R = 6 and 12.  Output: gpurun_out/r06b/dense_q4_pmc.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
export POOL_STRIDE=5 QUERY_OFFSET=3 AOC_LIB_FILE=libaoc_hip_q4.so
{
for R in 6 12; do
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  echo "--R=$R pmc $set"
  tools/pmc_kernel.sh dense_prune_q4 "$set" python tools/bench_dense.py $R
done
done
} > "$out/dense_q4_pmc.txt" 2>&1
cat "$out/dense_q4_pmc.txt"
