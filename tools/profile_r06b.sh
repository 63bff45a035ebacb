#!/bin/bash
# Round 6: wave priorities in the dense kernel (development build, AOC_DENSE_DEBUG 1024 = static priority for waves 4..7, 2048 = s_setprio 1 around each
# tile's 14 MFMAs) + the MFMA pacing probe.  Output: gpurun_out/r06b/
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r06b
mkdir -p "$out"
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_probe2 tools/probe/mfma_probe2.hip && /tmp/mfma_probe2 > "$out/mfma_probe2.txt"
export POOL_STRIDE=5 QUERY_OFFSET=3
{
for dbg in 0 1024 2048 3072 0 1024 2048 3072; do
  echo "== AOC_DENSE_DEBUG=$dbg"
  AOC_LIB_VARIANT=dev AOC_DENSE_DEBUG=$dbg python tools/bench_dense.py 6 2>&1 | grep -E "^split|rescored|max"
done
} > "$out/dense_prio.txt" 2>&1
cat "$out/dense_prio.txt" | grep -E "^==|^split"
