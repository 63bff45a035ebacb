#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -f "$out/cond_second_read.txt"
for shape in "4 256 121 213" "4 512 61 107" "6 256 145 261" "9 256 181 321"; do
  rm -rf /tmp/prof_cw
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cw -- python $GRAFT_REPO_ROOT/tools/bench_cond_warm.py $shape 2> /dev/null | grep "^z =" >> "$out/cond_second_read.txt"
  python $GRAFT_REPO_ROOT/tools/bench_cond_warm.py --report /tmp/prof_cw >> "$out/cond_second_read.txt"
done
