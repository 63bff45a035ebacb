export AOC_LIB_VARIANT=dev
for g in 32 64 128 256 512 1024; do echo "grid $g"; AOC_KM_ASSIGN_GRID=$g bash tools/kstats.sh 3 python $GRAFT_REPO_ROOT/tools/bench_kmeans.py 6 3 | grep -i "assign\|heads"; done
