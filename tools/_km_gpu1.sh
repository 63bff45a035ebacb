AOC_KM_PROF=1 python tools/bench_kmeans.py 6 1 2>&1 | tail -5
AOC_KM_PROF=101 python tools/bench_kmeans.py 6 1 2>&1 | tail -5
