export AOC_KM_PROF=1
python tools/bench_kmeans.py 1 1 2>&1 | tail -2
python tools/bench_kmeans.py 6 1 2>&1 | tail -2
AOC_KM_GRID=64 python tools/bench_kmeans.py 6 1 2>&1 | tail -2
AOC_KM_GRID=512 python tools/bench_kmeans.py 6 1 2>&1 | tail -2
