import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aoc_amd
if os.environ.get("AB_SO"):
    aoc_amd._lib.SO_PATH = os.path.abspath(os.environ["AB_SO"])
if os.environ.get("AB_KM"):
    sys.argv = [sys.argv[0]] + os.environ["AB_KM"].split()
    import runpy
    runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench_kmeans.py"), run_name="__main__")
else:
    import bench
    bench.main()
