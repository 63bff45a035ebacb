set -x
cd $GRAFT_REPO_ROOT
B="python bench.py --no-extras --no-cpu-baseline --exact-steps 0 --steps 20 --warmup 5"
$B --dump-timeline gpurun_out/tl2.json > gpurun_out/s2.json 2> gpurun_out/s2.err
$B --streams 3 > gpurun_out/s3.json 2> gpurun_out/s3.err
$B --streams 4 > gpurun_out/s4.json 2> gpurun_out/s4.err
$B --streams 3 --cu-reserve 48 > gpurun_out/s3r48.json 2>&1
$B --streams 4 --cu-reserve 64 > gpurun_out/s4r64.json 2>&1
$B --streams 1 > gpurun_out/s1.json 2>&1
for f in s2 s3 s4 s3r48 s4r64 s1; do python - <<P
import json
try:
    d=json.loads(open("gpurun_out/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"])
except Exception as e: print("$f", "ERR", e)
P
done
