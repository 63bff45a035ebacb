cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $out
python -m pytest tests/test_gpu_round2.py -q -m gpu -k "jf" 2>&1 | tail -3
python -m pytest tests/test_gpu_eval_loop.py tests/test_gpu_closed_loop.py -q -m gpu 2>&1 | tail -3
python bench.py --eval-sharded --no-cpu-baseline > $out/eval_sharded2.json 2> $out/eval_sharded2.err
python - <<P
import json
d=json.loads(open("$out/eval_sharded2.json").read().strip().splitlines()[-1]); print("eval", d["value"], d["eval"])
P
bash tools/kstats.sh 12 python $GRAFT_REPO_ROOT/bench.py --eval-sharded --no-cpu-baseline --eval-scale 0.015 2>&1 | tail -12
