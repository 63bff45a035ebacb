cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $out
B="--no-extras --no-cpu-baseline --exact-steps 0 --steps 10 --warmup 3"
for c in cfg3 cfg4; do
  python bench.py --config $c $B > $out/bench_$c.json 2> $out/bench_$c.err
  bash tools/kstats.sh 40 python $GRAFT_REPO_ROOT/bench.py --config $c $B --min-region-s 0.2 > $out/kstats_$c.txt 2>&1
done
python bench.py --eval-sharded --no-cpu-baseline > $out/eval_sharded.json 2> $out/eval_sharded.err
bash tools/kstats.sh 40 python $GRAFT_REPO_ROOT/bench.py --eval-sharded --no-cpu-baseline --eval-scale 0.015 > $out/kstats_eval.txt 2>&1
# RCCL path with one rank
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --exact-steps 0 > $out/torchrun1.json 2> $out/torchrun1.err
tail -c 600 $out/torchrun1.err
for f in bench_cfg3 bench_cfg4 eval_sharded torchrun1; do python - <<P
import json
try:
    d=json.loads(open("$out/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d.get("ms_per_step"), d.get("host_enqueue_ms_per_step"), d.get("strong_scaling"))
except Exception as e: print("$f", "ERR", e)
P
done
