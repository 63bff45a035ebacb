cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dense_split.py -q -m gpu -x -k "kmeans or cluster or replica" 2>&1 | tail -2
run() { # label so config
  AB_SO=$2 python tools/_ab/run.py --config $3 --no-extras --no-cpu-baseline --exact-steps 0 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$3 $1', d['value'], d['ms_per_step'])"
}
for v in base new; do so=tools/_ab/libaoc_base.so; [ $v = new ] && so=robust-video-object-segmentation_amd/csrc/libaoc_hip.so; AB_SO=$so AB_KM="6 1" python tools/_ab/run.py 2>&1 | grep "k-means chain" | sed "s/^/$v /"; AB_SO=$so AB_KM="6 3" python tools/_ab/run.py 2>&1 | grep "k-means chain" | sed "s/^/$v /"; done
for rep in 1 2 3; do
  for c in cfg2 cfg3; do
  run new robust-video-object-segmentation_amd/csrc/libaoc_hip.so $c
  run base tools/_ab/libaoc_base.so $c
  done
done
