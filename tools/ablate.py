"""Developer tool (GPU box): where does a bench step go?  Runs bench.py's timed region with parts of the hot path replaced by no-ops
(the results are then WRONG: this only measures sensitivities).  AOC_ABLATE = comma list of {dense, gates, local, kmeans, corr}.
Usage: AOC_ABLATE=dense python tools/ablate.py --python-frames --no-extras --steps 30 --no-cpu-baseline --exact-steps 0
(--python-frames: the ops are patched at the Python level, which the one-call-per-frame path of round 4 does not go through)"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

aoc = importlib.import_module("robust-video-object-segmentation_amd")
ops, hotpath = aoc.ops, aoc.hotpath
what = set(filter(None, os.environ.get("AOC_ABLATE", "").split(",")))
if "dense" in what:
    ops.dense_match = lambda *a, **k: None
if "local" in what:
    real_local = ops.local_window_match
    cache = {}

    def fake_local(q2, prev_map, bits2, radii, bias, O, transform, **k):
        key = (q2.shape, O, len(radii))
        if key not in cache:
            cache[key] = real_local(q2, prev_map, bits2, radii, bias, O, transform, **k)
        return cache[key]
    ops.local_window_match = fake_local
if "corr" in what:
    ops.proxy_corr_min = lambda *a, **k: None
    ops.proxy_corr_min_records = lambda *a, **k: (lambda: None) if k.get("prepare_only") else None      # (bench.correlation_roofline asks for a prepared launch)
    ops.proxy_corr_min_batched = lambda *a, **k: None
if "kmeans" in what:
    # keep the launch structure but run a single Lloyd iteration
    hotpath.KMEANS_ITERS = int(os.environ.get("AOC_ABLATE_ITERS", "1"))
if "gates" in what:
    orig = bench.frame_step

    class _NoGates:
        def __init__(self, g):
            self.g = g

        def __call__(self, acts, head):
            return []

        def __getattr__(self, n):
            return getattr(self.g, n)

    def frame_step(wl, gates, acts, *a, **k):
        return orig(wl, _NoGates(gates), acts, *a, **k)
    bench.frame_step = frame_step
print("ablate:", sorted(what), file=sys.stderr)
bench.main()
