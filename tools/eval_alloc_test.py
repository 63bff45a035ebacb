import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import aoc_amd
from aoc_amd import eval_runner
dev = torch.device("cuda:0")
specs = eval_runner.make_sequence_set("davis17", scale=0.27, seed=0)
np.random.seed(1234)
with torch.no_grad():
    eval_runner.eval_sharded(specs[:1], 0, 1, dev, max_frames=3)
    for tag in ["first", "second", "third", "after empty_cache", "again", "after empty_cache", "again"]:
        if tag.startswith("after"):
            torch.cuda.empty_cache()
        st0 = torch.cuda.memory_stats()
        tot = eval_runner.eval_sharded(specs, 0, 1, dev, lanes=4)
        st1 = torch.cuda.memory_stats()
        print(f"{tag}: {tot['frames'] / float(tot['loop_seconds_max']):.1f} frames/s; hipMalloc calls during the run (incl. sequence upload): {st1['num_device_alloc'] - st0['num_device_alloc']}, frees {st1['num_device_free'] - st0['num_device_free']}, reserved {st1['reserved_bytes.all.current'] / 1e9:.2f} GB", flush=True)
