"""MI355X-native AOC-Net matching / calibration hot path.

Python mirror of the reference's operator surface (networks/layers/matching.py,
networks/layers/attention.py, networks/aoc/conditioning_layer.py) over a C-ABI HIP library
(``csrc/libaoc_hip.so``, declared in ``include/aoc_hip.h``).  There is no CPU fallback: every
operator raises if the HIP library is missing or the tensors are not on the GPU.
"""
import os as _os

# The evaluation runner keeps 4 sequences in flight, each on its own HIP stream plus a side stream for its k-means chains: 8 streams.  The HIP
# runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a queue execute in submission order -- a
# lane's frame then waits behind another lane's chain (measured: closed loop 356 frames/s at 4 queues, 377 at 8, 380-385 at 16;
# profiles/r05_closed_loop_hw_queues.txt).  The runtime reads the variable when it initialises (the first HIP call of the process), so it is
# set here, on import, unless the caller has set it; the C-ABI library itself reads no environment.
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    import sys as _sys
    _torch = _sys.modules.get("torch")
    if _torch is not None and _torch.cuda.is_initialized():
        # too late: the runtime has read the variable already.  Entry points (bench.py, tools/) import this package before their first torch.cuda call.
        import warnings as _warnings
        _warnings.warn("aoc_amd imported after the HIP runtime was initialised with GPU_MAX_HW_QUEUES unset: streams share the default 4 hardware queues "
                       "(closed loop ~6 % slower); export GPU_MAX_HW_QUEUES=16 or import aoc_amd before the first torch.cuda call", RuntimeWarning)
    _os.environ["GPU_MAX_HW_QUEUES"] = "16"

from . import synthetic  # noqa: F401,E402
from . import _lib  # noqa: F401,E402
from . import ops  # noqa: F401,E402
from . import matching  # noqa: F401,E402
from . import attention  # noqa: F401,E402
from . import conditioning_layer  # noqa: F401,E402
from . import hotpath  # noqa: F401,E402
from . import sharding  # noqa: F401,E402
from . import eval_loop  # noqa: F401,E402
from . import gct  # noqa: F401,E402
from . import eval_runner  # noqa: F401,E402

__all__ = ["synthetic", "ops", "matching", "attention", "conditioning_layer", "hotpath", "sharding", "eval_loop", "gct", "eval_runner"]
