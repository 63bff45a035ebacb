"""MI355X-native AOC-Net matching / calibration hot path.

Python mirror of the reference's operator surface (networks/layers/matching.py,
networks/layers/attention.py, networks/aoc/conditioning_layer.py) over a C-ABI HIP library
(``csrc/libaoc_hip.so``, declared in ``include/aoc_hip.h``).  There is no CPU fallback: every
operator raises if the HIP library is missing or the tensors are not on the GPU.
"""
from . import synthetic  # noqa: F401
from . import _lib  # noqa: F401
from . import ops  # noqa: F401
from . import matching  # noqa: F401
from . import attention  # noqa: F401
from . import conditioning_layer  # noqa: F401
from . import hotpath  # noqa: F401
from . import sharding  # noqa: F401
from . import eval_loop  # noqa: F401
from . import gct  # noqa: F401
from . import eval_runner  # noqa: F401

__all__ = ["synthetic", "ops", "matching", "attention", "conditioning_layer", "hotpath", "sharding", "eval_loop", "gct", "eval_runner"]
