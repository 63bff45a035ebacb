"""Importable alias for the product package.

The package directory is named ``robust-video-object-segmentation_amd`` (fixed by the project
layout); a hyphen cannot appear in an ``import`` statement, so ``import aoc_amd`` resolves to it.
Use attribute access / ``from aoc_amd import matching`` (sub-modules are imported eagerly by the
package), not ``import aoc_amd.matching``.
"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("robust-video-object-segmentation_amd")
