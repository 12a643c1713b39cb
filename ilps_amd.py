"""Importable alias for the hyphenated package directory `indirect_learning_pose-shape_amd/`.

`import ilps_amd` (or `importlib.import_module("indirect_learning_pose-shape_amd")`) both
yield the same module object; submodules resolve as `ilps_amd.keras_smpl.batch_smpl` etc.
"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("indirect_learning_pose-shape_amd")
sys.modules[__name__] = _pkg
