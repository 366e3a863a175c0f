"""Importable alias of the package directory ``cmr-landmark-detection_amd/`` (a hyphen is not a Python
identifier):  ``import cmr_landmark_detection_amd as rvip; rvip.get_model(config)``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module('cmr-landmark-detection_amd')
sys.modules[__name__] = _pkg
