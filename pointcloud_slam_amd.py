"""Importable alias of the package directory ``pointcloud-slam_amd/`` (a hyphen
cannot appear in an ``import`` statement): ``import pointcloud_slam_amd as pcm``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("pointcloud-slam_amd")
sys.modules[__name__] = _pkg
