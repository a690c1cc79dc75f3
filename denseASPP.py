"""``from denseASPP import DenseASPP`` -- the import the reference scripts use
(reference: graph_seg_test.py:12, graph_seg_main.py:16).

Loads the package in ``semantic-segmentation_amd/`` (hyphenated directory) under the
importable name ``semantic_segmentation_amd``.
"""
import importlib.util
import os
import sys

_PKG = 'semantic_segmentation_amd'


def _load_package():
    if _PKG in sys.modules:
        return sys.modules[_PKG]
    root = os.path.dirname(os.path.abspath(__file__))
    pkg_dir = os.path.join(root, 'semantic-segmentation_amd')
    spec = importlib.util.spec_from_file_location(_PKG, os.path.join(pkg_dir, '__init__.py'),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_PKG] = mod
    spec.loader.exec_module(mod)
    return mod


_pkg = _load_package()
DenseASPP = _pkg.DenseASPP
__all__ = ['DenseASPP']
