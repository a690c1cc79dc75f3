"""MI355X-native DenseASPP forward/backward path (HIP kernels behind an nn.Module).

The directory name contains a hyphen, so it cannot be imported with a plain ``import``
statement; ``denseASPP.py`` at the repository root loads it under the module name
``semantic_segmentation_amd`` and re-exports :class:`DenseASPP`.
"""
from .dense_aspp import DenseASPP  # noqa: F401
from .spec import NetSpec  # noqa: F401

__all__ = ['DenseASPP', 'NetSpec']
