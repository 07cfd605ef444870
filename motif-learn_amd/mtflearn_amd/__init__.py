"""mtflearn_amd -- MI355X-native drop-in for motif-learn's ``ZPs`` / ``zmoments`` hot path.

``from mtflearn_amd import ZPs, zmoments`` mirrors ``from mtflearn import ZPs, zmoments``
(reference ``mtflearn/__init__.py:37-38``).  Only this path is provided; see DESIGN.md.
"""
__version__ = "0.1.0"

from .features import ZPs, zmoments
from . import features

__all__ = ["ZPs", "zmoments", "features"]
