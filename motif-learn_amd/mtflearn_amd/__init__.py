"""mtflearn_amd -- MI355X-native drop-in for motif-learn's ``ZPs`` / ``zmoments`` hot path.

``from mtflearn_amd import ZPs, zmoments`` mirrors ``from mtflearn import ZPs, zmoments``
(reference ``mtflearn/__init__.py:37-38``).  Only this path and the rows SURVEY 8(f) names around it are provided
(``mtflearn_amd.features``: parameter pickers, ``pca``; ``mtflearn_amd.clustering``: ``kmeans_lbs`` / ``gmm_lbs`` /
``sort_lbs``; ``mtflearn_amd.manifold``: ``ForceGraph8``); see DESIGN.md.
"""
__version__ = "0.1.0"

from .features import ZPs, zmoments
from . import features

__all__ = ["ZPs", "zmoments", "features"]
