"""Hot-path subset of ``mtflearn.features`` (reference ``mtflearn/features/__init__.py:1-4,11-12``)."""
from .zernike_polys import ZPs
from .moments import (zmoments, construct_rot_maps_matrix, construct_complex_matrix,
                      construct_real_matrix, nm2j, nm2j_complex, check_array1d)

__all__ = ["ZPs", "zmoments", "construct_rot_maps_matrix", "construct_complex_matrix",
           "construct_real_matrix", "nm2j", "nm2j_complex", "check_array1d"]
