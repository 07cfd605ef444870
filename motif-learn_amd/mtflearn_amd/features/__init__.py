"""Hot-path subset of ``mtflearn.features`` (reference ``mtflearn/features/__init__.py:1-4,11-16``): the transformer,
the moment container and index algebra, the two routines that pick its parameters, and the key-point caller."""
from .zernike_polys import ZPs
from .moments import (zmoments, construct_rot_maps_matrix, construct_complex_matrix,
                      construct_real_matrix, nm2j, nm2j_complex, check_array1d)
from .pickers import (estimate_patch_size, radial_profile, autocorrelation, estimate_n_max, estimate_n_max_from_patch,
                      _get_cumulative_energy)
from .consumers import pca
from .keypoints import KeyPoints

__all__ = ["ZPs", "zmoments", "construct_rot_maps_matrix", "construct_complex_matrix",
           "construct_real_matrix", "nm2j", "nm2j_complex", "check_array1d",
           "estimate_patch_size", "radial_profile", "autocorrelation", "estimate_n_max", "estimate_n_max_from_patch", "pca",
           "KeyPoints"]
