"""Key points around the moment transform: the caller side of SURVEY 8(f)2 -- reference ``mtflearn/features/_keypoint.py``.

``KeyPoints(pts, img, size)`` keeps the reference's constructor, attributes (``pts`` after border clearing, ``img``, ``size``,
``shape``, ``patches``) and methods, restated on whole arrays instead of per-point Python loops.  ``extract_patches`` returns the
``(N, size, size)`` batch the reference cuts (``_keypoint.py:60-78``) for callers that want the windows themselves;
``moments(zps)`` is the device shortcut this package adds: the moments of those same windows straight from the frame resident on
the GPU (``ZPs.transform_at`` -> ``zk_transform_points``), no batch in memory.  The reference's quirks are reproduced as they
are: the method ``KeyPoints.clear_border`` bounds y by the image WIDTH (``_keypoint.py:80-84``), a window that would start left
of / above the frame comes back empty or short from NumPy's slicing and makes the batch ragged (border clearing prevents it).
"""
from __future__ import annotations

import numpy as np

__all__ = ["disk_patch", "center_of_mass_refine", "com_refine", "clear_border", "KeyPoints"]


def _interior(pts, x_limit, y_limit, size):
    """Rows of ``pts`` (x, y) strictly more than ``size // 2 + 1`` away from 0 and from the given limits."""
    margin = size // 2 + 1
    xy = np.asarray(pts)
    inside = (xy[:, 0] > margin) & (xy[:, 0] < x_limit - margin) & (xy[:, 1] > margin) & (xy[:, 1] < y_limit - margin)
    return pts[inside]


def clear_border(pts, shape, size):
    """Points farther than ``size // 2 + 1`` from every border of an image of ``shape`` (``_keypoint.py:46-50``)."""
    return _interior(pts, shape[1], shape[0], size)


def disk_patch(radius, dtype=np.uint8):
    """``(2 radius + 1)^2`` mask of the pixels within ``radius`` of the centre (``_keypoint.py:6-9``)."""
    d2 = np.add.outer(np.arange(-radius, radius + 1) ** 2, np.arange(-radius, radius + 1) ** 2)
    return (d2 <= radius ** 2).astype(dtype)


def center_of_mass_refine(data, pts, size=3, mode=None):
    """Intensity centroid of the ``(2 size + 1)^2`` box (``mode='disk'``: disk) around every integer point, rows ``(x, y)``
    (``_keypoint.py:12-23``).  As in the reference the regions are painted into ONE label image in point order -- where boxes
    overlap the later point owns the pixels (a disk also paints its box's corners with 0) -- and
    ``scipy.ndimage.center_of_mass`` reduces each label."""
    from scipy import ndimage
    stamp = disk_patch(size) if mode == 'disk' else 1
    owner = np.zeros_like(data)
    for label, (px, py) in enumerate(pts, start=1):
        owner[py - size:py + size + 1, px - size:px + size + 1] = label * stamp
    yx = np.array(ndimage.center_of_mass(data, owner, list(range(1, len(pts) + 1))))
    return yx[:, ::-1]


def com_refine(pts, img, size, threshold=None):
    """Thresholded centroid refinement of key points (``_keypoint.py:26-43``): every window is thresholded (Li by default,
    ``'otsu'``: Otsu -- scikit-image, exactly as the reference needs it) and its centroid moves the point."""
    from scipy import ndimage
    from skimage import filters
    level_of = filters.threshold_otsu if threshold == 'otsu' else filters.threshold_li
    kp = KeyPoints(pts, img, size)
    windows = kp.extract_patches(size)
    moved = np.empty((len(windows), 2))
    for k, w in enumerate(windows):
        w[w < level_of(w)] = 0
        moved[k] = np.subtract(ndimage.center_of_mass(w), size / 2)[::-1]
    return moved + kp.pts


class KeyPoints:
    """Key points of an image together with a window size (``_keypoint.py:53-92``)."""

    def __init__(self, pts, img, size):
        self.img, self.size, self.shape = img, size, img.shape
        self.pts = clear_border(pts, self.shape, size)
        self.patches = None

    def extract_patches(self, size=None, flat=False):
        """The window ``img[y - s1 : y + s2, x - s1 : x + s2]`` of every rounded point, ``s1 = size // 2``, ``s2 = size - s1``
        (``_keypoint.py:62-78``); ``flat=True`` flattens each window.  Windows that lie inside the frame (what border clearing
        leaves) are gathered in one indexing step; anything else falls back to the per-point slices of the reference."""
        size = self.size if size is None else size
        first = np.rint(self.pts).astype(int) - size // 2                                  # (x, y) of every window's corner
        height, width = self.img.shape[:2]
        whole = len(first) > 0 and first.min() >= 0 and (first[:, 0] + size <= width).all() and (first[:, 1] + size <= height).all()
        if whole and self.img.ndim == 2:
            view = np.lib.stride_tricks.sliding_window_view(self.img, (size, size))
            cut = view[first[:, 1], first[:, 0]]                                             # a copy: (N, size, size)
        else:
            cut = np.array([self.img[y:y + size, x:x + size] for x, y in first])
        self.patches = cut.reshape(len(cut), -1) if flat and cut.ndim == 3 else cut
        return self.patches

    def moments(self, zps):
        """Zernike moments of the windows ``extract_patches(zps.size)`` would cut, computed on the GPU from the frame itself
        (extension; ``zps``: a :class:`mtflearn_amd.ZPs`).  Same numbers as ``zps.transform(self.extract_patches(zps.size))``."""
        return zps.transform_at(self.img, self.pts)

    def clear_border(self, size):
        """In-place border clearing for another window size; y is bounded by ``shape[1]``, as in the reference."""
        self.pts = _interior(self.pts, self.shape[1], self.shape[1], size)

    def refine(self, data=None, r=3, mode=None):
        """Replace the points by the intensity centroids around their integer parts (``center_of_mass_refine``)."""
        self.pts = center_of_mass_refine(self.img if data is None else data, self.pts.astype(int), size=r, mode=mode)
