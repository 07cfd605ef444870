"""Zernike-moment container and index algebra (host side of the drop-in boundary).

Mirrors the public names and behaviour of ``mtflearn/features/_zmoments.py`` (reference):
``nm2j`` (:3-69), ``nm2j_complex`` (:71-91), ``check_array1d`` (:94-108),
``construct_complex_matrix`` (:111-132), ``construct_real_matrix`` (:134-196),
``construct_rot_maps_matrix`` (:199-235) and ``class zmoments`` (:238-493) -- same argument
meaning, return types, ordering conventions, exception types and messages -- so code written
against ``mtflearn.features`` runs unchanged.  The implementation is independent: the mixing
matrices are built by index arithmetic, and the container shares one moment-axis helper
between its rank-2 ``(N, N_poly)`` and rank-3 ``(N_poly, H, W)`` layouts.

The heavy producers of these containers are the HIP kernels behind ``ZPs.transform``; the
methods here are the small per-moment-axis contractions of SURVEY 8a rows 8-14.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "nm2j", "nm2j_complex", "check_array1d", "construct_complex_matrix",
    "construct_real_matrix", "construct_rot_maps_matrix", "zmoments",
]


def _require_integral(values, message):
    if not np.all(np.isclose(values % 1, 0)):
        raise ValueError(message)


def nm2j(n, m):
    """Single index of the real Zernike term (n, m): ``j = ((n + 2) n + m) // 2``.

    Scalars in give a Python ``int`` out, array-likes give an ``ndarray`` (reference
    ``_zmoments.py:3-69``; behaviour pinned by its ``tests/features/test_zmoments.py:5-75``).
    """
    n_arr, m_arr = np.asarray(n), np.asarray(m)
    if n_arr.shape != m_arr.shape:
        raise ValueError("`n` and `m` must have the same shape.")
    _require_integral(n_arr, "Radial order `n` must be integer-valued.")
    _require_integral(m_arr, "Azimuthal frequency `m` must be integer-valued.")
    n_int, m_int = n_arr.astype(int), m_arr.astype(int)
    if (n_int < 0).any():
        raise ValueError("Radial order `n` must be non-negative.")
    if (np.abs(m_int) > n_int).any():
        raise ValueError("Azimuthal frequency `m` must satisfy |m| ≤ n.")
    if ((n_int - np.abs(m_int)) % 2 != 0).any():
        raise ValueError("`n - |m|` must be even.")
    j = (n_int * (n_int + 2) + m_int) // 2
    return j.item() if j.ndim == 0 else j


def nm2j_complex(n, m):
    """Index of the complex term (n, m >= 0), counting one slot per (n, |m|).

    ``(n^2 + 2n + 2m) // 4`` for even n and ``(n^2 + 2n + 2m - 1) // 4`` for odd n
    (reference ``_zmoments.py:71-91``).
    """
    n_arr, m_arr = np.atleast_1d(n), np.atleast_1d(m)
    if not (n_arr >= 0).all():
        raise ValueError("Radial order n must be non-negative.")
    if not (m_arr >= 0).all():
        raise ValueError("Azimuthal frequency m must be non-negative.")
    if not (np.abs(m_arr) <= n_arr).all():
        raise ValueError("Azimuthal frequency m must satisfy |m| ≤ n.")
    if not ((n_arr - np.abs(m_arr)) % 2 == 0).all():
        raise ValueError("n - |m| must be even.")
    q = np.array(n_arr ** 2 + 2 * n_arr + 2 * m_arr)
    q = (q - (n_arr % 2)) // 4
    return q.item() if q.size == 1 else q


def check_array1d(data):
    """Scalar or array-like -> flat 1-D ndarray (reference ``_zmoments.py:94-108``)."""
    return np.atleast_1d(data).ravel()


def construct_complex_matrix(n, m):
    """Real -> complex mixing matrix ``C`` with ``Z^c_{n,|m|} = Z_{n,+m} + i Z_{n,-m}``.

    Rows follow ascending complex index, columns follow (n, m) lexicographic order; entries
    are 1 for m >= 0 and 1j for m < 0 (reference ``_zmoments.py:111-132``).
    """
    n_arr, m_arr = np.asarray(n), np.asarray(m)
    order = np.lexsort((m_arr, n_arr))
    n_arr, m_arr = n_arr[order], m_arr[order]
    slot = np.atleast_1d(nm2j_complex(n_arr, np.abs(m_arr)))
    _, row = np.unique(slot, return_inverse=True)
    mat = np.zeros((row.max() + 1 if row.size else 0, n_arr.size), dtype=complex)
    mat[row, np.arange(n_arr.size)] = np.where(m_arr >= 0, 1.0 + 0.0j, 1j)
    return mat


def construct_real_matrix(n, m):
    """Complex -> real mixing: returns ``(inv_matrix, n_real, m_real)``.

    ``inv_matrix`` has 1 where a real term is the real part of a complex one and -1j where it
    is the imaginary part, so ``real = (inv_matrix @ complex).real``
    (reference ``_zmoments.py:134-196``).
    """
    n_c, m_c = np.asarray(n), np.asarray(m)
    twice = m_c != 0
    n_real = np.concatenate([n_c, n_c[twice]])
    m_real = np.concatenate([m_c, -m_c[twice]])
    order = np.lexsort((m_real, n_real))
    n_real, m_real = n_real[order], m_real[order]
    forward = construct_complex_matrix(n=n_real, m=m_real)
    inv = np.zeros(forward.shape[::-1], dtype=complex)
    inv[forward.T == 1] = 1.0
    inv[forward.T == 1j] = -1j
    return inv, n_real, m_real


def construct_rot_maps_matrix(n_folds, m):
    """Per-fold weights over moments: +1 where ``|m| % fold == 0`` and ``|m| > 1``, 0 for
    ``|m|`` in {0, 1}, ``-1/(fold-1)`` elsewhere (0 when fold == 1)
    (reference ``_zmoments.py:199-235``)."""
    folds = check_array1d(n_folds)
    abs_m = np.abs(check_array1d(m))
    neutral = abs_m <= 1
    weights = np.zeros((len(folds), len(abs_m)))
    for r, fold in enumerate(folds):
        resonant = (abs_m % fold == 0) & ~neutral
        weights[r, resonant] = 1
        weights[r, ~(resonant | neutral)] = -1.0 / (fold - 1) if fold > 1 else 0
    return weights


class zmoments:
    """Zernike moments with their (n, m) labels.

    ``data`` is ``(N, N_poly)`` (batch of patches) or ``(N_poly, H, W)`` (dense frame);
    ``n``/``m`` label the moment axis; ``patch_size`` is the window the moments came from.
    The constructor canonicalises the moment axis to (n, m) lexicographic order
    (reference ``_zmoments.py:240-277``).
    """

    def __init__(self, data, n, m, patch_size=None):
        self._setup(data, n, m, patch_size, own=False)

    @classmethod
    def _adopt(cls, data, n, m, patch_size=None):
        """Container around an array the caller hands over (results this package just produced): no copy when
        the moment axis is already in (n, m) order.  The public constructor always copies, as the reference's
        fancy-index permutation does (``_zmoments.py:277``), so in-place edits never reach the caller's array."""
        self = cls.__new__(cls)
        self._setup(data, n, m, patch_size, own=True)
        return self

    def _setup(self, data, n, m, patch_size, own):
        self.n = np.asarray(n)
        self.m = np.asarray(m)
        self.data = np.asarray(data)
        self.patch_size = patch_size
        if self.n.shape != self.m.shape:
            raise ValueError("`n` and `m` must have the same shape.")
        if self.data.ndim not in (2, 3):
            raise ValueError("Data must be 2D or 3D array.")
        have = self.data.shape[self._axis]
        if have != len(self.n):
            raise ValueError(
                f"Data shape mismatch: expected {len(self.n)} moments but got {have}")
        order = np.lexsort((self.m, self.n))
        if not np.array_equal(order, np.arange(order.size)):
            self.data = np.take(self.data, order, axis=self._axis)
        elif not own:
            self.data = self.data.copy()
        self.n = self.n[order]
        self.m = self.m[order]

    # -- layout helpers --------------------------------------------------------------
    @property
    def _axis(self):
        """Index of the moment axis: 1 for rank-2 data, 0 for rank-3."""
        return 1 if self.data.ndim == 2 else 0

    def _like(self, data, n=None, m=None):
        return zmoments._adopt(data=data, n=self.n if n is None else n,
                        m=self.m if m is None else m, patch_size=self.patch_size)

    def _mix(self, matrix, data=None):
        """Contract ``matrix (rows, N_poly)`` with the moment axis, keeping the layout."""
        data = self.data if data is None else data
        if data.ndim == 2:
            return np.dot(data, matrix.T)
        return np.tensordot(matrix, data, axes=([1], [0]))

    # -- reference API ---------------------------------------------------------------
    @property
    def valid_mask(self):
        """Boolean (H, W) mask of positions whose window did not touch the zero padding, with
        the reference's edge convention (``_zmoments.py:279-294``): first ``(K-1)//2`` and
        last ``K-1-(K-1)//2`` rows/columns are False.  ``None`` for rank-2 data."""
        if self.data.ndim == 2 or self.patch_size is None:
            return None
        head = (self.patch_size - 1) // 2
        tail = self.patch_size - 1 - head
        mask = np.ones(self.data.shape[1:]).astype(bool)
        mask[:head, :] = False
        mask[-tail:, :] = False
        mask[:, :head] = False
        mask[:, -tail:] = False
        return mask

    @property
    def is_complex(self):
        return np.iscomplexobj(self.data)

    def to_complex(self):
        """``Z^c_{n,|m|} = Z_{n,m} + i Z_{n,-m}`` (``_zmoments.py:300-316``)."""
        if self.data.dtype == complex:
            return self
        mix = construct_complex_matrix(n=self.n, m=self.m)
        real_slot = (mix == 1).astype(float)
        return self._like(self._mix(mix),
                          n=real_slot.dot(np.abs(self.n)).astype(int),
                          m=real_slot.dot(np.abs(self.m)).astype(int))

    def to_real(self):
        """Inverse of :meth:`to_complex` (``_zmoments.py:318-341``)."""
        if self.data.dtype != complex:
            return self
        inv, n_real, m_real = construct_real_matrix(self.n, self.m)
        return self._like(self._mix(inv).real, n=n_real, m=m_real)

    def normalize(self, order=None):
        """Divide by the ``order``-norm over the moment axis, no epsilon
        (``_zmoments.py:344-356``)."""
        norms = np.linalg.norm(self.data, ord=order, axis=self._axis, keepdims=True)
        return self._like(self.data / norms)

    def select(self, m_select):
        """Keep moments whose ``|m|`` is in ``m_select`` (order kept; ``_zmoments.py:359-369``)."""
        wanted = np.unique(np.abs(check_array1d(m_select)))
        keep = np.flatnonzero(np.isin(np.abs(self.m), wanted))
        return self._like(np.take(self.data, keep, axis=self._axis), n=self.n[keep], m=self.m[keep])

    def unselect(self, m_unselect):
        """Drop moments whose ``|m|`` is in ``m_unselect`` (``_zmoments.py:371-374``)."""
        drop = check_array1d(m_unselect)
        return self.select(np.array([v for v in np.unique(np.abs(self.m)) if v not in drop]))

    def rotate(self, theta):
        """Moments of the pattern rotated by ``theta`` degrees: ``Z^c exp(-i m theta)``,
        returned in complex form (``_zmoments.py:377-418``)."""
        zc = self.to_complex()
        phase = np.exp(-1j * np.deg2rad(theta) * zc.m)
        if zc.data.ndim == 3:
            phase = phase[:, None, None]
        return zmoments._adopt(data=zc.data * phase, n=zc.n, m=zc.m, patch_size=self.patch_size)

    def _prepared(self, m_unselect, p):
        picked = self.unselect(m_unselect)
        return picked if p is None else picked.normalize(order=p)

    def rot_maps(self, n_folds, p=2, m_unselect=None):
        """n-fold rotational-symmetry scores ``W . Zhat^2`` per patch / pixel
        (``_zmoments.py:420-462``).  ``m_unselect`` must contain 0."""
        if self.data.ndim not in (2, 3):
            raise ValueError("Input must be a 2D or 3D array.")
        if m_unselect is None:
            m_unselect = (0, 1)
        elif 0 not in m_unselect:
            raise ValueError("m=0 must be included in m_unselect.")
        zm = self._prepared(m_unselect, p)
        return zm._mix(construct_rot_maps_matrix(n_folds, zm.m), zm.data ** 2)

    def mirror_map(self, theta=None, p=2, m_unselect=(0, 1)):
        """Mirror-symmetry score: ``max_theta sum_k Re[(Zhat^c_k)^2 exp(-i m_k theta)]`` over
        the angle grid (default 360 angles in [0, 2 pi)) (``_zmoments.py:464-493``)."""
        if theta is None:
            theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
        zc = self._prepared(m_unselect, p).to_complex()
        re, im = zc.data.real, zc.data.imag
        stacked = np.concatenate([re ** 2 - im ** 2, 2 * re * im], axis=zc._axis)
        angles = np.multiply.outer(np.asarray(theta), zc.m)
        table = np.hstack([np.cos(angles), np.sin(angles)])
        return zc._mix(table, stacked).max(axis=zc._axis)
