"""CPU oracle for the Zernike-moment hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, in plain NumPy/SciPy, what jiadongdan/motif-learn computes on the
``ZPs.transform`` -> ``zmoments`` path.  It exists to *check* the HIP product path; it is
never the thing shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``motif-learn_amd/mtflearn_amd``) must not import anything from ``oracle/``.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference's two hot-path
modules in the build container and stores their outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function below against those vectors, and
against the reference's own known-answer tests (``tests/features/test_zmoments.py:5-88``).

Every function cites the reference lines it follows (paths relative to the reference root).
"""
from __future__ import annotations

import numpy as np
from scipy.signal import fftconvolve
from scipy.special import factorial

__all__ = [
    "radial_polynomial", "zernike_basis", "unit_disk_area",
    "moments_patches", "moments_frame_fft", "moments_frame_direct", "convolution_basis",
    "nm2j", "nm2j_complex", "complex_matrix", "real_matrix", "rot_maps_matrix",
    "sort_by_nm", "valid_mask", "to_complex", "to_real", "normalize", "select", "unselect",
    "rotate", "rot_maps", "mirror_map",
]


# --------------------------------------------------------------------------------------
# basis  (mtflearn/features/_zps.py:52-90)
# --------------------------------------------------------------------------------------
def radial_polynomial(n: int, m: int, rho: np.ndarray) -> np.ndarray:
    """R_n^{|m|}(rho); same term order and float factorials as ``_zps.py:52-64``."""
    am = abs(m)
    out = np.zeros_like(rho)
    for k in range((n - am) // 2 + 1):
        num = (-1) ** k * factorial(n - k)
        den = factorial(k) * factorial((n + am) // 2 - k) * factorial((n - am) // 2 - k)
        out += (num / den) * rho ** (n - 2 * k)
    return out


def zernike_basis(n_max: int, size: int):
    """(n, m, V) with V of shape (N_poly, size, size), float64 (``_zps.py:66-90``).

    Grid ``linspace(-1, 1, size)`` in x and y, ``rho <= 1`` mask, normalisation
    ``sqrt(2(n+1)/(1+[m==0]))``, ``sin(-m*theta)`` for m<0 and ``cos(m*theta)`` otherwise,
    order n ascending then m ascending.
    """
    axis = np.linspace(-1, 1, size)
    xv, yv = np.meshgrid(axis, axis)
    rho = np.sqrt(xv ** 2 + yv ** 2)
    theta = np.arctan2(yv, xv)
    ns, ms, polys = [], [], []
    for n in range(n_max + 1):
        for m in range(-n, n + 1, 2):
            radial = radial_polynomial(n, m, rho)
            norm = np.sqrt(2 * (n + 1) / (1 + (m == 0)))
            v = np.where(rho <= 1, radial * norm, 0)
            v = v * np.sin(-m * theta) if m < 0 else v * np.cos(m * theta)
            ns.append(n)
            ms.append(m)
            polys.append(v)
    return np.array(ns), np.array(ms), np.array(polys)


def unit_disk_area(size: int) -> float:
    """The constant both transform paths divide by (``_zps.py:154`` and ``:177``)."""
    return np.pi * size ** 2 / 4


# --------------------------------------------------------------------------------------
# transforms  (mtflearn/features/_zps.py:146-193)
# --------------------------------------------------------------------------------------
def moments_patches(patches: np.ndarray, basis: np.ndarray) -> np.ndarray:
    """Batch path, ``_zps.py:146-157``: one GEMM, (N,K,K) -> (N,N_poly) float64."""
    size = basis.shape[-1]
    flat_b = basis.reshape(-1, size * size)
    flat_p = patches.reshape(patches.shape[0], size * size)
    return np.dot(flat_p, flat_b.T) / unit_disk_area(size)


def moments_frame_fft(image: np.ndarray, basis: np.ndarray, n: np.ndarray) -> np.ndarray:
    """Dense path exactly as the reference runs it, ``_zps.py:159-193``.

    ``fftconvolve(broadcast image, basis, mode='same')``, the ``(-1)^n`` sign that undoes
    the kernel flip, and the area division.  With a float32 image SciPy transforms the image
    in single precision, so this function carries ~1e-7*max|Z| noise there.
    """
    size = basis.shape[-1]
    stack = np.broadcast_to(image, (len(n),) + image.shape)
    conv = fftconvolve(stack, basis, mode="same", axes=[1, 2])
    sign = 1 - n % 2
    sign[sign == 0] = -1
    return sign[:, None, None] * conv / unit_disk_area(size)


def moments_frame_direct(image: np.ndarray, basis: np.ndarray, rows=None, cols=None) -> np.ndarray:
    """What ``_zps.py:159-193`` computes, evaluated without the FFT (exact definition).

    Output pixel (i, j) is the inner product of the basis with the zero-padded window
    ``image[i-ea : i+eb+1, j-ea : j+eb+1]`` where ``eb=(K-1)//2`` and ``ea=K-1-eb``; this is
    the alignment ``mode='same'`` + the ``(-1)^n`` fix produce (verified to
    1e-16 against the FFT path on float64 input).  ``rows``/``cols`` restrict the evaluated
    output positions (for spot checks on large frames); default is every position.
    """
    n_poly, size, _ = basis.shape
    h, w = image.shape
    eb = (size - 1) // 2
    ea = size - 1 - eb
    padded = np.zeros((h + size - 1, w + size - 1), dtype=np.float64)
    padded[ea:ea + h, ea:ea + w] = image
    rows = np.arange(h) if rows is None else np.asarray(rows)
    cols = np.arange(w) if cols is None else np.asarray(cols)
    flat_b = basis.reshape(n_poly, size * size)
    out = np.empty((n_poly, len(rows), len(cols)), dtype=np.float64)
    win = np.lib.stride_tricks.sliding_window_view(padded, (size, size))
    for a, i in enumerate(rows):
        block = win[i, cols].reshape(len(cols), size * size)
        out[:, a, :] = (block @ flat_b.T).T
    return out / unit_disk_area(size)


def convolution_basis(basis: np.ndarray, n: np.ndarray) -> np.ndarray:
    """The numbers the reference's dense path multiplies a window with: ``(-1)^n * V[::-1, ::-1]``.

    ``_zps.py:165`` CONVOLVES the image with ``V`` (the kernel is applied point-flipped) and ``:173-178`` multiplies by
    ``(-1)^n``, which undoes the flip for an exactly point-symmetric ``V``.  The reference's float64 basis is
    point-symmetric only up to rounding -- 1e-13 of max|V| at n_max 10, 3e-10 at 20, 1e-8 at 24 -- so at high orders
    ``moments_frame_direct(image, V)`` (the inner-product definition) is that far from the reference's output while
    ``moments_frame_direct(image, convolution_basis(V, n))`` restates it to 1e-16 at every order
    (tests/test_oracle_golden.py::test_structured_goldens_pin_the_oracle)."""
    sign = np.where(np.asarray(n) % 2 == 0, 1.0, -1.0)
    return sign[:, None, None] * basis[:, ::-1, ::-1]


def moments_frame_at(image: np.ndarray, basis: np.ndarray, rows, cols) -> np.ndarray:
    """``moments_frame_direct`` at the PAIRED positions ``(rows[k], cols[k])`` -> ``(len(rows), n_poly)``: the same
    zero-padded windows and the same inner products (``_zps.py:159-193`` without the FFT), the frame padded once
    -- for spot checks on 4096 x 4096 frames."""
    n_poly, size, _ = basis.shape
    h, w = image.shape
    eb = (size - 1) // 2
    ea = size - 1 - eb
    padded = np.zeros((h + size - 1, w + size - 1), dtype=np.float64)
    padded[ea:ea + h, ea:ea + w] = image
    flat_b = basis.reshape(n_poly, size * size)
    win = np.lib.stride_tricks.sliding_window_view(padded, (size, size))
    block = win[np.asarray(rows), np.asarray(cols)].reshape(len(rows), size * size)
    return (block @ flat_b.T) / unit_disk_area(size)


# --------------------------------------------------------------------------------------
# index maps and mixing matrices  (mtflearn/features/_zmoments.py:3-235)
# --------------------------------------------------------------------------------------
def nm2j(n, m):
    """``j = ((n+2)n + m)//2`` with the reference's validation (``_zmoments.py:3-69``)."""
    n = np.asarray(n)
    m = np.asarray(m)
    if n.shape != m.shape:
        raise ValueError("`n` and `m` must have the same shape.")
    if not np.all(np.isclose(n % 1, 0)):
        raise ValueError("Radial order `n` must be integer-valued.")
    if not np.all(np.isclose(m % 1, 0)):
        raise ValueError("Azimuthal frequency `m` must be integer-valued.")
    n = n.astype(int)
    m = m.astype(int)
    if np.any(n < 0):
        raise ValueError("Radial order `n` must be non-negative.")
    if np.any(np.abs(m) > n):
        raise ValueError("Azimuthal frequency `m` must satisfy |m| ≤ n.")
    if np.any((n - np.abs(m)) % 2 != 0):
        raise ValueError("`n - |m|` must be even.")
    j = ((n + 2) * n + m) // 2
    return j.item() if j.shape == () else j


def nm2j_complex(n, m):
    """Index of the (n, m>=0) complex moment (``_zmoments.py:71-91``)."""
    n = np.atleast_1d(n)
    m = np.atleast_1d(m)
    if not np.all(n >= 0):
        raise ValueError("Radial order n must be non-negative.")
    if not np.all(m >= 0):
        raise ValueError("Azimuthal frequency m must be non-negative.")
    if not np.all(np.abs(m) <= n):
        raise ValueError("Azimuthal frequency m must satisfy |m| ≤ n.")
    if not np.all((n - np.abs(m)) % 2 == 0):
        raise ValueError("n - |m| must be even.")
    i = np.array(n ** 2 + 2 * n + 2 * m)
    even = np.array(n) % 2 == 0
    i[even] = i[even] // 4
    i[~even] = (i[~even] - 1) // 4
    return i.item() if i.size == 1 else i


def complex_matrix(n, m):
    """(N_c, N_real) matrix with 1 at (n, m>=0) and 1j at (n, m<0) (``_zmoments.py:111-132``)."""
    order = np.lexsort((m, n))
    n = np.asarray(n)[order]
    m = np.asarray(m)[order]
    jc = np.atleast_1d(nm2j_complex(n, np.abs(m)))
    rows = {v: r for r, v in enumerate(np.unique(jc))}
    mat = np.zeros((len(rows), len(n)), dtype=complex)
    for col, (key, mm) in enumerate(zip(jc, m)):
        mat[rows[key], col] = 1 if mm >= 0 else 1j
    return mat


def real_matrix(n, m):
    """Inverse mixing (``_zmoments.py:134-196``): returns (inv, n_real, m_real)."""
    n_real, m_real = [], []
    for nv, mv in zip(np.asarray(n), np.asarray(m)):
        if mv == 0:
            n_real.append(nv)
            m_real.append(0)
        else:
            n_real += [nv, nv]
            m_real += [mv, -mv]
    n_real = np.array(n_real)
    m_real = np.array(m_real)
    order = np.lexsort((m_real, n_real))
    n_real, m_real = n_real[order], m_real[order]
    fwd = complex_matrix(n_real, m_real)
    inv = np.zeros(fwd.T.shape, dtype=complex)
    inv[fwd.T == 1] = 1.0
    inv[fwd.T == 1j] = -1j
    return inv, n_real, m_real


def rot_maps_matrix(n_folds, m):
    """Fold weights (``_zmoments.py:199-235``)."""
    folds = np.atleast_1d(n_folds).ravel()
    am = np.abs(np.atleast_1d(m).ravel())
    mat = np.zeros((len(folds), len(am)))
    for r, f in enumerate(folds):
        hit = (am % f == 0) & (am > 1)
        special = (am == 0) | (am == 1)
        rest = ~(hit | special)
        mat[r, hit] = 1
        mat[r, rest] = -1.0 / (f - 1) if f > 1 else 0
    return mat


# --------------------------------------------------------------------------------------
# container behaviour as free functions on (data, n, m)  (_zmoments.py:238-493)
# --------------------------------------------------------------------------------------
def _moment_axis(data):
    if data.ndim == 2:
        return 1
    if data.ndim == 3:
        return 0
    raise ValueError("Data must be 2D or 3D array.")


def sort_by_nm(data, n, m):
    """Constructor's canonical ordering (``_zmoments.py:268-277``)."""
    n = np.asarray(n)
    m = np.asarray(m)
    data = np.asarray(data)
    order = np.lexsort((m, n))
    return np.take(data, order, axis=_moment_axis(data)), n[order], m[order]


def valid_mask(shape_hw, patch_size):
    """``_zmoments.py:279-294`` (including its even-size off-by-one, reproduced as is)."""
    mask = np.ones(shape_hw).astype(bool)
    eb = (patch_size - 1) // 2
    ea = patch_size - 1 - eb
    mask[:eb, :] = False
    mask[-ea:, :] = False
    mask[:, :eb] = False
    mask[:, -ea:] = False
    return mask


def to_complex(data, n, m):
    """``_zmoments.py:300-316``: returns (complex data, n_c, m_c)."""
    mat = complex_matrix(n, m)
    if data.ndim == 2:
        out = np.dot(mat, data.T).T
    else:
        out = np.tensordot(mat, data, axes=([1], [0]))
    pick = mat.copy()
    pick[pick == 1j] = 0
    m_c = pick.dot(np.abs(m)).real.astype(int)
    n_c = pick.dot(np.abs(n)).real.astype(int)
    return out, n_c, m_c


def to_real(data, n, m):
    """``_zmoments.py:318-341``: returns (real data, n_r, m_r)."""
    inv, n_r, m_r = real_matrix(n, m)
    if data.ndim == 2:
        out = np.dot(data, inv.T).real
    else:
        out = np.tensordot(inv, data, axes=([1], [0])).real
    return out, n_r, m_r


def normalize(data, order=None):
    """``_zmoments.py:344-356`` (no epsilon: zero vectors give NaN/inf as NumPy does)."""
    ax = _moment_axis(data)
    return data / np.linalg.norm(data, ord=order, axis=ax, keepdims=True)


def select(data, n, m, m_select):
    """``_zmoments.py:359-369``."""
    wanted = np.unique(np.abs(np.atleast_1d(m_select).ravel()))
    idx = np.where(np.isin(np.abs(m), wanted))[0]
    return np.take(data, idx, axis=_moment_axis(data)), n[idx], m[idx]


def unselect(data, n, m, m_unselect):
    """``_zmoments.py:371-374``."""
    drop = np.atleast_1d(m_unselect).ravel()
    keep = np.array([v for v in np.unique(np.abs(m)) if v not in drop])
    return select(data, n, m, keep)


def rotate(data, n, m, theta_deg):
    """``_zmoments.py:377-418``: complex moments times exp(-i m theta)."""
    zc, n_c, m_c = to_complex(data, n, m) if not np.iscomplexobj(data) else (data, n, m)
    fac = np.exp(-1j * np.deg2rad(theta_deg) * m_c)
    zc = zc * fac if zc.ndim == 2 else zc * fac[:, None, None]
    return zc, n_c, m_c


def rot_maps(data, n, m, n_folds, p=2, m_unselect=None):
    """``_zmoments.py:420-462``."""
    if m_unselect is None:
        m_unselect = (0, 1)
    elif 0 not in m_unselect:
        raise ValueError("m=0 must be included in m_unselect.")
    d, nn, mm = unselect(data, n, m, m_unselect)
    if p is not None:
        d = normalize(d, order=p)
    sq = d ** 2
    w = rot_maps_matrix(n_folds, mm)
    if data.ndim == 2:
        return np.dot(sq, w.T)
    return np.tensordot(w, sq, axes=([1], [0]))


def mirror_map(data, n, m, theta=None, p=2, m_unselect=(0, 1)):
    """``_zmoments.py:464-493``: max over theta of the mirror-symmetry score."""
    if theta is None:
        theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
    d, nn, mm = unselect(data, n, m, m_unselect)
    if p is not None:
        d = normalize(d, order=p)
    zc, n_c, m_c = to_complex(d, nn, mm)
    a, b = zc.real, zc.imag
    part1 = a ** 2 - b ** 2
    part2 = 2 * a * b
    cosmt = np.array([np.cos(k * t) for t in theta for k in m_c]).reshape(len(theta), -1)
    sinmt = np.array([np.sin(k * t) for t in theta for k in m_c]).reshape(len(theta), -1)
    table = np.hstack([cosmt, sinmt])
    if data.ndim == 2:
        return np.dot(np.hstack([part1, part2]), table.T).max(axis=1)
    return np.tensordot(table, np.vstack([part1, part2]), axes=([1], [0])).max(axis=0)
