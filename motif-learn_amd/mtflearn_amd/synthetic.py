"""Seeded synthetic STEM-like frames for tests and benchmarks.

The reference ships generators for this purpose (``mtflearn/datasets/_honeycomb_lattice.py:32-89,
169-227`` and ``_zps_test_data.py:62-65``) but its bundled notebook frames are absent from the
snapshot.  This module is our own generator with the same recipe -- honeycomb
lattice, bond length ``l`` px, Gaussian atoms of sigma ``l/4`` cut at 3 sigma, sub-lattice
intensities 1.0 / 0.5 -- plus optional Poisson-Gaussian noise, returned as float32 (the dtype
the reference's ``normalize_image`` hands to ``ZPs``, ``utils/_preprocessing_image.py:34``).
It is vectorised so 4096 x 4096 frames render in a few seconds.
"""
from __future__ import annotations

import numpy as np

__all__ = ["honeycomb_frame", "sliding_patches", "blob_patches"]


def _lattice_sites(h, w, l, angle_deg, shift):
    a1 = np.array([1.5 * l, np.sqrt(3.0) * l / 2.0])
    a2 = np.array([1.5 * l, -np.sqrt(3.0) * l / 2.0])
    reach = int(np.ceil(max(h, w) / l)) + 3
    idx = np.arange(-reach, reach + 1)
    n1, n2 = np.meshgrid(idx, idx, indexing="ij")
    cell = n1[..., None] * a1 + n2[..., None] * a2 + shift[0] * a1 + shift[1] * a2
    cell = cell.reshape(-1, 2)
    t = np.deg2rad(angle_deg)
    rot = np.array([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])
    centre = np.array([w / 2.0, h / 2.0])
    site_a = cell @ rot.T + centre
    site_b = (cell + np.array([l, 0.0])) @ rot.T + centre
    return site_a, site_b


def _stamp(img, sites, sigma, amp):
    h, w = img.shape
    rad = int(np.ceil(3 * sigma))
    keep = ((sites[:, 0] >= -rad) & (sites[:, 0] <= w - 1 + rad)
            & (sites[:, 1] >= -rad) & (sites[:, 1] <= h - 1 + rad))
    sites = sites[keep]
    if sites.size == 0:
        return
    off = np.arange(-rad - 1, rad + 2)
    oy, ox = np.meshgrid(off, off, indexing="ij")
    base = np.floor(sites).astype(np.int64)
    frac = sites - base
    for start in range(0, len(sites), 8192):
        b = base[start:start + 8192]
        f = frac[start:start + 8192]
        px = b[:, 0, None, None] + ox
        py = b[:, 1, None, None] + oy
        dx = ox - f[:, 0, None, None]
        dy = oy - f[:, 1, None, None]
        r2 = dx * dx + dy * dy
        val = amp * np.exp(-r2 / (2 * sigma * sigma))
        ok = (r2 <= (3 * sigma) ** 2) & (px >= 0) & (px < w) & (py >= 0) & (py < h)
        np.add.at(img, (py[ok], px[ok]), val[ok])


def honeycomb_frame(height, width=None, l=12.0, seed=0, angle=0.0, noise=True, dose=200.0,
                    read_sigma=0.02):
    """Float32 (height, width) frame: honeycomb lattice of Gaussian atoms (+ noise).

    ``noise=True`` applies Poisson counting noise at ``dose`` counts per unit intensity and
    additive Gaussian read-out noise, then rescales to [0, 1].
    """
    width = height if width is None else width
    rng = np.random.default_rng(seed)
    shift = rng.random(2)
    img = np.zeros((height, width), dtype=np.float64)
    site_a, site_b = _lattice_sites(height, width, float(l), angle, shift)
    sigma = l / 4.0
    _stamp(img, site_a, sigma, 1.0)
    _stamp(img, site_b, sigma, 0.5)
    if noise:
        img = rng.poisson(np.clip(img, 0, None) * dose) / dose
        img = img + rng.normal(0.0, read_sigma, img.shape)
        lo, hi = img.min(), img.max()
        img = (img - lo) / (hi - lo)
    return img.astype(np.float32)


def sliding_patches(frame, size, rows=None, cols=None):
    """All (or selected) un-padded ``size`` x ``size`` windows of ``frame`` as (N, size, size).

    Window ``(i, j)`` is ``frame[i:i+size, j:j+size]``; ordering is row-major over ``(i, j)``.
    """
    win = np.lib.stride_tricks.sliding_window_view(frame, (size, size))
    if rows is not None:
        win = win[np.asarray(rows)]
    if cols is not None:
        win = win[:, np.asarray(cols)]
    return np.ascontiguousarray(win.reshape(-1, size, size))


def blob_patches(size=32, n_fold=3, num_patches=8, seed=0, dtype=np.float32):
    """n-fold arrangements of Gaussian blobs at random rotations, one per patch, in [0, 1]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[:size, :size].astype(np.float64)
    c = size / 2.0
    s = size / 10.0
    out = np.empty((num_patches, size, size), dtype=np.float64)
    for k in range(num_patches):
        phase = rng.uniform(0, 2 * np.pi)
        img = np.exp(-((xx - c) ** 2 + (yy - c) ** 2) / (2 * s * s))
        for f in range(n_fold):
            ang = phase + 2 * np.pi * f / n_fold
            bx, by = c + size / 3.0 * np.cos(ang), c + size / 3.0 * np.sin(ang)
            img += np.exp(-((xx - bx) ** 2 + (yy - by) ** 2) / (2 * s * s))
        out[k] = img / img.max()
    return out.astype(dtype)
