"""``ZPs`` -- Zernike-moment transformer whose ``transform`` runs on MI355X.

Drop-in for ``mtflearn.features.ZPs`` (reference ``mtflearn/features/_zps.py:11-197``): same
constructor, attributes (``n_max``, ``size``, ``n``, ``m``, ``polynomials``), methods, sklearn
behaviour (``get_params`` / ``clone`` / ``repr``), warning and error messages, output container
(:class:`zmoments`) and output dtype (float64).  What differs is where the arithmetic happens:

* 3-D input ``(N, size, size)``  -> ``zk_transform_patches``  (reference ``_zps.py:146-157``, a GEMM)
* 2-D input ``(H, W)``           -> ``zk_transform_frame``    (reference ``_zps.py:159-193``, FFT
  convolution) -- here an exact direct sum, so it does not carry the single-precision FFT noise
  the reference has on float32 images.

Both are hand-written HIP kernels behind the C ABI of ``include/zernike_hip.h``; there is no CPU
fallback -- without a HIP device ``transform`` raises ``RuntimeError``.  The basis itself is built
on the host exactly as the reference builds it (``_zps.py:52-90``), so ``polynomials`` is
bit-identical.

Device selection, at first use: ``ZPs.to_device(i)`` if it was called, else the environment variable
``MTFLEARN_AMD_DEVICE``, else ``LOCAL_RANK`` (one process per GPU under a launcher), else -- when PyTorch is
already imported and has initialised HIP -- ``torch.cuda.current_device()``, else 0.

Thread safety: a ``ZPs`` object shares one device plan (staging buffers, one stream); calls on the same object
are serialised by a lock (the reference is stateless NumPy and needs none).  ``release()`` frees the plan's
staging memory.  Large results are page-locked arrays out of a recycling pool (``_native.PinnedPool``).
"""
from __future__ import annotations

import os
import sys
import threading
import warnings

import numpy as np
from scipy.special import factorial
from sklearn.base import BaseEstimator, TransformerMixin

from .. import _native
from .moments import zmoments

__all__ = ["ZPs"]


def _radial_terms(n, abs_m):
    """[(coefficient, power)] of R_n^{|m|}: coefficient (-1)^k (n-k)! / (k! a! b!), power n-2k."""
    half_sum, half_diff = (n + abs_m) // 2, (n - abs_m) // 2
    terms = []
    for k in range(half_diff + 1):
        upper = (-1) ** k * factorial(n - k)
        lower = factorial(k) * factorial(half_sum - k) * factorial(half_diff - k)
        terms.append((upper / lower, n - 2 * k))
    return terms


class ZPs(BaseEstimator, TransformerMixin):
    """Zernike polynomial basis of radial order <= ``n_max`` on a ``size`` x ``size`` grid.

    Parameters
    ----------
    n_max : int
        Maximum radial order.
    size : int
        Side of the polynomial grid / of the patches the moments are taken over.
    """

    def __init__(self, n_max: int, size: int):
        if n_max < 0:
            raise ValueError("n_max must be non-negative.")
        if size <= 0:
            raise ValueError("size must be positive.")
        if n_max > size:
            raise ValueError(
                f"n_max={n_max} exceeds size={size}. This will produce "
                f"meaningless results. Use n_max <= {size//2} for accurate moments.")
        if n_max > size / 2:
            warnings.warn(
                f"n_max={n_max} exceeds recommended limit of size/2≈{size // 2}. "
                f"High-order Zernike moments may suffer from aliasing and numerical "
                f"errors. For accurate results, use n_max <= {size//2}; for maximum "
                f"stability, use n_max <= {size // 2}.",
                UserWarning, stacklevel=2)
        self.n_max = n_max
        self.size = size
        self.n, self.m, self.polynomials = self._generate_polynomials()
        self._plan = None
        self._device = None
        self._lock = threading.RLock()

    # ------------------------------------------------------------------ basis (host)
    def _generate_polynomials(self):
        """(n, m, V): V[j] = R_n^{|m|}(rho) sqrt(2(n+1)/(1+[m==0])) {cos(m t) | sin(|m| t)} on
        ``linspace(-1, 1, size)^2``, zero outside rho <= 1; n ascending, then m ascending.
        Same floating-point operations, in the same order, as reference ``_zps.py:52-90``."""
        grid = np.linspace(-1, 1, self.size)
        xs, ys = np.meshgrid(grid, grid)
        rho = np.sqrt(xs ** 2 + ys ** 2)
        theta = np.arctan2(ys, xs)
        inside = rho <= 1
        radial_cache = {}
        orders, freqs, stack = [], [], []
        for n in range(self.n_max + 1):
            for m in range(-n, n + 1, 2):
                key = (n, abs(m))
                if key not in radial_cache:
                    radial = np.zeros_like(rho)
                    for coef, power in _radial_terms(n, abs(m)):
                        radial += coef * rho ** power
                    radial_cache[key] = radial
                scale = np.sqrt(2 * (n + 1) / (1 + (m == 0)))
                masked = np.where(inside, radial_cache[key] * scale, 0)
                angular = np.sin(-m * theta) if m < 0 else np.cos(m * theta)
                orders.append(n)
                freqs.append(m)
                stack.append(masked * angular)
        return np.array(orders), np.array(freqs), np.array(stack)

    def get_polynomials(self) -> np.ndarray:
        return self.polynomials

    # ------------------------------------------------------------------ sklearn surface
    def fit(self, X, y=None):
        """No-op (stateless transformer), kept for sklearn pipelines."""
        return self

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)

    # ------------------------------------------------------------------ device plan
    def _pick_device(self):
        if getattr(self, "_device", None) is not None:
            return self._device
        for var in ("MTFLEARN_AMD_DEVICE", "LOCAL_RANK"):
            if os.environ.get(var, "") != "":
                return int(os.environ[var])
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
            return int(torch.cuda.current_device())
        return 0

    def to_device(self, index):
        """Bind this transformer to GPU ``index`` (drops a plan created on another device)."""
        with self._lock:
            if self._plan is not None and self._plan.device != int(index):
                self._plan.close()
                self._plan = None
            self._device = int(index)
        return self

    def _device_plan(self):
        with self._lock:
            if self._plan is None:
                self._plan = _native.Plan(self.polynomials, self.n, self.m, device=self._pick_device())
            return self._plan

    def release(self):
        """Free the device staging buffers this object's calls have grown (they are re-created on demand)."""
        with self._lock:
            if self._plan is not None:
                self._plan.release_staging()

    def __getstate__(self):
        state = dict(super().__getstate__())
        state["_plan"] = None  # device handles do not pickle / deepcopy
        state.pop("_lock", None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._lock = threading.RLock()

    @staticmethod
    def _device_operand(images):
        """C-contiguous array in one of the element types the host entry points take.

        float32 / float64 pass as they are.  The usual detector formats -- uint8, uint16, int16 (and bool / int8,
        re-labelled or widened to those) -- are exact in float32, which is what NumPy's promotion to float64 would
        compute on: they cross PCIe as they are (1-2 bytes per pixel instead of 4, and no host-side ``astype`` pass)
        and are widened on the device.  float16 goes to float32, wider integers to float64."""
        if np.iscomplexobj(images):
            raise TypeError("complex images are not supported by the HIP kernels")
        dt = images.dtype
        if dt in (np.float32, np.float64, np.uint8, np.uint16, np.int16):
            pass
        elif dt == np.bool_:
            images = images.view(np.uint8) if images.flags.c_contiguous else images.astype(np.uint8)
        elif dt == np.int8:
            images = images.astype(np.int16)
        elif dt == np.float16:
            images = images.astype(np.float32)
        else:
            images = images.astype(np.float64)
        return np.ascontiguousarray(images)

    # ------------------------------------------------------------------ transform
    def transform(self, images) -> zmoments:
        """Zernike moments of a batch of patches (3-D input) or of every ``size`` x ``size``
        window of a frame (2-D input, zero-padded 'same' alignment)."""
        images = np.asarray(images)
        if images.ndim == 2:
            return self._transform_frame(images)
        if images.ndim == 3:
            return self._transform_patches(images)
        raise ValueError("Images must be 2D or 3D array.")

    def transform_at(self, image, points) -> zmoments:
        """Moments of the ``size`` x ``size`` windows centred on key points of ``image`` (extension).

        Same numbers as ``self.transform(KeyPoints(points, image, size).extract_patches())`` with the
        reference's ``features/_keypoint.py:60-78`` -- window ``image[y-s1:y+s2, x-s1:x+s2]`` for the
        rounded point ``(x, y)``, ``s1 = size // 2`` -- but the ``(N, size, size)`` batch is never built:
        the kernel reads the windows from the frame resident on the GPU.  Pixels outside the frame count
        as zero (the reference's ``KeyPoints`` drops border points before extracting)."""
        image = np.asarray(image)
        if image.ndim != 2:
            raise ValueError("transform_at needs a 2D image.")
        pts = np.rint(np.asarray(points, dtype=np.float64)).astype(np.int64).reshape(-1, 2)
        if pts.shape[0] == 0:
            return zmoments(np.empty((0, len(self.n))), self.n, self.m, patch_size=self.size)
        operand = self._device_operand(image)
        with self._lock:
            data = self._device_plan().transform_points(operand, pts)  # (plans without the key-point kernel gather on the device)
        return zmoments._adopt(data=data, n=self.n, m=self.m, patch_size=self.size)

    def transform_grid(self, image, step=1) -> zmoments:
        """Moments of the un-padded ``size`` x ``size`` windows taken every ``step`` pixels (extension).

        Same numbers as ``self.transform(extract_patches(image, size, step))`` with the reference's strided
        extractor (``denoise/_denoise_svd.py:15-49``): window origins ``0, step, 2 step, ...`` along each
        axis plus the last admissible origin ``extent - size`` when the stride does not land on it, windows
        ordered row-major over (row origin, column origin).  No ``(N, size, size)`` batch is materialised:
        the windows are read from the frame resident on the GPU (``transform_at``)."""
        image = np.asarray(image)
        if image.ndim != 2:
            raise ValueError("transform_grid needs a 2D image.")
        height, width = image.shape
        if height < self.size or width < self.size:
            raise ValueError(
                f"For FFT convolution, image size ({height}x{width}) must be at least "
                f"as large as polynomial size ({self.size}x{self.size})")
        if step < 1:
            raise ValueError("step must be a positive integer.")

        def origins(extent):
            last = extent - self.size
            idx = np.arange(0, last, step)
            if idx.size == 0 or idx[-1] != last:
                idx = np.append(idx, last)
            return idx

        rows, cols = origins(height), origins(width)
        half = self.size // 2
        yy, xx = np.meshgrid(rows + half, cols + half, indexing="ij")
        return self.transform_at(image, np.column_stack([xx.ravel(), yy.ravel()]))

    def symmetry_maps(self, image, n_folds=(2, 3, 4, 6), p=2, m_unselect=(0, 1), theta=None,
                      abs_moments=True, mirror=True):
        """Frame -> symmetry maps on the GPU, one fused kernel up to n_max 16 (extension; not in the reference API).

        Equivalent to ``zm = self.transform(image)`` followed by ``zm.rot_maps(n_folds, p, m_unselect)``,
        ``np.abs(zm.to_complex().data)`` and ``zm.mirror_map(theta, p, m_unselect)`` (reference
        ``_zmoments.py:300-316, 420-493``), but the ``(N_poly, H, W)`` moments never leave the GPU (n_max 17-24:
        they pass through a device scratch matrix, a row band at a time).
        Returns a dict with ``rot_maps (len(n_folds), H, W)``, ``abs (N_c, H, W)`` with its ``abs_n`` /
        ``abs_m`` labels, ``mirror_map (H, W)`` (entries present only when requested) and ``valid_mask``.
        Falls back to the unfused composition (device transform + the container's NumPy methods) for
        shapes or options the fused kernel does not cover (``p`` other than 2 / None, more than 8 folds,
        n_max > 24, shapes for which the separable tables are unavailable)."""
        image = np.asarray(image)
        if image.ndim != 2:
            raise ValueError("symmetry_maps needs a 2D image.")
        height, width = image.shape
        if height < self.size or width < self.size:
            raise ValueError(
                f"For FFT convolution, image size ({height}x{width}) must be at least "
                f"as large as polynomial size ({self.size}x{self.size})")
        if m_unselect is None:
            m_unselect = (0, 1)
        if 0 not in m_unselect:
            raise ValueError("m=0 must be included in m_unselect.")
        folds = [] if n_folds is None else list(np.atleast_1d(n_folds).ravel())
        if mirror and theta is None:
            theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
        complex_n = np.array([n for n in range(self.n_max + 1) for _ in range(n % 2, n + 1, 2)])
        complex_m = np.array([m for n in range(self.n_max + 1) for m in range(n % 2, n + 1, 2)])
        operand = self._device_operand(image)
        plan = self._device_plan()
        code = _native.dtype_code(operand.dtype)
        if code not in (_native.ZK_F32, _native.ZK_F64):
            code = _native.ZK_F32          # detector formats are widened to float32 on the device
        fused = (plan.supports(_native.OP_MAPS, code) and p in (2, None) and len(folds) <= 8
                 and all(int(f) == f and f > 0 for f in folds))
        out = {}
        if fused:
            with self._lock:
                rot, ab, mir = plan.frame_maps(operand, len(complex_n), folds=folds, m_unselect=m_unselect, p=p,
                                               theta=theta if mirror else None, want_abs=abs_moments)
        else:
            zm = self._transform_frame(image)
            rot = zm.rot_maps(folds, p=p, m_unselect=m_unselect) if folds else None
            ab = np.abs(zm.to_complex().data) if abs_moments else None
            mir = zm.mirror_map(theta=theta, p=p, m_unselect=m_unselect) if mirror else None
        if rot is not None:
            out["rot_maps"] = rot
        if ab is not None:
            out["abs"], out["abs_n"], out["abs_m"] = ab, complex_n, complex_m
        if mir is not None:
            out["mirror_map"] = mir
        out["valid_mask"] = self._valid_mask(height, width)   # same convention as zmoments.valid_mask
        return out

    def _rows_options(self, n_folds, p, m_unselect, theta, mirror):
        if m_unselect is None:
            m_unselect = (0, 1)
        if 0 not in m_unselect:
            raise ValueError("m=0 must be included in m_unselect.")
        folds = [] if n_folds is None else list(np.atleast_1d(n_folds).ravel())
        if mirror and theta is None:
            theta = np.linspace(0, 2 * np.pi, 360, endpoint=False)
        complex_n = np.array([n for n in range(self.n_max + 1) for _ in range(n % 2, n + 1, 2)])
        complex_m = np.array([m for n in range(self.n_max + 1) for m in range(n % 2, n + 1, 2)])
        fused = (self._device_plan().supports(_native.OP_MAPS, _native.ZK_F64) and p in (2, None) and len(folds) <= 8
                 and all(int(f) == f and f > 0 for f in folds))
        return folds, m_unselect, theta if mirror else None, complex_n, complex_m, fused

    @staticmethod
    def _rows_result(rot, ab, mir, complex_n, complex_m):
        out = {}
        if rot is not None:
            out["rot_maps"] = rot
        if ab is not None:
            out["abs"], out["abs_n"], out["abs_m"] = ab, complex_n, complex_m
        if mir is not None:
            out["mirror_map"] = mir
        return out

    def symmetry_of(self, moments, n_folds=(2, 3, 4, 6), p=2, m_unselect=(0, 1), theta=None, abs_moments=True, mirror=True):
        """Symmetry scores of a BATCH of moment vectors on the GPU (extension): ``moments`` is a rank-2 ``zmoments`` of this
        set or an ``(N, n_poly)`` array.  Equivalent to ``zm.rot_maps(n_folds, p, m_unselect)`` -> ``(N, len(n_folds))``,
        ``np.abs(zm.to_complex().data)`` -> ``(N, N_c)`` and ``zm.mirror_map(theta, p, m_unselect)`` -> ``(N,)``
        (reference ``_zmoments.py:300-316, 420-493`` on rank-2 data), one lane per row."""
        data = np.asarray(moments.data if isinstance(moments, zmoments) else moments, dtype=np.float64)
        if data.ndim != 2 or data.shape[1] != len(self.n):
            raise ValueError(f"symmetry_of needs an (N, {len(self.n)}) matrix of real Zernike moments.")
        folds, m_unselect, theta, cn, cm, fused = self._rows_options(n_folds, p, m_unselect, theta, mirror)
        if data.shape[0] == 0 or not fused:
            zm = zmoments(data, self.n, self.m, patch_size=self.size)
            return self._rows_result(zm.rot_maps(folds, p=p, m_unselect=m_unselect) if folds else None,
                                     np.abs(zm.to_complex().data) if abs_moments else None,
                                     zm.mirror_map(theta=theta, p=p, m_unselect=m_unselect) if mirror else None, cn, cm)
        with self._lock:
            rot, ab, mir = self._device_plan().moment_maps(data, len(cn), folds=folds, m_unselect=m_unselect, p=p, theta=theta,
                                                           want_abs=abs_moments)
        return self._rows_result(rot, ab, mir, cn, cm)

    def symmetry_at(self, image, points, n_folds=(2, 3, 4, 6), p=2, m_unselect=(0, 1), theta=None, abs_moments=True,
                    mirror=True):
        """Symmetry scores of the windows centred on key points (extension) = ``symmetry_of(transform_at(image, points))``
        -- the reference's notebook flow ``KeyPoints.extract_patches`` -> ``ZPs.transform`` -> ``rot_maps`` -- with neither
        the patch batch nor the ``(N, n_poly)`` moment matrix leaving the GPU."""
        image = np.asarray(image)
        if image.ndim != 2:
            raise ValueError("symmetry_at needs a 2D image.")
        pts = np.rint(np.asarray(points, dtype=np.float64)).astype(np.int64).reshape(-1, 2)
        folds, m_unselect, theta, cn, cm, fused = self._rows_options(n_folds, p, m_unselect, theta, mirror)
        if pts.shape[0] == 0 or not fused:
            return self.symmetry_of(self.transform_at(image, pts), n_folds=n_folds, p=p, m_unselect=m_unselect, theta=theta,
                                    abs_moments=abs_moments, mirror=mirror)
        operand = self._device_operand(image)
        with self._lock:
            rot, ab, mir = self._device_plan().points_maps(operand, pts, len(cn), folds=folds, m_unselect=m_unselect, p=p,
                                                           theta=theta, want_abs=abs_moments)
        return self._rows_result(rot, ab, mir, cn, cm)

    def _valid_mask(self, height, width):
        head = (self.size - 1) // 2
        tail = self.size - 1 - head
        mask = np.ones((height, width), dtype=bool)
        mask[:head, :] = False
        mask[-tail:, :] = False
        mask[:, :head] = False
        mask[:, -tail:] = False
        return mask

    def _transform_patches(self, images):
        _, height, width = images.shape
        if height != self.size or width != self.size:
            raise ValueError(
                f"For batch processing, image size ({height}x{width}) must match "
                f"polynomial size ({self.size}x{self.size})")
        if images.shape[0] == 0:
            data = np.empty((0, len(self.n)), dtype=np.float64)
        else:
            operand = self._device_operand(images)
            with self._lock:
                data = self._device_plan().transform_patches(operand)
        return zmoments._adopt(data=data, n=self.n, m=self.m, patch_size=self.size)

    def _transform_frame(self, image):
        height, width = image.shape
        if height < self.size or width < self.size:
            raise ValueError(
                f"For FFT convolution, image size ({height}x{width}) must be at least "
                f"as large as polynomial size ({self.size}x{self.size})")
        operand = self._device_operand(image)
        with self._lock:
            data = self._device_plan().transform_frame(operand)
        return zmoments._adopt(data=data, n=self.n, m=self.m, patch_size=self.size)
