"""Parameter pickers: the two routines that choose ``size`` and ``n_max`` for :class:`ZPs`, on the GPU.

Drop-ins for ``mtflearn.features.estimate_patch_size`` / ``radial_profile`` (reference
``mtflearn/features/_patch_size.py``), ``mtflearn.features.estimate_n_max`` / ``estimate_n_max_from_patch`` /
``_get_cumulative_energy`` (``mtflearn/features/_estimate_n_max.py``) and ``mtflearn.denoise.denoise_fft``
(``mtflearn/denoise/_denoise_fft.py``): same names, arguments, defaults, random-number draws (``np.random.randint``
in the reference's order, so ``np.random.seed(s)`` selects the same windows), error messages and return values.

What runs where.  The array work goes through ``libzernike_hip.so`` (``zk_autocorr_mean``, ``zk_polar_profile``,
``zk_power_spectra``, ``zk_denoise_fft``: hipFFT transforms with hand-written kernels around them, see
``csrc/zk_pickers.hip``); there is no CPU fallback.  The 1-D tails -- Gaussian smoothing + ``scipy.signal.find_peaks``
on a profile of a few hundred samples, cumulative sums, the median of the per-patch estimates -- are the reference's
own NumPy / SciPy calls on the host.

Parity status: ``_patch_size.py`` and ``_estimate_n_max.py`` import scikit-image, which is not installed
in the build image, and two of their steps ARE scikit-image calls -- ``skimage.transform.warp_polar`` (inside
``radial_profile``) and ``skimage.restoration.estimate_sigma``.  Those two are restated from scikit-image's
published algorithm (0.19-0.25: ``transform/_warps.py``, ``restoration/_denoise.py``; PyWavelets ``db2`` /
symmetric mode) and are **parity-unpinned**.  Everything else here is pinned by goldens captured from the
reference's own code (``oracle/make_golden_pickers.py`` -> ``tests/golden/pickers_golden.npz``: ``standardize_image``,
``autocorrelation``, ``find_highest_peak``, ``denoise_fft``, ``add_gaussian_noise``).  The reference's ``debug=True``
plots are not reproduced (UI, out of scope); the argument is accepted and ignored.
"""
from __future__ import annotations

from ctypes import POINTER, c_double, c_int32, c_void_p

import numpy as np
from scipy.ndimage import gaussian_filter1d
from scipy.signal import find_peaks

from .. import _native

__all__ = ["standardize_image", "autocorrelation", "radial_profile", "find_highest_peak", "estimate_patch_size",
           "denoise_fft", "estimate_sigma", "add_gaussian_noise", "get_ps", "estimate_n_max_from_patch",
           "estimate_n_max", "_get_cumulative_energy"]


def _device():
    import os
    import sys
    for var in ("MTFLEARN_AMD_DEVICE", "LOCAL_RANK"):
        if os.environ.get(var, "") != "":
            return int(os.environ[var])
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        return int(torch.cuda.current_device())
    return 0


def _operand(image):
    """C-contiguous float32 / float64 image (what NumPy's float64 promotion would compute on, see ZPs)."""
    image = np.asarray(image)
    if np.iscomplexobj(image):
        raise TypeError("complex images are not supported")
    if image.dtype not in (np.float32, np.float64):
        image = image.astype(np.float64)
    return np.ascontiguousarray(image)


def _origins(pairs):
    return np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))


def _call(code, what):
    if code != 0:
        msg = _native.last_error()
        # argument errors whose text is the reference's ValueError text keep their type
        if code == -10001:
            raise ValueError(msg)
        raise RuntimeError(f"{what} failed with code {code}: {msg}")


def _autocorr_mean(image, window, origins, standardize):
    lib = _native.load()
    if _native.device_count() == 0:
        raise RuntimeError("no HIP device visible: mtflearn_amd computes on MI355X only (there is no CPU fallback)")
    img = _operand(image)
    org = _origins(origins)
    out = np.empty((window, window), dtype=np.float64)
    _call(lib.zk_autocorr_mean(_device(), img.ctypes.data_as(c_void_p), _native.dtype_code(img.dtype), img.shape[0], img.shape[1],
                               window, org.ctypes.data_as(POINTER(c_int32)), org.shape[0], int(bool(standardize)),
                               out.ctypes.data_as(POINTER(c_double))), "zk_autocorr_mean")
    return out


# ------------------------------------------------------------------------------------------ _patch_size.py
def standardize_image(image):
    """Zero mean, unit (population) variance; reference ``_patch_size.py:9-19`` (host: one pass over the image)."""
    mean = np.mean(image)
    std = np.std(image)
    if std == 0:
        raise ValueError("Standard deviation is zero, can't standardize the image.")
    return (image - mean) / std


def autocorrelation(image, mode='same', method='fft', standardize=True):
    """Autocorrelation map of a 2-D image; reference ``_patch_size.py:22-46``
    (``scipy.signal.correlate(image, image, mode, method)`` of the standardised image).  On the GPU for the case the
    reference's callers use (``mode='same'``, square image); other modes are not implemented."""
    image = np.asarray(image)
    if image.ndim != 2 or image.shape[0] != image.shape[1] or mode != 'same':
        raise NotImplementedError("the device autocorrelation covers mode='same' on square 2-D images")
    return _autocorr_mean(image, image.shape[0], [(0, 0)], standardize)


def radial_profile(data, center=None, method="max"):
    """Radial profile of 2-D data through a polar resampling; reference ``_patch_size.py:48-100``
    (``skimage.transform.warp_polar(data, center, scaling='linear')`` aggregated over its 360 angles).
    ``center`` defaults to ``(h // 2, w // 2)`` (fftshift convention)."""
    methods = {"mean": 0, "max": 1, "sum": 2}
    if method not in methods:
        raise ValueError(f"Invalid method '{method}'. Must be 'mean', 'max', or 'sum'.")
    data = np.ascontiguousarray(data, dtype=np.float64)
    if data.ndim != 2:
        raise ValueError("radial_profile needs 2-D data")
    h, w = data.shape
    ci, cj = (h // 2, w // 2) if center is None else (int(center[0]), int(center[1]))
    lib = _native.load()
    if _native.device_count() == 0:
        raise RuntimeError("no HIP device visible: mtflearn_amd computes on MI355X only (there is no CPU fallback)")
    out = np.empty(int(lib.zk_polar_radii(h, w)), dtype=np.float64)
    _call(lib.zk_polar_profile(_device(), data.ctypes.data_as(POINTER(c_double)), 1, h, w, ci, cj, methods[method],
                               out.ctypes.data_as(POINTER(c_double))), "zk_polar_profile")
    return out


def find_highest_peak(radial_profile, min_distance=5, prominence_factor=0.15,
                      min_width=2, smooth_sigma=1.0, max_distance=None, debug=False):
    """Highest peak of a radial profile; reference ``_patch_size.py:102-218`` (the same SciPy calls: Gaussian
    smoothing, ``find_peaks`` with prominence / width / distance criteria).  Returns
    ``(highest_peak, all_peaks, properties)``; ``(None, None, properties)`` when nothing qualifies."""
    search_profile = radial_profile[min_distance:]
    if max_distance is not None:
        search_profile = search_profile[:max_distance - min_distance]
    smoothed = gaussian_filter1d(search_profile, sigma=smooth_sigma)
    profile_range = np.ptp(smoothed)
    min_prominence = prominence_factor * profile_range
    peaks, properties = find_peaks(smoothed, prominence=min_prominence, width=min_width, distance=3)
    if len(peaks) == 0:
        return None, None, properties
    highest_idx = np.argmax(smoothed[peaks])
    return peaks[highest_idx] + min_distance, peaks + min_distance, properties


def estimate_patch_size(img, window_size=None, standardize=True, n_samples=None,
                        min_distance=5, prominence_factor=0.15,
                        min_width=2, smooth_sigma=1.0, radial_method='max', debug=False):
    """Lattice spacing of an image from the mean autocorrelation of random windows; reference
    ``_patch_size.py:221-302``.  Returns the radius (pixels) of the highest peak of the radial profile, or None."""
    img = np.asarray(img)
    h, w = img.shape
    if window_size is None:
        window_size = h // 2
    if n_samples is None:
        n_samples = min(100, (h // window_size) * (w // window_size))
    if n_samples == 0:
        raise ValueError(f"Window size {window_size} is too large for image of size {img.shape}")
    origins = []
    for _ in range(n_samples):                           # the reference's draws, in the reference's order
        y = np.random.randint(0, h - window_size)
        x = np.random.randint(0, w - window_size)
        origins.append((y, x))
    autocorr_mean = _autocorr_mean(img, window_size, origins, standardize)
    line = radial_profile(autocorr_mean, method=radial_method)
    peak, _, _ = find_highest_peak(line, min_distance=min_distance, prominence_factor=prominence_factor,
                                   min_width=min_width, smooth_sigma=smooth_sigma, max_distance=len(line), debug=False)
    return peak


# ------------------------------------------------------------------------------------------ _denoise_fft.py
def denoise_fft(image, p):
    """Keep the top ``p`` fraction of Fourier coefficients by power; reference ``denoise/_denoise_fft.py:4-47``."""
    if not isinstance(image, np.ndarray):
        raise TypeError("Input image must be a numpy array.")
    if image.ndim != 2:
        raise ValueError("Input image must be a 2D array.")
    if not (0 < p <= 1):
        raise ValueError("Fraction p must be between 0 and 1.")
    lib = _native.load()
    if _native.device_count() == 0:
        raise RuntimeError("no HIP device visible: mtflearn_amd computes on MI355X only (there is no CPU fallback)")
    img = _operand(image)
    out = np.empty(img.shape, dtype=np.float64)
    _call(lib.zk_denoise_fft(_device(), img.ctypes.data_as(c_void_p), _native.dtype_code(img.dtype), img.shape[0], img.shape[1],
                             float(p), out.ctypes.data_as(POINTER(c_double))), "zk_denoise_fft")
    return out


# ------------------------------------------------------------------------------------------ _estimate_n_max.py
def estimate_sigma(image):
    """Robust wavelet estimate of the Gaussian noise standard deviation of a 2-D image:
    ``skimage.restoration.estimate_sigma`` (median absolute ``db2`` diagonal detail coefficient / 0.6745), on the device
    (``zk_wavelet_sigma``).  Restated from scikit-image / PyWavelets (not installed here): **parity-unpinned**; it only
    decides which branch ``estimate_n_max`` takes (``sigma > t``)."""
    lib = _native.load()
    if _native.device_count() == 0:
        raise RuntimeError("no HIP device visible: mtflearn_amd computes on MI355X only (there is no CPU fallback)")
    img = _operand(image)
    if img.ndim != 2:
        raise ValueError("estimate_sigma needs a 2-D image")
    from ctypes import byref
    sigma = c_double()
    _call(lib.zk_wavelet_sigma(_device(), img.ctypes.data_as(c_void_p), _native.dtype_code(img.dtype), img.shape[0], img.shape[1],
                               byref(sigma)), "zk_wavelet_sigma")
    return float(sigma.value)


def add_gaussian_noise(img, sigma=0.1, seed=None):
    """Zero-mean Gaussian noise; reference ``datasets/_noise_models.py:40-66``."""
    img = np.asarray(img, dtype=np.float32)
    if sigma < 0:
        raise ValueError("sigma must be non-negative.")
    rng = np.random.default_rng(seed)
    noise = rng.normal(0.0, sigma, size=img.shape).astype(np.float32)
    return img + noise


def _window_1d(window_type, size):
    if window_type is None:
        return None
    if window_type in ('hann', 'hanning'):
        return np.hanning(size)
    if window_type == 'hamming':
        return np.hamming(size)
    if window_type == 'blackman':
        return np.blackman(size)
    if window_type == 'tukey':
        from scipy.signal import windows
        return windows.tukey(size, alpha=0.5)
    raise ValueError(f"Unknown window type: {window_type}")


def _power_spectra(image, size, origins, window_type):
    lib = _native.load()
    if _native.device_count() == 0:
        raise RuntimeError("no HIP device visible: mtflearn_amd computes on MI355X only (there is no CPU fallback)")
    img = _operand(image)
    org = _origins(origins)
    win = _window_1d(window_type, size)
    win_p = None if win is None else np.ascontiguousarray(win, dtype=np.float64).ctypes.data_as(POINTER(c_double))
    out = np.empty((org.shape[0], size, size), dtype=np.float64)
    _call(lib.zk_power_spectra(_device(), img.ctypes.data_as(c_void_p), _native.dtype_code(img.dtype), img.shape[0], img.shape[1],
                               size, org.ctypes.data_as(POINTER(c_int32)), org.shape[0], win_p,
                               out.ctypes.data_as(POINTER(c_double))), "zk_power_spectra")
    return out


def _radial_profiles(stack):
    """``radial_profile(power)`` (default method 'max') of every map of a ``(n, s, s)`` stack in one call."""
    lib = _native.load()
    stack = np.ascontiguousarray(stack, dtype=np.float64)
    n, h, w = stack.shape
    out = np.empty((n, int(lib.zk_polar_radii(h, w))), dtype=np.float64)
    _call(lib.zk_polar_profile(_device(), stack.ctypes.data_as(POINTER(c_double)), n, h, w, -1, -1, 1,
                               out.ctypes.data_as(POINTER(c_double))), "zk_polar_profile")
    return out


def _cumulative_from_profile(profile, normalize, epsilon):
    weighted_power = profile * np.arange(len(profile))
    cumulative_energy = np.cumsum(weighted_power)
    if normalize:
        total_energy = cumulative_energy[-1]
        if total_energy > epsilon:
            cumulative_energy = cumulative_energy / total_energy
        else:
            cumulative_energy = np.zeros_like(cumulative_energy)
    return cumulative_energy


def _get_cumulative_energy(patch, window_type='hann', normalize=True,
                           return_profile=False, epsilon=1e-10):
    """Cumulative radial energy of a patch's windowed power spectrum; reference ``_estimate_n_max.py:8-86``."""
    patch = np.asarray(patch)
    size = patch.shape[0]
    power = _power_spectra(patch, size, [(0, 0)], window_type)
    profile = _radial_profiles(power)[0]
    cumulative_energy = _cumulative_from_profile(profile, normalize, epsilon)
    if return_profile:
        return cumulative_energy, profile
    return cumulative_energy


def _n_max_from_curves(l_noise, l_clean, size):
    l = l_clean - l_noise
    return min(max(12, np.argmax(l) * 2), size // 2)


def estimate_n_max_from_patch(patch, p=0.01):
    """Reference ``_estimate_n_max.py:88-94``."""
    patch = np.asarray(patch)
    patch_denoised = denoise_fft(patch, p=p)
    return _n_max_from_curves(_get_cumulative_energy(patch), _get_cumulative_energy(patch_denoised), patch.shape[0])


def _draw_origins(h, w, n_samples, patch_size):
    out = []
    for _ in range(n_samples):                           # the reference's draws (get_ps), in the reference's order
        y = np.random.randint(0, h - patch_size)
        x = np.random.randint(0, w - patch_size)
        out.append((y, x))
    return out


def get_ps(img, n_samples, patch_size):
    """Random patches of an image; reference ``_estimate_n_max.py:96-105`` (host slicing: it only returns views)."""
    img = np.asarray(img)
    h, w = img.shape
    return np.array([img[y:y + patch_size, x:x + patch_size] for y, x in _draw_origins(h, w, n_samples, patch_size)])


def estimate_n_max(img, patch_size, n_samples=50, p=0.01, t=0.01):
    """Radial order up to which the patches of an image carry signal; reference ``_estimate_n_max.py:108-125``:
    noise level -> FFT-denoised twin of the image (or, for a clean image, a noisy twin) -> random patches of both ->
    the radius at which the denoised patch's cumulative spectral energy leads the noisy one's by most -> median."""
    img = np.asarray(img)
    sigma = estimate_sigma(img)
    if sigma > t:
        img_denoised = denoise_fft(img, p=p)
    else:
        img_denoised = img.copy()
        img = add_gaussian_noise(img_denoised, sigma=0.3)
    h, w = img.shape
    org_noisy = _draw_origins(h, w, n_samples, patch_size)          # two independent draws, as in the reference
    org_clean = _draw_origins(h, w, n_samples, patch_size)
    prof_noisy = _radial_profiles(_power_spectra(img, patch_size, org_noisy, 'hann'))
    prof_clean = _radial_profiles(_power_spectra(img_denoised, patch_size, org_clean, 'hann'))
    n_max_list = []
    for a, b in zip(prof_noisy, prof_clean):
        n_max_list.append(_n_max_from_curves(_cumulative_from_profile(a, True, 1e-10),
                                             _cumulative_from_profile(b, True, 1e-10), patch_size))
    return np.median(n_max_list)
