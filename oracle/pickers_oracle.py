"""CPU restatement of the reference's parameter pickers -- TEST INFRASTRUCTURE, never imported by the product.

Follows ``/root/reference/mtflearn/features/_patch_size.py``, ``_estimate_n_max.py`` and
``denoise/_denoise_fft.py`` line by line (citations per function), with NumPy / SciPy only.

Pinning.  ``standardize_image``, ``autocorrelation``, ``find_highest_peak``, ``denoise_fft`` and
``add_gaussian_noise`` are checked against goldens captured from the reference itself
(``oracle/make_golden_pickers.py`` -> ``tests/golden/pickers_golden.npz``; ``tests/test_oracle_golden.py``).
``warp_polar`` (inside ``radial_profile``) and ``estimate_sigma`` are scikit-image functions; scikit-image is not
installed in the build image (``import skimage`` -> ModuleNotFoundError, an absent dependency, not a refusal), its
version is unpinned in the reference's ``pyproject.toml``, so both are restated here from the published algorithm
(scikit-image 0.19-0.25 ``transform/_warps.py``, ``transform/_warps_cy.pyx``, ``restoration/_denoise.py``;
PyWavelets ``dwtn`` with ``db2`` / mode 'symmetric'): **parity unpinned** for these two and for whatever is
composed from them (``radial_profile``, ``_get_cumulative_energy``, ``estimate_patch_size``, ``estimate_n_max``).
"""
import numpy as np
from scipy.ndimage import gaussian_filter1d
from scipy.signal import correlate, find_peaks


def standardize_image(image):
    """``_patch_size.py:9-19``."""
    mean = np.mean(image)
    std = np.std(image)
    if std == 0:
        raise ValueError("Standard deviation is zero, can't standardize the image.")
    return (image - mean) / std


def autocorrelation(image, mode='same', method='fft', standardize=True):
    """``_patch_size.py:22-46``."""
    if standardize:
        image = standardize_image(image)
    return correlate(image, image, mode=mode, method=method)


def warp_polar_linear(data, center):
    """``skimage.transform.warp_polar(data, center=center, scaling='linear')`` for 2-D float data: output
    (360, ceil(radius)) with radius = sqrt((h/2)^2 + (w/2)^2); bilinear (order 1), mode='constant', cval=0,
    clip=True.  Restated, see the module docstring."""
    data = np.asarray(data, dtype=np.float64)
    h, w = data.shape
    radius = np.sqrt((h / 2) ** 2 + (w / 2) ** 2)
    width = int(np.ceil(radius))
    k_angle = 360 / (2 * np.pi)
    k_radius = width / radius
    cols, rows = np.meshgrid(np.arange(width), np.arange(360))            # output (row = angle, col = radius)
    angle = rows / k_angle
    rr = (cols / k_radius) * np.sin(angle) + center[0]
    cc = (cols / k_radius) * np.cos(angle) + center[1]
    r0, c0 = np.floor(rr).astype(np.int64), np.floor(cc).astype(np.int64)
    r1, c1 = np.ceil(rr).astype(np.int64), np.ceil(cc).astype(np.int64)
    dr, dc = rr - r0, cc - c0

    def px(r, c):
        inside = (r >= 0) & (r < h) & (c >= 0) & (c < w)
        return np.where(inside, data[np.clip(r, 0, h - 1), np.clip(c, 0, w - 1)], 0.0)

    top = (1 - dc) * px(r0, c0) + dc * px(r0, c1)
    bottom = (1 - dc) * px(r1, c0) + dc * px(r1, c1)
    out = (1 - dr) * top + dr * bottom
    lo, hi = data.min(), data.max()                                       # _clip_warp_output
    preserve_cval = not (lo <= 0 <= hi)
    cval_mask = out == 0
    out = np.clip(out, lo, hi)
    if preserve_cval:
        out[cval_mask] = 0
    return out


def radial_profile(data, center=None, method="max"):
    """``_patch_size.py:48-100``."""
    h, w = data.shape
    if center is None:
        center = (h // 2, w // 2)
    polar_image = warp_polar_linear(data, center)
    if method == "mean":
        return np.mean(polar_image, axis=0)
    if method == "max":
        return np.max(polar_image, axis=0)
    if method == "sum":
        return np.sum(polar_image, axis=0)
    raise ValueError(f"Invalid method '{method}'. Must be 'mean', 'max', or 'sum'.")


def find_highest_peak(profile, min_distance=5, prominence_factor=0.15, min_width=2, smooth_sigma=1.0, max_distance=None):
    """``_patch_size.py:102-218`` without the plotting branch."""
    search_profile = profile[min_distance:]
    if max_distance is not None:
        search_profile = search_profile[:max_distance - min_distance]
    smoothed = gaussian_filter1d(search_profile, sigma=smooth_sigma)
    min_prominence = prominence_factor * np.ptp(smoothed)
    peaks, properties = find_peaks(smoothed, prominence=min_prominence, width=min_width, distance=3)
    if len(peaks) == 0:
        return None, None, properties
    highest_idx = np.argmax(smoothed[peaks])
    return peaks[highest_idx] + min_distance, peaks + min_distance, properties


def autocorr_mean(img, window_size, origins, standardize=True):
    """The sampling loop of ``estimate_patch_size`` (``_patch_size.py:268-279``) for given window origins."""
    maps = [autocorrelation(img[y:y + window_size, x:x + window_size], standardize=standardize) for y, x in origins]
    return np.mean(maps, axis=0)


def estimate_patch_size(img, window_size=None, standardize=True, n_samples=None, min_distance=5, prominence_factor=0.15,
                        min_width=2, smooth_sigma=1.0, radial_method='max'):
    """``_patch_size.py:221-302`` (same np.random draws)."""
    h, w = img.shape
    if window_size is None:
        window_size = h // 2
    if n_samples is None:
        n_samples = min(100, (h // window_size) * (w // window_size))
    if n_samples == 0:
        raise ValueError(f"Window size {window_size} is too large for image of size {img.shape}")
    origins = []
    for _ in range(n_samples):
        y = np.random.randint(0, h - window_size)
        x = np.random.randint(0, w - window_size)
        origins.append((y, x))
    line = radial_profile(autocorr_mean(img, window_size, origins, standardize), method=radial_method)
    return find_highest_peak(line, min_distance, prominence_factor, min_width, smooth_sigma, max_distance=len(line))[0]


def denoise_fft(image, p):
    """``denoise/_denoise_fft.py:4-47``."""
    fft_image = np.fft.fft2(image)
    power = (np.abs(fft_image) ** 2).ravel()
    num_keep = int(np.ceil(p * power.size))
    top = np.argpartition(power, -num_keep)[-num_keep:]
    mask = np.zeros(power.size, dtype=bool)
    mask[top] = True
    return np.real(np.fft.ifft2(fft_image * mask.reshape(fft_image.shape)))


def cumulative_energy(patch, window_type='hann', normalize=True, epsilon=1e-10):
    """``_estimate_n_max.py:8-86`` (hann / no window)."""
    size = patch.shape[0]
    windowed = patch * np.outer(np.hanning(size), np.hanning(size)) if window_type is not None else patch.copy()
    power = np.abs(np.fft.fftshift(np.fft.fft2(windowed))) ** 2
    profile = radial_profile(power)
    cum = np.cumsum(profile * np.arange(len(profile)))
    if normalize:
        cum = cum / cum[-1] if cum[-1] > epsilon else np.zeros_like(cum)
    return cum, profile, power


_DB2_HI = np.array([-0.48296291314469025, 0.836516303737469, -0.22414386804185735, -0.12940952255092145])


def estimate_sigma(image):
    """``skimage.restoration.estimate_sigma`` for a 2-D image (restated, unpinned): median |db2 'dd' coefficient| / 0.6745."""
    def high(x, axis):
        x = np.moveaxis(np.asarray(x, dtype=np.float64), axis, -1)
        ext = np.concatenate([x[..., 2::-1], x, x[..., :-4:-1]], axis=-1)
        full = np.apply_along_axis(lambda v: np.convolve(v, _DB2_HI, mode='valid'), -1, ext)
        return np.moveaxis(full[..., 1::2], -1, axis)
    d = high(high(image, 0), 1)
    d = d[np.nonzero(d)]
    return float(np.median(np.abs(d)) / 0.6744897501960817)


def add_gaussian_noise(img, sigma=0.1, seed=None):
    """``datasets/_noise_models.py:40-66``."""
    img = np.asarray(img, dtype=np.float32)
    rng = np.random.default_rng(seed)
    return img + rng.normal(0.0, sigma, size=img.shape).astype(np.float32)


def estimate_n_max(img, patch_size, n_samples=50, p=0.01, t=0.01):
    """``_estimate_n_max.py:96-125`` (same np.random draws)."""
    def get_ps(im):
        h, w = im.shape
        out = []
        for _ in range(n_samples):
            y = np.random.randint(0, h - patch_size)
            x = np.random.randint(0, w - patch_size)
            out.append(im[y:y + patch_size, x:x + patch_size])
        return np.array(out)
    sigma = estimate_sigma(img)
    if sigma > t:
        img_denoised = denoise_fft(img, p=p)
    else:
        img_denoised = img.copy()
        img = add_gaussian_noise(img_denoised, sigma=0.3)
    ps, ps_d = get_ps(img), get_ps(img_denoised)
    out = []
    for a, b in zip(ps, ps_d):
        l = cumulative_energy(b)[0] - cumulative_energy(a)[0]
        out.append(min(max(12, np.argmax(l) * 2), a.shape[0] // 2))
    return np.median(out)
