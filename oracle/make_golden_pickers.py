#!/usr/bin/env python3
"""Generate tests/golden/pickers_golden.npz from the REFERENCE's parameter-picker code (build container only).

TEST INFRASTRUCTURE.  ``mtflearn/features/_patch_size.py`` and ``_estimate_n_max.py`` import scikit-image at module
level, which is not installed here.  The modules are loaded under stub packages (as oracle/make_golden.py does for the
package ``__init__`` files) with EMPTY stand-in modules for ``skimage.transform`` / ``skimage.restoration`` whose
``warp_polar`` / ``estimate_sigma`` raise if called: they only satisfy the import statement.  This script records
outputs of reference functions that never reach those two calls -- ``standardize_image``, ``autocorrelation``,
``find_highest_peak`` (``_patch_size.py``), ``denoise_fft`` (``denoise/_denoise_fft.py``, numpy only),
``add_gaussian_noise`` (``datasets/_noise_models.py``, numpy only).  ``radial_profile`` and everything composed from it
cannot be captured (it IS the scikit-image call): see oracle/pickers_oracle.py, "parity unpinned".
No reference source or bytecode is copied; the fixtures are data.

Usage:  python oracle/make_golden_pickers.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "pickers_golden.npz")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("mtflearn", "mtflearn.features", "mtflearn.datasets", "mtflearn.denoise"):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, *name.split("."))]
        sys.modules[name] = mod

    def absent(*a, **k):
        raise RuntimeError("scikit-image is not installed: this stand-in only satisfies the import")
    for name, attr in (("skimage", None), ("skimage.transform", "warp_polar"), ("skimage.restoration", "estimate_sigma")):
        mod = types.ModuleType(name)
        if attr:
            setattr(mod, attr, absent)
        sys.modules[name] = mod
    import matplotlib
    matplotlib.use("Agg")
    from mtflearn.features import _patch_size
    from mtflearn.denoise._denoise_fft import denoise_fft
    from mtflearn.datasets._noise_models import add_gaussian_noise
    from mtflearn.datasets._honeycomb_lattice import HoneyCombLattice
    return _patch_size, denoise_fft, add_gaussian_noise, HoneyCombLattice


def main():
    ps, denoise_fft, add_noise, HoneyComb = import_reference()
    g = {}
    rng = np.random.default_rng(20261004)
    lattice = HoneyComb(size=192, l=12, seed=0).to_image().astype(np.float32)
    g["lattice_192"] = lattice
    noisy = add_noise(lattice, sigma=0.2, seed=5)
    g["noisy_192"] = noisy                                          # reference add_gaussian_noise(seed=5)

    # standardize_image / autocorrelation (scipy.signal.correlate 'same', 'fft')
    win = noisy[17:17 + 64, 40:40 + 64]
    g["win_64"] = win
    g["std_win_64"] = ps.standardize_image(win)
    g["autocorr_64"] = ps.autocorrelation(image=win, standardize=True)
    g["autocorr_64_raw"] = ps.autocorrelation(image=win, standardize=False)
    odd = rng.random((33, 33))
    g["win_33"] = odd
    g["autocorr_33"] = ps.autocorrelation(image=odd, standardize=True)
    origins = np.array([[0, 0], [64, 32], [90, 96], [31, 7]])
    g["origins_96"] = origins
    g["autocorr_mean_96"] = np.mean([ps.autocorrelation(image=noisy[y:y + 96, x:x + 96], standardize=True)
                                     for y, x in origins], axis=0)

    # find_highest_peak on two synthetic profiles and on a real autocorrelation cross-section
    r = np.arange(140, dtype=np.float64)
    prof = np.exp(-r / 6.0) + 0.35 * np.exp(-0.5 * ((r - 24.0) / 2.5) ** 2) + 0.22 * np.exp(-0.5 * ((r - 48.5) / 3.0) ** 2) \
        + 0.01 * rng.standard_normal(r.size)
    g["profile_a"] = prof
    peak, peaks, props = ps.find_highest_peak(prof, max_distance=len(prof))
    g["profile_a_peak"], g["profile_a_all"] = np.array(peak), np.array(peaks)
    g["profile_a_prominences"], g["profile_a_widths"] = props["prominences"], props["widths"]
    flat = np.exp(-r / 10.0)
    g["profile_flat"] = flat
    peak, peaks, _ = ps.find_highest_peak(flat, max_distance=len(flat))
    g["profile_flat_found"] = np.array(0 if peak is None else 1)
    line = g["autocorr_mean_96"][48, 48:]
    g["profile_c"] = line
    peak, peaks, _ = ps.find_highest_peak(line, min_distance=5, max_distance=len(line))
    g["profile_c_peak"] = np.array(-1 if peak is None else peak)
    g["profile_c_all"] = np.array([] if peaks is None else peaks)

    # denoise_fft: p chosen so that the cut-off does not fall inside a tie (Hermitian pairs have equal power;
    # numpy.argpartition's choice among equal values is arbitrary)
    def untied_fraction(img, p0):
        power = np.sort((np.abs(np.fft.fft2(img)) ** 2).ravel())[::-1]
        k = next(k for k in range(int(p0 * img.size), img.size) if power[k - 1] > power[k] * (1 + 1e-9))
        p = (k - 0.5) / img.size
        assert int(np.ceil(p * img.size)) == k
        return p

    img = noisy[:96, :128].astype(np.float64)
    p = untied_fraction(img, 0.02)
    g["denoise_in"], g["denoise_p"] = img, np.array(p)
    g["denoise_out"] = denoise_fft(img, p)
    img32 = np.ascontiguousarray(noisy[:64, :64])
    p32 = untied_fraction(img32, 0.05)
    g["denoise_in_f32"], g["denoise_p_f32"] = img32, np.array(p32)
    g["denoise_out_f32"] = denoise_fft(img32, p32)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
