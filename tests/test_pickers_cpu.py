"""Parameter pickers, CPU side: the oracle restatement (oracle/pickers_oracle.py) against goldens captured from the
reference's own code, and the product's host-only pieces (peak search, noise model, wavelet noise estimate)."""
import os

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def pg():
    with np.load(os.path.join(ROOT, "tests", "golden", "pickers_golden.npz")) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="module")
def po():
    from oracle import pickers_oracle
    return pickers_oracle


def test_oracle_autocorrelation_matches_reference(pg, po):
    np.testing.assert_allclose(po.standardize_image(pg["win_64"]), pg["std_win_64"], rtol=1e-12, atol=1e-12)
    for key_in, key_out, std in (("win_64", "autocorr_64", True), ("win_64", "autocorr_64_raw", False), ("win_33", "autocorr_33", True)):
        got = po.autocorrelation(pg[key_in], standardize=std)
        np.testing.assert_allclose(got, pg[key_out], rtol=1e-10, atol=1e-9 * np.abs(pg[key_out]).max())
    got = po.autocorr_mean(pg["noisy_192"], 96, pg["origins_96"])
    np.testing.assert_allclose(got, pg["autocorr_mean_96"], rtol=1e-10, atol=1e-9 * np.abs(got).max())
    with pytest.raises(ValueError, match="Standard deviation is zero"):
        po.standardize_image(np.ones((4, 4)))


def test_oracle_and_product_peak_search_match_reference(pg, po):
    from mtflearn_amd.features import pickers
    for impl in (po.find_highest_peak, pickers.find_highest_peak):
        peak, peaks, props = impl(pg["profile_a"], max_distance=len(pg["profile_a"]))
        assert peak == int(pg["profile_a_peak"])
        np.testing.assert_array_equal(peaks, pg["profile_a_all"])
        np.testing.assert_allclose(props["prominences"], pg["profile_a_prominences"], rtol=1e-12)
        np.testing.assert_allclose(props["widths"], pg["profile_a_widths"], rtol=1e-12)
        peak, peaks, _ = impl(pg["profile_flat"], max_distance=len(pg["profile_flat"]))
        assert (peak is None) == (int(pg["profile_flat_found"]) == 0) and (peaks is None) == (peak is None)
        peak, peaks, _ = impl(pg["profile_c"], min_distance=5, max_distance=len(pg["profile_c"]))
        assert (-1 if peak is None else peak) == int(pg["profile_c_peak"])
        np.testing.assert_array_equal([] if peaks is None else peaks, pg["profile_c_all"])


def test_oracle_denoise_fft_matches_reference(pg, po):
    for k in ("", "_f32"):
        got = po.denoise_fft(pg["denoise_in" + k], float(pg["denoise_p" + k]))
        np.testing.assert_allclose(got, pg["denoise_out" + k], rtol=1e-9, atol=1e-9 * np.abs(pg["denoise_out" + k]).max())


def test_noise_model_matches_reference(pg, po):
    from mtflearn_amd.features import pickers
    for impl in (po.add_gaussian_noise, pickers.add_gaussian_noise):
        got = impl(pg["lattice_192"], sigma=0.2, seed=5)
        assert got.dtype == np.float32
        np.testing.assert_array_equal(got, pg["noisy_192"])
    with pytest.raises(ValueError, match="non-negative"):
        pickers.add_gaussian_noise(pg["lattice_192"], sigma=-1)


def test_oracle_wavelet_noise_estimate_tracks_sigma(pg, po):
    """estimate_sigma is restated from scikit-image / PyWavelets (parity unpinned): on white noise of known sigma the
    estimate is the sigma, and it separates the clean lattice from its noisy twin."""
    rng = np.random.default_rng(1)
    noise = 0.37 * rng.standard_normal((512, 512))
    assert abs(po.estimate_sigma(noise) - 0.37) < 0.01
    assert po.estimate_sigma(pg["lattice_192"]) < 0.01 < po.estimate_sigma(pg["noisy_192"])


def test_picker_argument_errors_need_no_device():
    from mtflearn_amd.features import pickers
    with pytest.raises(ValueError, match="Invalid method 'median'"):
        pickers.radial_profile(np.zeros((8, 8)), method="median")
    with pytest.raises(TypeError, match="numpy array"):
        pickers.denoise_fft([[1.0, 2.0]], 0.5)
    with pytest.raises(ValueError, match="2D array"):
        pickers.denoise_fft(np.zeros(4), 0.5)
    with pytest.raises(ValueError, match="between 0 and 1"):
        pickers.denoise_fft(np.zeros((4, 4)), 0.0)
    with pytest.raises(ValueError, match="too large for image"):
        pickers.estimate_patch_size(np.zeros((16, 16)), window_size=32)
    with pytest.raises(ValueError, match="Unknown window type"):
        pickers._window_1d("kaiser", 8)
