"""CPU-side checks of the registration row: the oracle's SIFT against known answers
(no golden vector for this path exists in the reference), and the host wrapper's interface rules."""
import logging

import numpy as np
import pytest


def _blob_image(h, w, cx, cy, sigma, amp=120.0):
    ys, xs = np.mgrid[0:h, 0:w]
    g = 60.0 + amp * np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2.0 * sigma * sigma))
    return np.repeat(np.clip(np.rint(g), 0, 255).astype(np.uint8)[:, :, None], 3, 2)


@pytest.mark.parametrize("sigma", [3.0, 6.0])
def test_oracle_sift_finds_a_gaussian_blob_at_its_centre_and_scale(sigma):
    from oracle import sift_ref as R

    img = _blob_image(96, 128, 70.3, 41.6, sigma)
    f = R.detect_and_compute(img)
    assert len(f["xy"]) >= 1
    k = int(np.argmax(f["response"]))
    assert np.hypot(f["xy"][k, 0] - 70.3, f["xy"][k, 1] - 41.6) < 0.35
    # DoG responds at sigma_blob ~ scale; OpenCV's size = 2 * scale (in input pixels)
    assert 0.75 * 2 * sigma < f["size"][k] < 1.45 * 2 * sigma
    np.testing.assert_allclose(np.linalg.norm(f["desc"], axis=1), 1.0, atol=1e-3)   # RootSIFT rows are unit vectors
    assert (f["desc"] >= 0).all()


def test_oracle_sift_is_translation_covariant():
    from oracle import sift_ref as R

    rng = np.random.default_rng(2)
    base = np.zeros((90, 150), np.float64)
    for _ in range(25):
        cx, cy, s = rng.uniform(20, 130), rng.uniform(20, 70), rng.uniform(2, 5)
        ys, xs = np.mgrid[0:90, 0:150]
        base += rng.uniform(40, 110) * np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * s * s))
    img = np.repeat(np.clip(base + 40, 0, 255).astype(np.uint8)[:, :, None], 3, 2)
    a = R.detect_and_compute(img[:, 0:120])
    b = R.detect_and_compute(img[:, 8:128])                     # same content shifted by 8 px (4 px at octave 1, ...)
    qi, ti, _ = R.match_ratio(a["desc"], b["desc"], 0.7)
    assert len(qi) >= 10
    d = a["xy"][qi] - b["xy"][ti]
    assert np.abs(np.median(d[:, 0]) - 8.0) < 0.1 and np.abs(np.median(d[:, 1])) < 0.1


def test_gaussian_taps_and_octave_count_follow_opencv_rules():
    from oracle import sift_ref as R

    t = R.gaussian_taps(1.6)
    assert len(t) == 15 and abs(float(t.sum()) - 1.0) < 1e-6 and (t == t[::-1]).all()     # ksize = round(8*1.6+1)|1 = 14|1
    assert len(R.gaussian_taps(1.2262735)) == 11
    assert R.n_octaves(4320, 7680) == 11 and R.n_octaves(360, 480) == 7                  # round(log2(min) - 2) + 1
    sig = R.layer_sigmas()
    np.testing.assert_allclose(sig[:3], [1.6, 1.2262735, 1.5450078], rtol=1e-6)


def test_estimate_homography_interface_rules_without_a_gpu():
    """Argument validation happens before the library is touched (registration.py:21-56 keyword set)."""
    from geotrax_amd.registration import estimate_homography

    img = np.zeros((32, 32, 3), np.uint8)
    log = logging.getLogger("reg")
    for kw in (dict(detector_name="orb"), dict(matcher_name="flann"), dict(filter_type="distance"), dict(sift_enable_precise_upscale=False)):
        with pytest.raises(NotImplementedError):
            estimate_homography(img, img, log, **kw)
    # max_features <= 10000: the reference's loop body never runs and it returns four Nones (registration.py:57,93-95)
    assert estimate_homography(img, img, log, max_features=10000) == (None, None, None, None)
