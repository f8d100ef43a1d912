"""CPU checks of the ECC restatement (oracle/ecc_ref.py) -- no GPU. The oracle is test infrastructure; what it is held against
here are properties OpenCV's procedure has by construction: the preprocessing kernels on hand-computable images, the fit on
image pairs with a known Euclidean motion, and the two error conditions under which cv2.findTransformECC raises."""
import numpy as np
import pytest

from oracle import ecc_ref


def _texture(h, w, seed=0, cell=8):
    """A smooth random BGR image (bicubic-ish: box-smoothed noise upsampled by `cell`) with margin to move a window over."""
    rng = np.random.default_rng(seed)
    base = rng.random((h // cell + 12, w // cell + 12, 3))
    big = np.kron(base, np.ones((cell, cell, 1)))
    k = np.ones(2 * cell + 1) / (2 * cell + 1)
    for ax in (0, 1):
        big = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, big)
    big = big[3 * cell:-3 * cell, 3 * cell:-3 * cell]
    return (255 * (big - big.min()) / (big.max() - big.min()))


def _view(big, h, w, tx, ty, theta):
    """frame(y, x) = big(R (x, y) + (tx, ty) + margin), bilinear."""
    c, s = np.cos(theta), np.sin(theta)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx = c * xs - s * ys + tx + 2 * 8
    sy = s * xs + c * ys + ty + 2 * 8
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    x0 = np.clip(x0, 0, big.shape[1] - 2)
    y0 = np.clip(y0, 0, big.shape[0] - 2)
    v = (big[y0, x0] * (1 - fx) * (1 - fy) + big[y0, x0 + 1] * fx * (1 - fy) + big[y0 + 1, x0] * (1 - fx) * fy + big[y0 + 1, x0 + 1] * fx * fy)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def test_preprocessing_on_hand_computable_images():
    flat = np.full((10, 12, 3), 77, np.uint8)
    g = ecc_ref.gray_bgr(flat)
    assert (g == 77).all()                                              # the three weights sum to 2^14
    assert (ecc_ref.gaussian_blur3(g) == 77).all()                      # (79 + 98 + 79)^2 = 2^16: a constant image is a fixed point
    assert ecc_ref.prepare(flat).shape == (5, 6)
    imp = np.zeros((7, 7), np.uint8)
    imp[3, 3] = 255
    b = ecc_ref.gaussian_blur3(imp)
    want = np.outer([79, 98, 79], [79, 98, 79]) * 255
    np.testing.assert_array_equal(b[2:5, 2:5], (want + (1 << 15)) >> 16)
    assert b.sum() == b[2:5, 2:5].sum()
    edge = np.zeros((4, 6), np.uint8)
    edge[:, 0] = 200                                                    # BORDER_REFLECT_101: column -1 mirrors column 1 (zero), not column 0
    assert ecc_ref.gaussian_blur3(edge)[0, 0] == (98 * 200 * 256 + (1 << 15)) >> 16
    ramp = np.arange(8 * 6, dtype=np.uint8).reshape(8, 6)
    h = ecc_ref.half_size(ramp)
    assert h.shape == (4, 3) and h[0, 0] == (0 + 1 + 6 + 7 + 2) >> 2 and h[3, 2] == (40 + 41 + 46 + 47 + 2) >> 2
    gx, gy = ecc_ref.gradients(ramp.astype(np.float32))
    assert (gx[:, 1:-1] == 1).all() and (gx[:, 0] == 0).all() and (gx[:, -1] == 0).all() and (gy[1:-1] == 6).all() and (gy[0] == 0).all()


def test_warp_coordinates_follow_warpaffine_fixed_point():
    M = np.array([[1, 0, 2.5], [0, 1, -0.25]], np.float32)
    (sx, sy, fx, fy), (nx, ny) = ecc_ref.warp_coords(M, 4, 5)
    np.testing.assert_array_equal(sx[0], [2, 3, 4, 5, 6])
    assert (fx == 16).all() and (sy[:, 0] == [-1, 0, 1, 2]).all() and (fy == 24).all()
    np.testing.assert_array_equal(nx[0], [3, 4, 5, 6, 7])             # 2.5 rounds up with round_delta = 1/2
    np.testing.assert_array_equal(ny[:, 0], [0, 1, 2, 3])             # -0.25 + 0.5 -> floor 0
    img = np.arange(20, dtype=np.float32).reshape(4, 5)
    w = ecc_ref.warp_linear(img, (sx, sy, fx, fy))
    assert w[1, 0] == pytest.approx(0.25 * 0.5 * (2 + 3) + 0.75 * 0.5 * (7 + 8))                 # rows 0 / 1 at 0.75, columns 2 / 3 at 0.5
    assert w[0, 0] == pytest.approx(0.75 * 0.5 * (2 + 3)) and w[0, 4] == 0.0                     # row -1 and column 6 are the constant border


@pytest.mark.parametrize("warp", ["exact", "fixed"])
@pytest.mark.parametrize("motion", [(3.0, -1.5, 0.0), (-4.25, 2.0, 0.004), (0.0, 0.0, -0.01), (7.5, 5.0, 0.002)])
def test_fit_recovers_a_known_euclidean_motion(motion, warp):
    """Both forms of warpAffine's bilinear path. The texture here is piecewise linear (box-filtered blocks): the Gauss-Newton step
    keeps hopping over its kinks by a few thousandths of a pixel, so the 1e-6 criterion may never be met -- the cap of 100 ends the
    fit, well inside the accuracy asked for (on the rendered scenes of tests/test_ecc_gpu.py the exact form ends after 8-10 iterations)."""
    tx, ty, th = motion
    h, w = 216, 384
    big = _texture(h + 40, w + 40, seed=3)
    ref = ecc_ref.EccRef(max_iters=100, warp=warp)
    np.testing.assert_array_equal(ref.apply(_view(big, h, w, 0, 0, 0)), np.eye(2, 3))
    H = ref.apply(_view(big, h, w, tx, ty, th))
    assert ref.last["status"] == 0 and 2 <= ref.last["iters"] <= 100 and ref.last["rho"] > 0.95
    # frame(x) = first(R x + t) in full-resolution pixels; the warp maps first-frame (template) pixels to frame pixels: x -> R^-1 (x - t),
    # in half-resolution units (pixel centres: x_half = (x_full - 0.5) / 2)
    c, s = np.cos(th), np.sin(th)
    Rinv = np.array([[c, s], [-s, c]])
    t_full = -Rinv @ np.array([tx, ty])
    t_half = (t_full + (Rinv @ np.array([0.5, 0.5]) - 0.5)) / 2
    np.testing.assert_allclose(H[:, :2], Rinv, atol=2e-4)
    np.testing.assert_allclose(H[:, 2], t_half, atol=0.05)
    again = ref.apply(_view(big, h, w, tx, ty, th))                     # the template stays the first frame: the same frame, the same warp
    np.testing.assert_array_equal(again, H)


def test_exact_positions_end_fits_that_fixed_point_positions_keep_dithering():
    """A rendered scene pair: with OpenCV >= 4.11's floating-point source positions the coefficient settles and the 1e-6 criterion
    ends the fit; with the 1/32-pixel positions of older builds the samples change in steps, the coefficient dithers in its sixth
    decimal and the fit runs to its cap. The two warps agree to a hundredth of a pixel."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "geo-trax_amd"))
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=3, h=270, w=480)
    f0, f1 = sc.render(0), sc.render(30)
    out = {}
    for warp in ("exact", "fixed"):
        e = ecc_ref.EccRef(max_iters=150, warp=warp)
        e.apply(f0)
        out[warp] = (e.apply(f1), dict(e.last))
    assert out["exact"][1]["iters"] < 40 and out["exact"][1]["status"] == 0
    assert out["fixed"][1]["iters"] >= out["exact"][1]["iters"]
    assert np.abs(out["exact"][0] - out["fixed"][0]).max() < 0.02


def test_error_conditions_keep_the_matrix_as_the_failed_call_left_it():
    flat = np.full((64, 96, 3), 100, np.uint8)                           # zero variance: the correlation coefficient is 0 / 0
    ref = ecc_ref.EccRef()
    ref.apply(flat)
    np.testing.assert_array_equal(ref.apply(flat), np.eye(2, 3))
    assert ref.last["status"] == 1 and ref.last["iters"] == 1
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    ref = ecc_ref.EccRef(max_iters=50, warp="fixed")
    ref.apply(a)
    H = ref.apply(255 - a)                                               # anti-correlated: lambda's denominator is not positive
    assert ref.last["status"] == 2 and np.isfinite(H).all()
