"""oracle/stabilo_ref.c (the CPU baseline's C restatement) held against oracle/stabilo_ref.py (the parity oracle): gray,
keypoints (level, pixel, orientation bin, descriptor), matches -- bit for bit; the homography -- to 1e-6 px on a 9 x 16 grid
(its f64 linear solves are not LAPACK's). Projective and affine, with and without the foreground mask and the ratio test."""
import numpy as np
import pytest

from oracle import stabilo_c as sc
from oracle import stabilo_ref as sr

CFG = dict(downsample_ratio=0.5, max_features=400, ref_multiplier=2.0, filter_ratio=0.9, ransac_threshold=2.0, mask_use=True, mask_margin_ratio=0.15,
           fast_threshold=20, n_levels=8, scale_factor=1.2, seed=0)


def _pair(h=432, w=768):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "geo-trax_amd"))
    from geotrax_amd.synth import make_scene

    s = make_scene(seed=3, h=h, w=w)
    return s.render(0, 150), s.render(40, 150), s.boxes(0), s.boxes(40)


@pytest.mark.parametrize("half,mask,extra", [(True, True, {}), (False, False, {}), (True, True, {"transformation_type": "affine"}), (True, False, {"filter_type": "none"})])
def test_c_restatement_equals_the_numpy_oracle(half, mask, extra):
    f0, f1, b0, b1 = _pair()
    cfg = dict(CFG, downsample_ratio=0.5 if half else 1.0, mask_use=mask, **extra)
    pat = sr.brief_pattern()
    a, b = sr.StabilizerRef(cfg, f0.shape[:2], pat, n_hyp=256), sc.StabilizerC(cfg, f0.shape[:2], pat, n_hyp=256)
    np.testing.assert_array_equal(sc.gray(f1, half), sr.bgr2gray(f1, half))
    sc.set_threads(3)
    a.set_ref_frame(f0, b0 if mask else None)
    b.set_ref_frame(f0, b0 if mask else None)
    Ha, na = a.stabilize(f1, b1 if mask else None)
    Hb, nb = b.stabilize(f1, b1 if mask else None)
    for side in ("ref", "cur"):
        x, y = getattr(a, side), getattr(b, side)
        assert len(x["bin"]) == len(y["bin"]) > 100
        for k in ("level", "px", "bin", "desc", "xy"):
            np.testing.assert_array_equal(x[k], y[k], err_msg=f"{side} {k}")
    for x, y in zip(a.m, b.m):
        np.testing.assert_array_equal(x, y)
    assert Ha is not None and Hb is not None and na == nb
    ys, xs = np.meshgrid(np.linspace(0, f0.shape[0] - 1, 9), np.linspace(0, f0.shape[1] - 1, 16), indexing="ij")
    g = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    pa, pb = Ha @ g, Hb @ g
    assert np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max() < 1e-6
    sc.set_threads(1)                                              # the thread count changes nothing
    Hc, _ = b.stabilize(f1, b1 if mask else None)
    np.testing.assert_array_equal(Hb, Hc)
