"""Projective helpers behind gtx_warp_boxes / gtx_perspective_points against the reference's
golden outputs and its own unit-test cases. Host-only entry points: run on CPU."""
import gzip
from pathlib import Path

import numpy as np

from geotrax_amd import geometry as geo

G = Path(__file__).parent / "golden"


def _golden_tracks():
    with gzip.open(G / "U_video_cut.txt.gz", "rt") as f:
        return np.loadtxt(f, delimiter=",")


def test_box_warp_rule_on_golden_clip():
    """K10: raw boxes (cols 2-5) + per-frame homographies (_vid_transf.txt) -> stabilized boxes
    (cols 6-9). The corner-hull rule reproduces every golden box after frame 0 (19 682) to the %g print quantum
    (x >= 1000 px is stored to 0.01 px); frame 0 is copied through (extract.py:178-179)."""
    t = _golden_tracks()
    tr = np.loadtxt(G / "U_video_cut_vid_transf.txt", delimiter=",")
    Hs = {int(r[0]): r[1:].reshape(3, 3) for r in tr}
    assert len(Hs) == 149 and set(Hs) == set(range(1, 150))
    n = 0
    for f in range(150):
        rows = t[t[:, 0] == f]
        if f == 0:
            np.testing.assert_array_equal(rows[:, 6:10], rows[:, 2:6])
            continue
        got = geo.warp_boxes(Hs[f], rows[:, 2:6])
        assert np.abs(got - rows[:, 6:10]).max() < 0.0125
        n += len(rows)
    assert n == len(t) - (t[:, 0] == 0).sum() and n > 19600


def test_perspective_points_on_golden_georeference():
    """K12: Ortho_X/Y of the golden CSV == round(persp(H_geo, x_stab, y_stab), 1) for every row
    the CSV keeps (it drops trajectories shorter than min_traj_length)."""
    t = _golden_tracks()
    Hg = np.loadtxt(G / "U_video_cut_geo_transf.txt", delimiter=",").reshape(3, 3)
    c = np.load(G / "U_video_cut_csv_cols.npz")
    key = {(int(f), int(i)): k for k, (f, i) in enumerate(zip(t[:, 0], t[:, 1]))}
    rows = np.array([key[(int(f), int(i))] for f, i in zip(c["frame"], c["vehicle_id"])])
    ox, oy = geo.apply_homography(t[rows, 6], t[rows, 7], Hg)
    assert len(ox) == 19787
    np.testing.assert_array_equal(np.round(ox, 1), c["ortho_x"])
    np.testing.assert_array_equal(np.round(oy, 1), c["ortho_y"])


def test_apply_homography_reference_cases():
    # tests/test_georeference.py:31-43 of the reference: identity and pure translation
    x, y = np.array([10.0, 20.0, 30.5]), np.array([5.0, 15.0, 25.5])
    ox, oy = geo.apply_homography(x, y, np.eye(3))
    np.testing.assert_allclose(ox, x)
    np.testing.assert_allclose(oy, y)
    Ht = np.array([[1, 0, 7.0], [0, 1, -3.0], [0, 0, 1]])
    ox, oy = geo.apply_homography(x, y, Ht)
    np.testing.assert_allclose(ox, x + 7)
    np.testing.assert_allclose(oy, y - 3)


def test_ortho2geo_reference_case():
    # tests/test_georeference.py:46-51: plain affine
    lat, lon = geo.ortho2geo(np.array([10.0]), np.array([20.0]), (126.0, 37.0, 1e-6, -2e-6, 0.0, 0.0))
    np.testing.assert_allclose(lon, 126.0 + 1e-5)
    np.testing.assert_allclose(lat, 37.0 - 4e-5)


def test_warp_boxes_projective_and_empty():
    H = np.array([[1.01, 0.02, 3.0], [-0.015, 0.99, -2.0], [1e-5, -2e-5, 1.0]])
    b = np.array([[100, 200, 40, 20], [3000, 1500, 90, 45]], dtype=np.float32)
    got = geo.warp_boxes(H, b)
    for (cx, cy, w, h), g in zip(b.astype(np.float64), got):
        cs = np.array([[cx - w / 2, cy - h / 2, 1], [cx + w / 2, cy - h / 2, 1], [cx + w / 2, cy + h / 2, 1], [cx - w / 2, cy + h / 2, 1]]).T
        p = H @ cs
        p = p[:2] / p[2]
        exp = [(p[0].min() + p[0].max()) / 2, (p[1].min() + p[1].max()) / 2, np.ptp(p[0]), np.ptp(p[1])]
        np.testing.assert_allclose(g, exp, rtol=1e-6)
    assert geo.warp_boxes(H, np.zeros((0, 4), np.float32)).shape == (0, 4)


def test_estimate_affine_partial_equals_the_oracle_and_recovers_a_similarity():
    """gtx_op_estimate_affine_partial (host C++; the fit of `gmc_method: orb` / `sift`) == oracle/gmc_ref.estimate_affine_partial on
    seeded matches with 30 % outliers, and both recover the similarity the inliers were made with."""
    from geotrax_amd.gmc import estimate_affine_partial
    from oracle import gmc_ref

    rng = np.random.default_rng(7)
    for n in (5, 60, 800):
        p = rng.uniform(0, 1000, (n, 2)).astype(np.float32)
        a, s = 0.01, 1.002
        M = np.array([[s * np.cos(a), -s * np.sin(a), 12.5], [s * np.sin(a), s * np.cos(a), -7.25]])
        q = (p.astype(np.float64) @ M[:, :2].T + M[:, 2] + 0.3 * rng.standard_normal((n, 2))).astype(np.float32)
        bad = rng.random(n) < 0.3
        q[bad] += rng.uniform(-200, 200, (int(bad.sum()), 2)).astype(np.float32)
        got, inl = estimate_affine_partial(p, q, seed=3)
        want = gmc_ref.estimate_affine_partial(p, q, seed=3)
        assert got is not None and want is not None
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
        if n >= 60:
            assert np.abs(got - M).max() < 0.2 and inl > 0.6 * n
    got, inl = estimate_affine_partial(np.zeros((1, 2), np.float32), np.zeros((1, 2), np.float32))
    assert got is None and inl == 0
