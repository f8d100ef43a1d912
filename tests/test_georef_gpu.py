"""gtx_op_georef_points (K12): the frame-pixel -> orthophoto -> lat/lon -> local-metres chain as one
HIP pass, against oracle/georef_ref.py (independent restatement: OpenCV's perspectiveTransform rule, the reference's
affine, and Snyder's transverse-Mercator series -- a different formula from the Krueger series the kernel uses),
against the reference tests' known answers, and (tighter, as an internal consistency check) against the package's
own host functions.

Tolerances (f64 both sides; the device libm differs from numpy's in the last ulps of sin/sinh/atanh):
orthophoto pixels and degrees 1e-12 relative, metres 1e-6 absolute (a micrometre at 5e6 m is 2e-13 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ORTHO = (126.6412, 37.3951, 2.4e-7, -1.9e-7, 1.1e-9, -0.7e-9)          # lng0, lat0, dlng, dlat, skew_x, skew_y (Songdo-like)
HOM = np.array([[1.91, 0.08, 5120.5], [-0.07, 1.88, 3310.25], [1.3e-6, -0.9e-6, 1.0]])


@pytest.mark.parametrize("target", ["EPSG:5186", "EPSG:32652", "EPSG:32734", None])
@pytest.mark.parametrize("n", [1, 777, 200_000])
def test_chain_matches_host_functions(gtx_ctx, target, n):
    from geotrax_amd.georeference import apply_homography, geo2local, ortho2geo, transform_points

    rng = np.random.default_rng(n)
    x, y = rng.uniform(0, 3840, n), rng.uniform(0, 2160, n)
    ortho = ORTHO if target != "EPSG:32734" else (20.9, -33.1, 2.4e-7, -1.9e-7, 0.0, 0.0)
    got = transform_points(x, y, HOM, ortho, "EPSG:4326" if target else None, target, ctx=gtx_ctx)
    ox, oy = apply_homography(x, y, HOM)
    lat, lon = ortho2geo(ox, oy, ortho)
    np.testing.assert_allclose(got["ortho_x"], ox, rtol=1e-13)
    np.testing.assert_allclose(got["ortho_y"], oy, rtol=1e-13)
    np.testing.assert_allclose(got["latitude"], lat, rtol=1e-14)
    np.testing.assert_allclose(got["longitude"], lon, rtol=1e-14)
    if target is None:
        assert "x_local" not in got
        return
    xl, yl = geo2local(lat, lon, "EPSG:4326", target)
    np.testing.assert_allclose(got["x_local"], xl, atol=1e-6, rtol=0)
    np.testing.assert_allclose(got["y_local"], yl, atol=1e-6, rtol=0)


@pytest.mark.parametrize("target,ortho", [("EPSG:5186", ORTHO), ("EPSG:32652", (128.6, 36.0, 2.4e-7, -1.9e-7, 0.0, 0.0)),
                                          ("EPSG:32734", (20.9, -33.1, 2.4e-7, -1.9e-7, 0.0, 0.0))])
def test_chain_matches_the_independent_oracle(gtx_ctx, target, ortho):
    """HIP chain vs oracle/georef_ref.py. The oracle's transverse Mercator is Snyder's longitude-difference series (USGS PP
    1395), the kernel's is the Krueger n-series: no shared formula. Pixels / degrees agree to rounding; metres to 0.1 mm
    (the oracle series' own truncation: 0.01-0.3 mm within a degree of the central meridian)."""
    from geotrax_amd.georeference import transform_points
    from oracle import georef_ref as R

    rng = np.random.default_rng(11)
    x, y = rng.uniform(0, 3840, 50_000), rng.uniform(0, 2160, 50_000)
    got = transform_points(x, y, HOM, ortho, "EPSG:4326", target, ctx=gtx_ctx)
    ox, oy = R.apply_homography(x, y, HOM)
    lat, lon = R.ortho2geo(ox, oy, ortho)
    np.testing.assert_allclose(got["ortho_x"], ox, rtol=1e-13)
    np.testing.assert_allclose(got["ortho_y"], oy, rtol=1e-13)
    np.testing.assert_allclose(got["latitude"], lat, rtol=1e-14)
    np.testing.assert_allclose(got["longitude"], lon, rtol=1e-14)
    xl, yl = R.geo2local(lat, lon, "EPSG:4326", target)
    np.testing.assert_allclose(got["x_local"], xl, atol=4e-4, rtol=0)
    np.testing.assert_allclose(got["y_local"], yl, atol=4e-4, rtol=0)


def test_chain_reference_known_answers(gtx_ctx):
    """The reference's own pinned values (tests/test_georeference.py:31-63): identity and translation homographies,
    and pyproj's EPSG:4326 -> EPSG:32631 answer for (6.6 E, 46.5 N) through an identity geotransform."""
    from geotrax_amd.georeference import frame2local, transform_points

    ident_ortho = (0.0, 0.0, 1.0, 1.0, 0.0, 0.0)
    got = transform_points([6.6], [46.5], np.eye(3), ident_ortho, "EPSG:4326", "EPSG:32631", ctx=gtx_ctx)
    np.testing.assert_allclose(got["x_local"], [776225.4478], atol=1e-3)
    np.testing.assert_allclose(got["y_local"], [5155902.1301], atol=1e-3)
    shift = np.array([[1.0, 0, 10.0], [0, 1.0, -5.0], [0, 0, 1.0]])
    got = transform_points([1.0, 2.0], [3.0, 4.0], shift, ident_ortho, ctx=gtx_ctx)
    np.testing.assert_array_equal(got["ortho_x"], [11.0, 12.0])
    np.testing.assert_array_equal(got["ortho_y"], [-2.0, -1.0])
    # projection origin of Korea 2000 / Central Belt 2010
    got = transform_points([127.0], [38.0], np.eye(3), ident_ortho, "EPSG:4326", "EPSG:5186", ctx=gtx_ctx)
    np.testing.assert_allclose([got["x_local"][0], got["y_local"][0]], [200000.0, 600000.0], atol=1e-6)
    # frame2local with a context is the same chain
    pts = np.array([[100.0, 200.0], [3000.0, 1500.0]])
    a = frame2local(pts, HOM, ORTHO, "EPSG:4326", "EPSG:5186", ctx=gtx_ctx)
    b = frame2local(pts, HOM, ORTHO, "EPSG:4326", "EPSG:5186")
    np.testing.assert_allclose(a, b, atol=1e-6, rtol=0)


def test_chain_edge_cases(gtx_ctx):
    from geotrax_amd.georeference import transform_points

    got = transform_points([], [], HOM, ORTHO, "EPSG:4326", "EPSG:5186", ctx=gtx_ctx)
    assert all(len(v) == 0 for v in got.values()) and set(got) == {"ortho_x", "ortho_y", "latitude", "longitude", "x_local", "y_local"}
    # a point on the homography's line at infinity maps to (0, 0), as cv2.perspectiveTransform does
    Hinf = np.array([[1.0, 0, 0], [0, 1.0, 0], [1.0, 0, -5.0]])
    got = transform_points([5.0], [7.0], Hinf, (0.0, 0.0, 1.0, 1.0, 0.0, 0.0), ctx=gtx_ctx)
    assert got["ortho_x"][0] == 0.0 and got["ortho_y"][0] == 0.0
    with pytest.raises(NotImplementedError):
        transform_points([1.0], [1.0], HOM, ORTHO, "EPSG:4326", "EPSG:3857", ctx=gtx_ctx)
    with pytest.raises(ValueError):
        transform_points([1.0, 2.0], [1.0], HOM, ORTHO, ctx=gtx_ctx)


def test_hip_chain_on_the_golden_georeference_rows(gtx_ctx):
    """The HIP kernel itself (gtx_op_georef_points) on the reference-held rows: Ortho_X / Ortho_Y of the golden CSV ==
    round(persp(H_geo, x_stab, y_stab), 1) for all 19 787 rows the CSV keeps (tests/test_geometry.py pins the HOST function on
    the same rows; SURVEY 8c's fixture table: 100 % of the rows, exactly)."""
    import gzip
    from pathlib import Path

    from geotrax_amd.georeference import transform_points

    G = Path(__file__).parent / "golden"
    with gzip.open(G / "U_video_cut.txt.gz", "rt") as f:
        t = np.loadtxt(f, delimiter=",")
    Hg = np.loadtxt(G / "U_video_cut_geo_transf.txt", delimiter=",").reshape(3, 3)
    c = np.load(G / "U_video_cut_csv_cols.npz")
    key = {(int(fr), int(i)): k for k, (fr, i) in enumerate(zip(t[:, 0], t[:, 1]))}
    rows = np.array([key[(int(fr), int(i))] for fr, i in zip(c["frame"], c["vehicle_id"])])
    got = transform_points(t[rows, 6], t[rows, 7], Hg, ORTHO, None, None, ctx=gtx_ctx)
    assert len(got["ortho_x"]) == 19787
    np.testing.assert_array_equal(np.round(got["ortho_x"], 1), c["ortho_x"])
    np.testing.assert_array_equal(np.round(got["ortho_y"], 1), c["ortho_y"])
