"""geotrax_amd.georeference against vectors produced by the reference's own functions
(tests/golden/make_golden.py -> georeference_vectors.npz / georeference_table.csv) and against the
known answers of the reference's tests (tests/test_georeference.py: reprojection, lane polygons)."""
import io
import logging
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).parent / "golden"
logger = logging.getLogger("georef-test")


@pytest.fixture(scope="module")
def vec():
    return dict(np.load(GOLD / "georeference_vectors.npz", allow_pickle=True))


def test_visibility_matches_reference(vec):
    from geotrax_amd.georeference import calculate_visibility

    np.testing.assert_array_equal(calculate_visibility(vec["track_id"], vec["bbox"], (2160, 3840), 4), vec["visibility"])
    np.testing.assert_array_equal(calculate_visibility(vec["track_id"], vec["bbox"], (2160, 3840), 10), vec["visibility_m10"])
    assert 0 < vec["visibility"].sum() < len(vec["visibility"])


@pytest.mark.parametrize("name,ft,ks,interp", [("gauss14", "gaussian", 14, False), ("gauss3_interp", "gaussian", 3, True),
                                               ("savgol7", "savgol", 7, False), ("savgol8_interp", "savgol", 8, True)])
def test_kinematics_match_reference(vec, name, ft, ks, interp):
    from geotrax_amd.georeference import compute_kinematics

    v, a = compute_kinematics(vec["track_id"], vec["frame"], vec["x_local"], vec["y_local"], vec["visibility"], 29.97, ft, ks,
                              is_interpolated=vec["is_interpolated"] if interp else None)
    np.testing.assert_array_equal(np.isnan(v), np.isnan(vec[f"speed_{name}"]))
    np.testing.assert_array_equal(np.isnan(a), np.isnan(vec[f"accel_{name}"]))
    np.testing.assert_allclose(v, vec[f"speed_{name}"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(a, vec[f"accel_{name}"], rtol=1e-12, atol=1e-12, equal_nan=True)
    assert np.isfinite(v).sum() > 100


def test_gap_interpolation_matches_reference(vec):
    from geotrax_amd.georeference import apply_filter, compute_acceleration, compute_speed, interpolate_missing_points

    xi, yi, present = interpolate_missing_points(np.array([3, 4, 7, 8, 12]), np.array([0.0, 1.0, 5.5, 6.0, 9.0]), np.array([2.0, 2.5, 1.0, 0.0, -4.0]))
    np.testing.assert_allclose(xi, vec["interp_x"], rtol=1e-15)
    np.testing.assert_allclose(yi, vec["interp_y"], rtol=1e-15)
    np.testing.assert_array_equal(present, vec["interp_present"])
    # known answers of the reference's tests (test_georeference.py:64-98)
    np.testing.assert_allclose(compute_speed(np.array([0.0, 1.0, 2.0]), np.zeros(3), fps=2.0), [2.0, 2.0])
    np.testing.assert_allclose(compute_acceleration(np.array([1.0, 2.0, 4.0]), fps=1.0), [1.0, 2.0])
    np.testing.assert_allclose(apply_filter(np.full(10, 3.0), 2, "gaussian"), 3.0)
    np.testing.assert_allclose(apply_filter(np.arange(10.0), 5, "savgol")[2:-2], np.arange(10.0)[2:-2], atol=1e-9)
    assert apply_filter(np.arange(8.0), 4, "savgol").shape == (8,)
    with pytest.raises(ValueError):
        apply_filter(np.zeros(5), 3, "nope")


def test_formatted_table_is_identical_to_the_reference_csv(vec):
    from geotrax_amd.georeference import create_and_format_georeferenced_df

    lane = vec["lane"]
    section = np.where(np.isnan(lane), None, "A").astype(object)
    df = create_and_format_georeferenced_df(vec["track_id"], np.array([]), vec["frame"], vec["stab_x"] * 3.7123, vec["stab_y"] * 3.7123,
                                            vec["x_local"], vec["y_local"], vec["lat"], vec["lon"], (vec["veh_len"], vec["veh_wid"]),
                                            vec["cls"], vec["speed_gauss3_interp"], vec["accel_gauss3_interp"], section, lane,
                                            vec["visibility"], 15, is_interpolated=vec["is_interpolated"], logger=logger)
    buf = io.StringIO()
    df.to_csv(buf, index=False)
    assert buf.getvalue() == (GOLD / "georeference_table.csv").read_text()      # byte for byte: columns, order, rounding, filter


def test_table_rules_from_the_reference_tests():
    """tests/test_georeference.py:214-292 of the reference, same inputs."""
    from geotrax_amd.georeference import create_and_format_georeferenced_df

    def inputs(n_veh=3, n_frames=4):
        total = n_veh * n_frames
        return dict(track_id=np.repeat(np.arange(1, n_veh + 1), n_frames), timestamps=np.array([]), frame_num=np.tile(np.arange(n_frames), n_veh),
                    x_stab_ortho=np.full(total, 123.456), y_stab_ortho=np.full(total, 234.567), x_local=np.full(total, 1000.123),
                    y_local=np.full(total, 2000.456), latitude=np.full(total, 37.123456789), longitude=np.full(total, 127.987654321),
                    veh_dim_real=(np.full(total, 4.567), np.full(total, 1.876)), class_id=np.zeros(total, dtype=int), v_speed=np.full(total, 50.0),
                    v_acceleration=np.zeros(total), road_section=None, lane_number=None, visibility=np.ones(total, dtype=bool))

    df = create_and_format_georeferenced_df(**inputs(), min_traj_length=0, logger=logger)
    assert len(df) == 12 and "Timestamp" not in df.columns and "Road_Section" not in df.columns and "Lane_Number" not in df.columns
    assert df["Visibility"].dtype.kind == "i"
    one = create_and_format_georeferenced_df(**inputs(1, 1), min_traj_length=0, logger=logger)
    assert one["Ortho_X"].iloc[0] == pytest.approx(123.5) and one["Local_X"].iloc[0] == pytest.approx(1000.12)
    assert one["Longitude"].iloc[0] == pytest.approx(127.9876543, abs=1e-9)
    assert len(create_and_format_georeferenced_df(**inputs(3, 4), min_traj_length=5, logger=logger)) == 0


def test_reprojection_known_answers():
    from geotrax_amd.georeference import geo2local, ortho2local

    # the reference's own known answer (tests/test_georeference.py:54-63): pyproj, EPSG:4326 -> EPSG:32631
    x, y = ortho2local(np.array([6.6]), np.array([46.5]), (0.0, 0.0, 1.0, 1.0, 0.0, 0.0), "EPSG:4326", "EPSG:32631")
    np.testing.assert_allclose(x, [776225.4478], atol=1e-3)
    np.testing.assert_allclose(y, [5155902.1301], atol=1e-3)
    # projection origin of Korea 2000 / Central Belt 2010 (the reference's default target, default.yaml:153)
    x, y = geo2local(np.array([38.0]), np.array([127.0]), "epsg:4326", "epsg:5186")
    np.testing.assert_allclose([x[0], y[0]], [200000.0, 600000.0], atol=1e-6)
    # southern hemisphere UTM: false northing 10 000 km; on the central meridian the easting is 500 km
    x, y = geo2local(np.array([-33.0]), np.array([21.0]), "EPSG:4326", "EPSG:32734")
    assert abs(x[0] - 500000.0) < 1e-6 and 6.3e6 < y[0] < 6.4e6
    # 1 degree of latitude along the central meridian ~ 111 km * 0.9996
    _, y2 = geo2local(np.array([-32.0]), np.array([21.0]), "EPSG:4326", "EPSG:32734")
    assert abs((y2[0] - y[0]) - 110_880) < 150
    with pytest.raises(NotImplementedError):
        geo2local(np.array([0.0]), np.array([0.0]), "EPSG:4326", "EPSG:3857")


def test_dimensions_follow_the_local_metric():
    from geotrax_amd.georeference import convert_dimensions

    # identity homography; orthophoto pixel = 1e-6 deg: on the 127E meridian at 38N one degree is ~111.0 km north, ~87.8 km east
    H = np.eye(3)
    params = (127.0, 38.0, 1e-6, -1e-6, 0.0, 0.0)
    tid = np.array([7, 7, 9, 9, 9])
    dims = np.array([[100.0, 40.0], [100.0, 40.0], [np.nan, np.nan], [50.0, 20.0], [50.0, 20.0]])
    L, Wd = convert_dimensions(tid, dims, (2160, 3840), H, params, "epsg:4326", "epsg:5186")
    assert np.isnan(L[2:]).all() and np.isnan(Wd[2:]).all()                  # the track's first row has no estimate -> none for the track
    np.testing.assert_allclose(L[:2], 100 * 1e-6 * 87_800, rtol=0.02)        # along x: longitude
    np.testing.assert_allclose(Wd[:2], 40 * 1e-6 * 111_000, rtol=0.02)       # along y: latitude


def test_lane_lookup_first_polygon_wins_and_boundary_is_outside():
    import pandas as pd
    from geotrax_amd.georeference import assign_road_section_lane

    seg = pd.DataFrame([["A", 1, 0, 0, 0, 100, 100, 100, 100, 0], ["A", 2, 100, 0, 100, 100, 200, 100, 200, 0], ["B", 1, 50, 50, 50, 150, 150, 150, 150, 50]],
                       columns=["section", "lane", "tlx", "tly", "blx", "bly", "brx", "bry", "trx", "try"])
    x = np.array([10.0, 150.0, 75.0, 100.0, 500.0, 120.0])
    y = np.array([10.0, 10.0, 75.0, 20.0, 500.0, 140.0])
    sec, lane = assign_road_section_lane(x, y, seg)
    assert list(sec[:3]) == ["A", "A", "A"] and list(lane[:3]) == [1, 2, 1]   # (75,75) lies in A1 and B1: file order decides
    assert pd.isna(sec[3]) and pd.isna(lane[3])                              # on the shared edge of A1/A2: 'within' excludes boundaries
    assert pd.isna(sec[4]) and sec[5] == "B" and lane[5] == 1
    assert assign_road_section_lane(x, y, pd.DataFrame()) == (None, None)


def test_save_homography_format(tmp_path):
    from geotrax_amd.georeference import save_homography

    H = np.array([[1.0000000000000002, 2e-7, -3.5], [0.25, 1.0, 1e10], [1e-9, 0.0, 1.0]])
    save_homography(tmp_path / "x" / "h.txt", H, logger)
    txt = (tmp_path / "x" / "h.txt").read_text().strip()
    assert txt.count(",") == 8 and "\n" not in txt
    np.testing.assert_array_equal(np.array([float(t) for t in txt.split(",")]).reshape(3, 3), H)   # %.20g round-trips
    # same layout as the reference's committed golden file
    ref = (GOLD / "U_video_cut_geo_transf.txt").read_text().strip()
    assert ref.count(",") == 8 and "\n" not in ref


def test_independent_oracle_meets_the_reference_known_answers():
    """oracle/georef_ref.py (Snyder's series, cv2.perspectiveTransform's rule) on the reference tests' own pins
    (tests/test_georeference.py:31-63 of the reference) and against the package's Krueger-series projection."""
    from geotrax_amd import georeference as G
    from oracle import georef_ref as R

    x, y = R.apply_homography([1.0, 2.0], [3.0, 4.0], np.eye(3))
    np.testing.assert_array_equal([x, y], [[1.0, 2.0], [3.0, 4.0]])
    x, y = R.apply_homography([1.0, 2.0], [3.0, 4.0], np.array([[1, 0, 10.0], [0, 1, -5.0], [0, 0, 1]]))
    np.testing.assert_array_equal([x, y], [[11.0, 12.0], [-2.0, -1.0]])
    lat, lon = R.ortho2geo(np.array([10.0]), np.array([20.0]), (1.0, 2.0, 0.1, -0.2, 0.0, 0.0))
    np.testing.assert_allclose([lat[0], lon[0]], [2.0 - 4.0, 1.0 + 1.0])
    e, n = R.geo2local(np.array([46.5]), np.array([6.6]), "epsg:4326", "epsg:32631")
    assert abs(e[0] - 776225.4478) < 1e-2 and abs(n[0] - 5155902.1301) < 1e-2           # pyproj's value, the reference's own tolerance
    rng = np.random.default_rng(0)
    lat, lon = rng.uniform(37.2, 37.6, 2000), rng.uniform(126.4, 127.6, 2000)
    a, b = R.geo2local(lat, lon, "EPSG:4326", "EPSG:5186"), G.geo2local(lat, lon, "EPSG:4326", "EPSG:5186")
    assert np.abs(a[0] - b[0]).max() < 1e-4 and np.abs(a[1] - b[1]).max() < 1e-4             # two different series, 0.1 mm
