"""Host post-processing (geotrax_amd/postprocess.py) against (1) vectors produced by the
reference's own functions (tests/golden/postprocess_vectors.npz, made by make_golden.py),
(2) the reference's committed golden output for U_video_cut, and (3) the cases of the
reference's tests/test_extract.py, restated with this build's signatures."""
import gzip
import logging
from pathlib import Path

import numpy as np
import pytest

from geotrax_amd import postprocess as pp

G = Path(__file__).parent / "golden"
logger = logging.getLogger(__name__)

DIM_CFGS = [
    dict(eps=4, r0=1.25, gsd=0.02725, theta_bar=15, tau_c={0: 1.83, 1: 2.85, 2: 1.70, 3: 1.80, -1: 1.70}),
    dict(eps=5, r0=3.0, gsd=0.02725, theta_bar=15, tau_c={-1: 1.0}),
    dict(eps=0, r0=0.6, gsd=0.05, theta_bar=30, tau_c={0: 1.2, -1: 2.0}),
]


@pytest.fixture(scope="module")
def vec():
    return np.load(G / "postprocess_vectors.npz")


def _same(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype
    np.testing.assert_array_equal(a, b)  # NaNs compare equal here


def test_every_stage_equals_the_reference(vec):
    saw_nan = False
    for key, min_len, di in zip(vec["case_keys"], vec["min_len"], vec["dim_cfg_index"]):
        tracks = vec[f"{key}_in"]
        r = pp.remove_short_tracks(tracks.copy(), logger, int(min_len))
        _same(r, vec[f"{key}_short"])
        r = pp.calculate_unique_classes(r)
        _same(r, vec[f"{key}_cls"])
        r = pp.estimate_vehicle_dimensions(r, DIM_CFGS[int(di)], (3840, 2160))
        _same(r, vec[f"{key}_dim"])
        _same(pp.interpolate_tracks(r.copy(), logger, 4), vec[f"{key}_interp"])
        saw_nan |= bool(np.isnan(vec[f"{key}_dim"][:, -1]).any())
    assert saw_nan  # the "no usable observation" path is exercised by the vectors


def test_postprocess_tracks_end_to_end(vec):
    import argparse

    for key, min_len, di in zip(vec["case_keys"], vec["min_len"], vec["dim_cfg_index"]):
        cfg = {'main': {'args': argparse.Namespace(interpolate=True),
                        'extraction': {'min_track_length': int(min_len), 'dimension_estimation': DIM_CFGS[int(di)]},
                        'tracker': {'active': 'botsort', 'botsort': {'track_buffer': 4}}}}
        _same(pp.postprocess_tracks(vec[f"{key}_in"].copy(), cfg, logger, (3840, 2160)), vec[f"{key}_all"])


def test_aggregate_results_equals_the_reference(vec):
    lists = {n: [vec[f"agg_{n}_{k}"] for k in range(2)] for n in ("frame", "id", "bbox", "bbox_stab", "cls", "conf")}
    tracks, transf = pp.aggregate_results(lists["frame"], lists["id"], lists["bbox"], lists["bbox_stab"], lists["cls"],
                                          lists["conf"], [vec["agg_transform_0"]], logger)
    _same(tracks, vec["agg_tracks"])
    _same(transf, vec["agg_transforms"])
    assert tracks.shape == (3, 12) and tracks[2, 1] == 65535       # uint16 id survives, the two -1 rows are gone
    e_tracks, e_transf = pp.aggregate_results([], [], [], [], [], [], [], logger)
    assert e_tracks.shape == tuple(vec["agg_empty_tracks_shape"]) and e_transf.shape == tuple(vec["agg_empty_transforms_shape"])
    # stabilization off: 8-column table
    t8, _ = pp.aggregate_results(lists["frame"], lists["id"], lists["bbox"], [], lists["cls"], lists["conf"], [], logger)
    assert t8.shape == (3, 8)


def test_golden_clip_dimensions_reproduce():
    """data/results-full/U_video_cut.txt: columns 0-11 -> columns 12-13 with the default config
    (reproducible to the %g print quantum, NaN pattern identical); short-track
    removal and the class vote are no-ops on an already post-processed file."""
    with gzip.open(G / "U_video_cut.txt.gz", "rt") as f:
        gold = np.loadtxt(f, delimiter=",")
    assert gold.shape == (19817, 14)
    t12 = gold[:, :12].astype(np.float32)
    assert len(pp.remove_short_tracks(t12.copy(), logger, 3)) == len(t12)
    np.testing.assert_array_equal(pp.calculate_unique_classes(t12.copy())[:, 10], t12[:, 10])
    out = pp.estimate_vehicle_dimensions(t12.copy(), DIM_CFGS[0], (3840, 2160))
    np.testing.assert_array_equal(np.isnan(out[:, 12:]), np.isnan(gold[:, 12:]))
    ok = ~np.isnan(gold[:, 12])
    np.testing.assert_allclose(out[ok, 12:], gold[ok, 12:], rtol=2e-5)


# ---- the reference's own unit-test cases (tests/test_extract.py), same inputs and expectations

def test_remove_short_tracks_cases():
    tracks = np.array([[0, 1], [1, 1], [2, 1], [0, 2], [1, 2]], dtype=np.float32)
    r = pp.remove_short_tracks(tracks, logger, min_length=3)
    assert r.shape == (3, 2) and set(np.unique(r[:, 1])) == {1}
    assert pp.remove_short_tracks(np.empty((0, 2), dtype=np.float32), logger).size == 0


def test_unique_classes_cases():
    t = np.array([[0, 1, 0, 0.90], [1, 1, 1, 0.95], [0, 2, 2, 0.80]], dtype=np.float32)
    r = pp.calculate_unique_classes(t)
    np.testing.assert_array_equal(r[r[:, 1] == 1][:, -2], [1, 1])
    assert r[r[:, 1] == 2][0, -2] == 2
    t = np.array([[0, 1, 7, 0.9], [1, 1, 7, 0.8]], dtype=np.float32)
    np.testing.assert_array_equal(pp.calculate_unique_classes(t)[:, -2], [7, 7])
    # exact tie resolves to the lowest class id
    t = np.array([[0, 1, 3, 0.5], [1, 1, 2, 0.5]], dtype=np.float32)
    np.testing.assert_array_equal(pp.calculate_unique_classes(t)[:, -2], [2, 2])


def test_class_vote_near_ties_follow_the_reference_summation():
    """tests/golden/class_vote_ties.npz (made by the reference's calculate_unique_classes, extract.py:380-401): tables on
    which the winning class depends on the summation order / precision of the votes. The reference sums row by row in
    the table's float32; so must this build."""
    vec = np.load(G / "class_vote_ties.npz")
    n = len([k for k in vec.files if k.startswith("in")])
    assert n >= 10
    flips = 0
    for i in range(n):
        t = vec[f"in{i}"]
        np.testing.assert_array_equal(pp.calculate_unique_classes(t.copy())[:, -2], vec[f"out{i}"], err_msg=f"table {i}")
        exact = [t[t[:, -2] == c, -1].astype(np.float64).sum() for c in (0, 1)]
        flips += int(np.argmax(exact)) != int(vec[f"out{i}"][0])
    assert flips >= 3            # the fixture really contains cases where float64 accumulation picks the other class


def test_dimension_cases():
    cfg = DIM_CFGS[1]
    assert pp.estimate_vehicle_dimensions(np.empty((0, 12), dtype=np.float32), cfg, (1920, 1080)).shape == (0, 14)
    t = np.array([[f, 1, 960, 540, 100, 30, 960, 540, 100, 30, 0, 0.9] for f in range(3)], dtype=np.float32)
    r = pp.estimate_vehicle_dimensions(t, cfg, (1920, 1080))
    assert r.shape == (3, 14)
    np.testing.assert_allclose(r[:, -2], 100.0)
    np.testing.assert_allclose(r[:, -1], 30.0)
    t = np.array([[0, 1, 5, 540, 20, 10, 5, 540, 20, 10, 0, 0.9]], dtype=np.float32)
    r = pp.estimate_vehicle_dimensions(t, cfg, (1920, 1080))
    assert r.shape == (1, 14) and np.isnan(r[0, -2]) and np.isnan(r[0, -1])


def test_interpolation_cases():
    assert pp.interpolate_tracks(np.empty((0, 14), dtype=np.float32), logger, 30).size == 0
    t = np.zeros((3, 14), dtype=np.float32)
    t[:, 0], t[:, 1] = [0, 1, 2], 1
    r = pp.interpolate_tracks(t, logger, 30)
    assert r.shape == (3, 15) and not r[:, 14].any()
    t = np.zeros((2, 14), dtype=np.float32)
    t[0, :2], t[1, :2], t[1, 6] = [0, 1], [3, 1], 3.0
    r = pp.interpolate_tracks(t, logger, 30)
    assert r.shape == (4, 15)
    o = np.argsort(r[:, 0])
    np.testing.assert_allclose(r[o, 6], [0, 1, 2, 3], atol=1e-5)
    np.testing.assert_array_equal(r[o, 14], [0, 1, 1, 0])
    t = np.zeros((2, 14), dtype=np.float32)
    t[0, :2], t[1, :2] = [0, 1], [5, 1]
    r = pp.interpolate_tracks(t, logger, 2)           # gap larger than max_gap stays open
    assert r.shape == (2, 15) and not r[:, 14].any()
    t = np.zeros((4, 14), dtype=np.float32)
    t[:, 0], t[:, 1] = [0, 2, 0, 1], [1, 1, 2, 2]      # two tracks, only the first has a gap
    r = pp.interpolate_tracks(t, logger, 30)
    assert r.shape == (5, 15) and (r[:, 1] == 1).sum() == 3 and (r[:, 1] == 2).sum() == 2


# ---- the long-video forms (round 4): same results as the row-by-row / mask-per-track forms the reference uses -------------------

def _long_table(rng, n_tracks=260, frames=900, dtype=np.float32):
    rows = []
    for tid in rng.permutation(np.arange(1, n_tracks + 1)):
        f0 = int(rng.integers(0, frames - 20))
        fr = np.arange(f0, min(frames, f0 + int(rng.integers(3, 400))))
        if len(fr) > 6 and rng.random() < 0.5:
            fr = np.delete(fr, rng.integers(1, len(fr) - 1, int(rng.integers(1, 5))))
            if rng.random() < 0.3:
                fr = np.concatenate([fr[: len(fr) // 2], fr[len(fr) // 2:] + int(rng.integers(20, 60))])   # a gap beyond the track buffer
        moving = rng.random() < 0.7
        sx, sy = (rng.normal(0, 3), rng.normal(0, 3)) if moving else (0.0, 0.0)
        x = rng.uniform(-20, 3860) + np.cumsum(rng.normal(sx, 0.7, len(fr)))
        y = rng.uniform(-20, 2180) + np.cumsum(rng.normal(sy, 0.7, len(fr)))
        w, h = rng.uniform(30, 110) + rng.normal(0, 1.5, len(fr)), rng.uniform(15, 60) + rng.normal(0, 1.5, len(fr))
        rows.append(np.stack([fr, np.full(len(fr), tid), x, y, w, h, x + rng.normal(0, 2, len(fr)), y + rng.normal(0, 2, len(fr)), w, h,
                              rng.integers(0, 4, len(fr)), rng.uniform(0.3, 0.95, len(fr))], 1))
    t = np.concatenate(rows)
    return t[np.argsort(t[:, 0], kind="stable")].astype(dtype)


def _dimensions_row_by_row(tracks, dim_cfg, frame_wh):
    """estimate_vehicle_dimensions exactly as extract.py:404-484 walks it: NumPy scalars, one row per Python iteration."""
    from geotrax_amd.postprocess import _CARDINALS

    w_img, h_img = frame_wh
    eps = dim_cfg['eps']
    vis = (tracks[:, 2] - tracks[:, 4] / 2 > eps) & (tracks[:, 3] - tracks[:, 5] / 2 > eps)
    vis &= (tracks[:, 2] + tracks[:, 4] / 2 < w_img - 1 - eps) & (tracks[:, 3] + tracks[:, 5] / 2 < h_img - 1 - eps)
    valid = tracks[vis]
    radius, theta, tau_c = dim_cfg['r0'] / dim_cfg['gsd'], np.deg2rad(dim_cfg['theta_bar']), dim_cfg['tau_c']
    est = {}
    for track_id in np.unique(valid[:, 1]):
        t = valid[valid[:, 1] == track_id]
        length, width = np.maximum(t[:, 4], t[:, 5]), np.minimum(t[:, 4], t[:, 5])
        xc, yc = t[:, 6], t[:, 7]
        keep, moved = np.zeros(len(t), bool), False
        prev, xp, yp = 0, xc[0], yc[0]
        for k in range(1, len(t)):
            dx, dy = xc[k] - xp, yc[k] - yp
            if np.sqrt(dx ** 2 + dy ** 2) >= radius:
                moved = True
                az = np.arctan2(-dy, dx)
                xp, yp = xc[k], yc[k]
                if np.any(np.abs(az - _CARDINALS) <= theta):
                    keep[prev:k] = True
                prev = k
        if not moved:
            keep = length >= width * tau_c.get(int(t[0, 10]), tau_c[-1])
        est[int(track_id)] = (np.percentile(length[keep], 25) if keep.any() else np.nan, np.percentile(width[keep], 25) if keep.any() else np.nan)
    out = np.append(tracks, np.zeros((len(tracks), 2)), axis=1)
    out[:, -2] = [est.get(int(i), (np.nan, np.nan))[0] for i in out[:, 1]]
    out[:, -1] = [est.get(int(i), (np.nan, np.nan))[1] for i in out[:, 1]]
    return out


DIM_CFG = {"gsd": 0.02725, "eps": 4, "r0": 1.25, "theta_bar": 15, "tau_c": {0: 1.83, 1: 2.85, 2: 1.70, 3: 1.80, -1: 1.70}}


@pytest.mark.parametrize("dtype,seed", [(np.float32, 0), (np.float32, 1), (np.float64, 2)])
def test_dimension_estimation_equals_the_row_by_row_walk(dtype, seed):
    """float32 tables (what aggregate_results makes) go through gtx_track_anchor_walk: glibc's powf(x, 2) -- NumPy's float32 scalar
    `x ** 2`, not x * x --, sqrtf, atan2f, radius compared in float32; float64 tables keep the scalar loop. Same table out, NaNs included."""
    from geotrax_amd.postprocess import estimate_vehicle_dimensions

    t = _long_table(np.random.default_rng(seed), dtype=dtype)
    got = estimate_vehicle_dimensions(t.copy(), DIM_CFG, (3840, 2160))
    want = _dimensions_row_by_row(t.copy(), DIM_CFG, (3840, 2160))
    assert got.dtype == want.dtype and got.shape == want.shape
    assert np.array_equal(got, want, equal_nan=True)
    assert np.isfinite(got[:, -1]).mean() > 0.2 and np.isnan(got[:, -1]).any()        # both outcomes occur


def test_the_anchor_walk_of_the_library_is_the_numpy_scalar_walk_on_hard_cases():
    """Steps of exactly the radius, of one float32 ulp less and more, azimuths on the 15 degree boundary, huge and tiny coordinates."""
    from geotrax_amd.postprocess import _anchor_walk, _anchor_walk_all

    rng = np.random.default_rng(5)
    radius, theta = 1.25 / 0.02725, np.deg2rad(15)
    r32 = np.float32(radius)
    tracks = []
    for _ in range(400):
        n = int(rng.integers(2, 60))
        step = rng.choice([r32, np.nextafter(r32, np.float32(0)), np.nextafter(r32, np.float32(1e9)), np.float32(radius * 0.5), np.float32(radius * 3)], n)
        ang = rng.choice([0, 15, 14.999999, 15.000001, 75, 90, 105, 165, 180, -90, -165, 37.3], n) * np.pi / 180 + rng.choice([0, 1e-7, -1e-7], n)
        scale = rng.choice([1.0, 1.0, 1e3, 1e-2])
        x = np.cumsum(step * np.cos(ang)).astype(np.float32) * np.float32(scale) + np.float32(rng.uniform(0, 4000))
        y = np.cumsum(-step * np.sin(ang)).astype(np.float32) * np.float32(scale) + np.float32(rng.uniform(0, 2000))
        tracks.append((x, y))
    valid = np.zeros((sum(len(x) for x, _ in tracks), 12), np.float32)
    at = 0
    for k, (x, y) in enumerate(tracks):
        valid[at:at + len(x), 1], valid[at:at + len(x), 6], valid[at:at + len(x), 7] = k, x, y
        at += len(x)
    order = np.arange(len(valid))
    bounds = np.flatnonzero(np.diff(valid[:, 1])) + 1
    keep, moved = _anchor_walk_all(valid, order, bounds, 6, 7, radius, theta)
    at = 0
    for k, (x, y) in enumerate(tracks):
        k_ref, m_ref = _anchor_walk(x, y, radius, theta)
        assert np.array_equal(keep[at:at + len(x)], k_ref) and bool(moved[k]) == m_ref, k
        at += len(x)
    assert moved.any() and not moved.all() and keep.any()


def test_interpolate_tracks_groups_like_a_mask_per_track():
    from geotrax_amd.postprocess import interpolate_tracks

    def mask_per_track(tracks, max_gap):                       # extract.py:309-359 as written
        new_rows = []
        for track_id in np.unique(tracks[:, 1]):
            t = tracks[tracks[:, 1] == track_id]
            t = t[np.argsort(t[:, 0])]
            frames = t[:, 0].astype(int)
            gaps = np.diff(frames)
            for i in np.flatnonzero(gaps > 1):
                gap = int(gaps[i])
                if gap > max_gap:
                    continue
                for step in range(1, gap):
                    alpha = step / gap
                    row = t[i] * (1.0 - alpha) + t[i + 1] * alpha
                    row[0] = float(frames[i] + step)
                    new_rows.append(row)
        out = np.concatenate([tracks, np.zeros((len(tracks), 1), dtype=tracks.dtype)], axis=1)
        if new_rows:
            extra = np.array(new_rows, dtype=tracks.dtype)
            out = np.concatenate([out, np.concatenate([extra, np.ones((len(extra), 1), dtype=tracks.dtype)], axis=1)], axis=0)
            out = out[np.lexsort((out[:, 0], out[:, 1]))]
        return out

    log = logging.getLogger("t")
    for seed, dtype in [(0, np.float32), (3, np.float64)]:
        t = _long_table(np.random.default_rng(seed), dtype=dtype)
        got, want = interpolate_tracks(t.copy(), log, 30), mask_per_track(t.copy(), 30)
        assert got.dtype == want.dtype and np.array_equal(got, want) and (got[:, -1] == 1).any()


def test_kinematics_and_gap_filling_equal_the_point_by_point_forms():
    from geotrax_amd import georeference as gr

    def fill_point_by_point(frames, x, y):                      # georeference.py:738-766 as written
        xs, ys, present = [x[0]], [y[0]], [1]
        for i in range(1, len(frames)):
            gap = int(frames[i] - frames[i - 1])
            if gap > 1:
                dx, dy = (x[i] - x[i - 1]) / gap, (y[i] - y[i - 1]) / gap
                for step in range(1, gap):
                    xs.append(x[i - 1] + step * dx)
                    ys.append(y[i - 1] + step * dy)
                    present.append(0)
            xs.append(x[i]); ys.append(y[i]); present.append(1)
        return xs, ys, np.nonzero(present)[0]

    def kinematics_mask_per_track(track_ids, frame_num, x, y, vis, fps, ftype, ksize, interp):   # :705-735 as written
        n = len(track_ids)
        speed, accel = np.full(n, np.nan), np.full(n, np.nan)
        for tid in np.unique(track_ids):
            idx = np.where(track_ids == tid)[0]
            use = np.asarray(vis)[idx] & (np.asarray(interp)[idx] == 0)
            if use.sum() < 3:
                continue
            xs, ys, present = fill_point_by_point(frame_num[idx][use], x[idx][use], y[idx][use])
            v = gr.apply_filter(gr.compute_speed(xs, ys, fps), ksize, ftype)
            a = gr.compute_acceleration(v, fps)
            v = np.insert(v * 3.6, 0, np.nan)
            a = np.insert(a, 0, [np.nan] * 2)
            speed[idx[use]], accel[idx[use]] = v[present], a[present]
        return speed, accel

    rng = np.random.default_rng(7)
    for _ in range(200):                                        # the gap filler alone: single points, no gaps, long gaps, float frame numbers
        n = int(rng.integers(1, 40))
        frames = np.cumsum(rng.choice([1, 1, 1, 2, 3, 9], n)).astype(rng.choice([np.int64, np.float64]))
        x, y = rng.standard_normal(n) * 100, rng.standard_normal(n) * 100
        gx, gy, gp = gr.interpolate_missing_points(frames, x, y)
        wx, wy, wp = fill_point_by_point(frames, x, y)
        assert np.array_equal(np.asarray(gx), np.asarray(wx)) and np.array_equal(np.asarray(gy), np.asarray(wy)) and np.array_equal(gp, wp)
    t = _long_table(np.random.default_rng(11), n_tracks=150, frames=600, dtype=np.float64)
    t = t[np.lexsort((t[:, 0], t[:, 1]))]
    ids, fr = t[:, 1].astype(np.int64), t[:, 0].astype(np.int64)
    x, y = t[:, 2] * 0.03, t[:, 3] * 0.03
    vis, interp = rng.random(len(t)) < 0.9, (rng.random(len(t)) < 0.05).astype(int)
    for ftype, ks in [("gaussian", 14), ("savgol", 7)]:
        got = gr.compute_kinematics(ids, fr, x, y, vis, 29.97, ftype, ks, interp)
        want = kinematics_mask_per_track(ids, fr, x, y, vis, 29.97, ftype, ks, interp)
        assert np.array_equal(got[0], want[0], equal_nan=True) and np.array_equal(got[1], want[1], equal_nan=True)
        assert np.isfinite(got[0]).mean() > 0.5
