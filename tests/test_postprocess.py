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
