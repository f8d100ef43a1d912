"""OC-SORT (tracker.ocsort of the reference's config, default.yaml:391-404): the C++ tracker behind gtx_tracker_*
(csrc/ocsort.cpp) against oracle/ocsort_ref.py on seeded detection streams with births, deaths, occlusions, low-score
detections, crossings and empty frames, plus the properties the algorithm promises. Runs on CPU.
Parity unpinned against ultralytics' port / the authors' package (neither is importable here): see the oracle's header."""
import numpy as np
import pytest

from test_tracker import _stream


def _run(tracker, stream):
    out = []
    for xyxy, conf, cls in stream:
        out.append(tracker(xyxy, conf, cls))
    return out


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("kw", [{}, {"use_byte": True}, {"delta_t": 1, "inertia": 0.4, "track_buffer": 12, "match_thresh": 0.7}])
def test_cpp_ocsort_equals_oracle(seed, kw):
    from geotrax_amd.tracker import Tracker
    from oracle.ocsort_ref import OCSortRef

    ref, trk = OCSortRef(**kw), Tracker("ocsort", **kw)
    for t, (xyxy, conf, cls) in enumerate(_stream(seed, n_frames=90, p_miss=0.12)):
        want = ref.update(xyxy, conf, cls)
        got_xyxy, got_id, got_score, got_cls, got_idx = trk.update(xyxy, conf, cls)
        assert len(want) == len(got_id), f"frame {t}"
        if len(want):
            np.testing.assert_array_equal(want[:, 4].astype(np.int32), got_id, err_msg=f"frame {t}")
            np.testing.assert_array_equal(want[:, 7].astype(np.int32), got_idx)
            np.testing.assert_array_equal(want[:, 6].astype(np.int32), got_cls)
            np.testing.assert_allclose(got_xyxy, want[:, :4], rtol=0, atol=1e-4)
            np.testing.assert_allclose(got_score, want[:, 5], rtol=0, atol=1e-6)


def test_ocsort_keeps_identities_through_an_occlusion():
    """20 objects on straight lines, 6 of them hidden for 8 frames: every object ends with the id it started with (the
    observation-centric re-update and the velocity-direction term are what make the re-association unambiguous)."""
    from geotrax_amd.tracker import Tracker

    rng = np.random.default_rng(3)
    n, T = 20, 70
    p0 = np.stack([np.linspace(200, 3600, n), rng.uniform(300, 1800, n)], 1)
    v = np.stack([rng.uniform(-3, 3, n), rng.uniform(4, 9, n) * rng.choice([-1, 1], n)], 1)
    wh = rng.uniform(40, 90, (n, 2))
    trk = Tracker("ocsort")
    ids = {}
    for t in range(T):
        c = p0 + v * t
        boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        vis = np.ones(n, bool)
        if 30 <= t < 38:
            vis[:6] = False
        xyxy, tid, _, _, idx = trk.update(boxes[vis], np.full(vis.sum(), 0.8, np.float32), np.zeros(vis.sum(), np.int32))
        objs = np.where(vis)[0][idx]
        for o, i in zip(objs, tid):
            ids.setdefault(int(o), set()).add(int(i))
        if t >= 3:
            assert len(tid) == vis.sum() or 38 <= t < 41          # re-found tracks report again after min_hits frames
    assert all(len(s) == 1 for s in ids.values()), ids
    assert len({next(iter(s)) for s in ids.values()}) == n


def test_ocsort_contract_details():
    from geotrax_amd.tracker import Tracker

    trk = Tracker("ocsort")
    box = np.array([[100, 100, 160, 140]], np.float32)
    # frames 1..min_hits report a new track at once (ids from 1, in detection order); boxes are the observations themselves
    xyxy, tid, score, cls, idx = trk.update(np.concatenate([box, box + 400]), np.array([0.9, 0.6], np.float32), np.array([2, 1], np.int32))
    assert sorted(tid.tolist()) == [1, 2] and sorted(idx.tolist()) == [0, 1]
    assert set(map(tuple, xyxy.tolist())) == {tuple(box[0].tolist()), tuple((box + 400)[0].tolist())}
    # detections at or below track_high_thresh never start or feed a track without use_byte
    for _ in range(5):
        xyxy, tid, *_ = trk.update(box + 900, np.array([0.2], np.float32), np.array([0], np.int32))
        assert len(tid) == 0
    # an empty frame is legal and ages the tracks; after track_buffer misses they are gone and ids are not reused
    for _ in range(40):
        trk.update(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32))
    for k in range(4):                                    # past frame min_hits a new track reports once it has min_hits hits
        xyxy, tid, *_ = trk.update(box, np.array([0.9], np.float32), np.array([2], np.int32))
        assert tid.tolist() == ([] if k < 3 else [3])
    trk.reset()
    assert trk.update(box, np.array([0.9], np.float32), np.array([2], np.int32))[1].tolist() == [1]


def test_config_selects_ocsort(tmp_path):
    """tracker.active: ocsort resolves through the config surface to the C++ OC-SORT with the block's parameters; the
    tracker this build does not have says so (FastTracker and TrackTrack are built since round 5: tests/test_fasttrack.py, test_tracktrack.py)."""
    from geotrax_amd.model import YOLO
    from geotrax_amd.tracker import Tracker

    m = YOLO.__new__(YOLO)
    m._gmc_method = m._gmc = None
    t = m._make_tracker({"tracker_type": "ocsort", "track_high_thresh": 0.3, "delta_t": 2, "inertia": 0.1, "use_byte": True, "match_thresh": 0.75})
    assert isinstance(t, Tracker) and m._gmc_method is None
    with pytest.raises(NotImplementedError):
        m._make_tracker({"tracker_type": "strongsort"})
    assert isinstance(m._make_tracker({"tracker_type": "fasttrack", "occ_cover_thresh": 0.6}), Tracker)
    t = m._make_tracker({"tracker_type": "tracktrack", "gmc_method": "sparseOptFlow", "tai_thr": 0.5})   # round 5: tests/test_tracktrack.py
    assert isinstance(t, Tracker) and m._gmc_method == "sparseOptFlow"
    assert m._make_tracker({"tracker_type": "tracktrack", "with_reid": True}).with_reid        # round 6: `model: auto` (tests/test_tracktrack.py)
    with pytest.raises(NotImplementedError):
        m._make_tracker({"tracker_type": "tracktrack", "with_reid": True, "model": "osnet_x0_25.pt"})   # a separate ReID network is not built


@pytest.mark.parametrize("kind", ["bytetrack", "botsort", "ocsort"])
def test_batch_replay_equals_frame_by_frame_updates(kind):
    """gtx_tracker_replay (rank 0's sequential half of a frame-sharded run, one C call per gathered run) returns exactly
    what the per-frame gtx_tracker_update calls return on the same packed records, camera-motion warps included."""
    from geotrax_amd.distributed import pack_frame_record, unpack_frame_gmc, unpack_frame_record
    from geotrax_amd.tracker import Tracker

    max_det, rng = 200, np.random.default_rng(4)
    recs = []
    for xyxy, conf, cls in _stream(2, n_obj=60, n_frames=50, p_miss=0.1):
        g = np.array([[1 + 1e-4 * rng.normal(), 0, 0.3 * rng.normal()], [0, 1, 0.2 * rng.normal()]]) if rng.random() < 0.9 else None
        recs.append(pack_frame_record(max_det, xyxy, conf, cls, None, gmc=g, with_gmc=True))
    recs = np.stack(recs)
    a, b = Tracker(kind), Tracker(kind)
    per, xy, tid, sc, cl, idx = a.replay(recs, max_det, with_gmc=True)
    o = 0
    for f, rec in enumerate(recs):
        x, c, k, _ = unpack_frame_record(rec, max_det)
        bx, ids, s2, c2, i2 = b.update(x, c, k, gmc=unpack_frame_gmc(rec))
        assert per[f] == len(ids)
        np.testing.assert_array_equal(tid[o:o + per[f]], ids)
        np.testing.assert_array_equal(xy[o:o + per[f]], bx)
        np.testing.assert_array_equal(sc[o:o + per[f]], s2)
        np.testing.assert_array_equal(cl[o:o + per[f]], c2)
        np.testing.assert_array_equal(idx[o:o + per[f]], i2)
        o += per[f]
    assert o == len(tid) > 500
    with pytest.raises(Exception):
        a.replay(recs[:, :-1], max_det, with_gmc=True)          # a stride that does not match max_det is refused


# ---------------------------------------------------------------------------------------------------------------------
# tracker.active: deepocsort (default.yaml:406-427) without the appearance branch = OC-SORT + camera-motion compensation


def _camera_path(seed, n_frames, step=18.0, rot=2e-3, zoom=1e-3):
    """Per-frame 2x3 warps previous frame -> this frame (pan with a little rotation and zoom) and their running product."""
    rng = np.random.default_rng(seed)
    warps, total = [], [np.eye(3)]
    for _ in range(n_frames):
        a, s = rng.normal(0, rot), 1 + rng.normal(0, zoom)
        g = np.array([[s * np.cos(a), -s * np.sin(a), rng.normal(0, step)], [s * np.sin(a), s * np.cos(a), rng.normal(0, step)]])
        warps.append(g)
        total.append(np.vstack([g, [0, 0, 1]]) @ total[-1])
    return warps, total[1:]


def _moved(xyxy, T):
    """Boxes seen through the camera transform T (3x3, affine): corners mapped, axis-aligned again."""
    if len(xyxy) == 0:
        return xyxy
    p = np.stack([xyxy[:, [0, 1]], xyxy[:, [2, 3]]], 1).astype(np.float64)      # [n, 2, 2]
    q = p @ T[:2, :2].T + T[:2, 2]
    return np.concatenate([q.min(1), q.max(1)], 1).astype(np.float32)


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("kw", [{}, {"use_byte": True, "track_high_thresh": 0.3, "new_track_thresh": 0.3}])
def test_cpp_deepocsort_equals_oracle(seed, kw):
    from geotrax_amd.tracker import Tracker
    from oracle.ocsort_ref import OCSortRef

    ref, trk = OCSortRef(cmc=True, **kw), Tracker("deepocsort", **kw)
    warps, total = _camera_path(seed + 10, 90)
    rows = 0
    for t, (xyxy, conf, cls) in enumerate(_stream(seed, n_frames=90, p_miss=0.12)):
        xyxy = _moved(xyxy, total[t])
        want = ref.update(xyxy, conf, cls, warps[t])
        got_xyxy, got_id, got_score, got_cls, got_idx = trk.update(xyxy, conf, cls, warps[t])
        assert len(want) == len(got_id), f"frame {t}"
        if len(want):
            np.testing.assert_array_equal(want[:, 4].astype(np.int32), got_id, err_msg=f"frame {t}")
            np.testing.assert_array_equal(want[:, 7].astype(np.int32), got_idx)
            np.testing.assert_allclose(got_xyxy, want[:, :4], rtol=0, atol=2e-3)
        rows += len(want)
    assert rows > 1000


def test_camera_motion_compensation_keeps_the_identities_a_panning_camera_breaks():
    """A camera that jumps ~25 px per frame over 40-90 px objects: with the warps handed in, Deep OC-SORT ends with the
    identity partition of the same scene seen by a still camera; plain OC-SORT on the moving view fragments it. On the still
    scene the two trackers are the same tracker."""
    from geotrax_amd.tracker import Tracker

    warps, total = _camera_path(5, 60, step=25.0)
    frames = list(_stream(4, n_obj=30, n_frames=60, p_miss=0.05, p_low=0.0))

    def partition(tracker, moving, with_warp):
        seen = {}
        for t, (xyxy, conf, cls) in enumerate(frames):
            b = _moved(xyxy, total[t]) if moving else xyxy
            _, tid, _, _, idx = tracker.update(b, conf, cls, warps[t] if with_warp else None)
            for i, d in zip(tid, idx):
                seen.setdefault(int(i), []).append((t, int(d)))
        return sorted(tuple(v) for v in seen.values())

    still = partition(Tracker("ocsort"), False, False)
    assert partition(Tracker("deepocsort"), False, False) == still
    compensated = partition(Tracker("deepocsort"), True, True)
    plain = partition(Tracker("ocsort"), True, False)
    assert compensated == still
    assert len(plain) > len(still) + 5


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_cpp_deepocsort_with_reid_equals_oracle(seed):
    """Deep OC-SORT's appearance branch (`with_reid: true, model: auto`, default.yaml:420-425; csrc/ocsort.cpp) against
    oracle/ocsort_ref.py: adaptive-weighted similarity term in the first association, dynamic-alpha EMA of the track vectors, also
    seen through a panning camera with the warps handed in; the term does change associations on this stream."""
    from geotrax_amd.tracker import Tracker
    from oracle.ocsort_ref import OCSortRef
    from test_tracker import _stream_with_feats

    kw = dict(with_reid=True, proximity_thresh=0.3, appearance_thresh=0.6, alpha_fixed_emb=0.9, track_high_thresh=0.3, new_track_thresh=0.3)
    ref, trk, plain = OCSortRef(cmc=True, **kw), Tracker("deepocsort", **kw), OCSortRef(cmc=True, **{**kw, "with_reid": False})
    assert trk.with_reid
    warps, total = _camera_path(seed + 20, 90, step=6.0)
    rows, differs = 0, False
    for t, (xyxy, conf, cls, feats) in enumerate(_stream_with_feats(seed, n_obj=70, jitter=5.0, p_miss=0.15, n_frames=90)):
        xyxy = _moved(xyxy, total[t])
        want = ref.update(xyxy, conf, cls, warps[t], feats=feats)
        p = plain.update(xyxy, conf, cls, warps[t])
        got_xyxy, got_id, got_score, got_cls, got_idx = trk.update(xyxy, conf, cls, warps[t], feats=feats)
        assert len(want) == len(got_id), f"frame {t}"
        if len(want):
            np.testing.assert_array_equal(want[:, 4].astype(np.int32), got_id, err_msg=f"frame {t}")
            np.testing.assert_array_equal(want[:, 7].astype(np.int32), got_idx)
            np.testing.assert_allclose(got_xyxy, want[:, :4], rtol=0, atol=2e-3)
        differs |= len(p) != len(want) or not np.array_equal(p[:, [4, 7]], want[:, [4, 7]])
        rows += len(want)
    assert rows > 1000 and differs


def test_config_selects_deepocsort_and_its_appearance_branch():
    from geotrax_amd.model import YOLO

    m = YOLO.__new__(YOLO)
    m._gmc = m._gmc_method = None
    trk = m._make_tracker({"tracker_type": "deepocsort", "track_high_thresh": 0.3, "new_track_thresh": 0.3, "gmc_method": "sparseOptFlow",
                           "with_reid": False, "delta_t": 3, "inertia": 0.2, "use_byte": False, "alpha_fixed_emb": 0.95})
    assert m._gmc_method == "sparseOptFlow" and trk.update(np.zeros((0, 4)), np.zeros(0), np.zeros(0, np.int32))[1].size == 0
    m._make_tracker({"tracker_type": "deepocsort", "gmc_method": "none"})
    assert m._gmc_method is None
    assert m._make_tracker({"tracker_type": "deepocsort", "with_reid": True, "model": "auto", "appearance_thresh": 0.9}).with_reid
    with pytest.raises(NotImplementedError):
        m._make_tracker({"tracker_type": "deepocsort", "with_reid": True, "model": "osnet_x0_25.pt"})


def test_ocsort_choice_is_announced_and_min_hits_is_a_config_key(caplog):
    """tracker.active: ocsort | deepocsort runs an OC-SORT whose mapping onto ultralytics' config keys is unpinned and
    which ignores fuse_score -- the run says so once, and min_hits (the authors' probation length) can be set from the config."""
    import logging

    from geotrax_amd.model import YOLO

    m = YOLO.__new__(YOLO)
    with caplog.at_level(logging.WARNING):
        t = m._make_tracker({"tracker_type": "ocsort", "fuse_score": True, "min_hits": 1, "track_high_thresh": 0.3})
        m._make_tracker({"tracker_type": "ocsort"})
    msgs = [r.message for r in caplog.records if "OC-SORT" in r.message]
    assert len(msgs) == 1 and "fuse_score" in msgs[0] and "min_hits" in msgs[0]
    # min_hits = 1: a new track is reported on its second frame already (3: only after three consecutive hits)
    box = np.array([[100, 100, 160, 140]], np.float32)
    seen = [len(t.update(box + 2 * k, np.array([0.9], np.float32), np.array([0], np.int32))[1]) for k in range(3)]
    t3 = m._make_tracker({"tracker_type": "ocsort", "track_high_thresh": 0.3})
    seen3 = [len(t3.update(box + 2 * k, np.array([0.9], np.float32), np.array([0], np.int32))[1]) for k in range(3)]
    assert sum(seen) >= sum(seen3) and seen[1] == 1
