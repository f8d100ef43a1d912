"""Host tracker (C++ ByteTrack / BoT-SORT association behind gtx_tracker_*) against
oracle/bytetrack_ref.py on seeded detection streams with births, deaths, occlusions, low-score
detections, crossings and empty frames. Runs on CPU: the tracker never touches the GPU."""
import numpy as np
import pytest


def _stream(seed, n_obj=40, n_frames=60, w=3840, h=2160, p_miss=0.08, p_low=0.1, jitter=1.0):
    rng = np.random.default_rng(seed)
    pos = np.stack([rng.uniform(100, w - 100, n_obj), rng.uniform(100, h - 100, n_obj)], 1)
    vel = rng.uniform(-6, 6, (n_obj, 2))
    size = np.stack([rng.uniform(40, 160, n_obj), rng.uniform(25, 70, n_obj)], 1)
    birth = rng.integers(0, n_frames // 3, n_obj) * (rng.random(n_obj) < 0.4)
    death = n_frames - rng.integers(0, n_frames // 3, n_obj) * (rng.random(n_obj) < 0.4)
    base_conf = rng.uniform(0.35, 0.95, n_obj)
    for t in range(n_frames):
        if t in (17, 18):  # two empty frames
            yield np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32)
            continue
        rows, cf, cl = [], [], []
        for k in range(n_obj):
            if not (birth[k] <= t < death[k]) or rng.random() < p_miss:
                continue
            c = pos[k] + vel[k] * t + rng.normal(0, jitter, 2)
            s = size[k] * (1 + rng.normal(0, 0.02, 2))
            rows.append([c[0] - s[0] / 2, c[1] - s[1] / 2, c[0] + s[0] / 2, c[1] + s[1] / 2])
            conf = base_conf[k] + rng.normal(0, 0.03)
            if rng.random() < p_low:
                conf = rng.uniform(0.11, 0.24)
            cf.append(np.clip(conf, 0.05, 0.99))
            cl.append(k % 4)
        for _ in range(rng.integers(0, 3)):  # false positives
            c = rng.uniform([50, 50], [w - 50, h - 50])
            rows.append([c[0] - 30, c[1] - 20, c[0] + 30, c[1] + 20])
            cf.append(rng.uniform(0.26, 0.5))
            cl.append(int(rng.integers(0, 4)))
        order = np.argsort(-np.asarray(cf), kind="stable")  # NMS output order: by confidence
        yield (np.asarray(rows, np.float32).reshape(-1, 4)[order], np.asarray(cf, np.float32)[order],
               np.asarray(cl, np.int32)[order])


@pytest.mark.parametrize("kind", ["bytetrack", "botsort"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tracker_matches_oracle(kind, seed):
    from geotrax_amd.tracker import Tracker
    from oracle.bytetrack_ref import ByteTrackRef

    trk = Tracker(kind)
    ref = ByteTrackRef(botsort=(kind == "botsort"))
    n_rows = 0
    rng = np.random.default_rng(seed + 100)
    for t, (xyxy, conf, cls) in enumerate(_stream(seed)):
        gmc = None
        if kind == "botsort":
            gmc = np.array([[1 + 1e-4 * rng.normal(), 1e-4 * rng.normal(), 0.3 * rng.normal()],
                            [1e-4 * rng.normal(), 1 + 1e-4 * rng.normal(), 0.3 * rng.normal()]])
        b, i, s, c, d = trk.update(xyxy, conf, cls, gmc)
        r = ref.update(xyxy, conf, cls, gmc)
        assert len(i) == len(r), f"frame {t}: {len(i)} tracks vs {len(r)}"
        np.testing.assert_array_equal(i, r[:, 4].astype(np.int32), err_msg=f"frame {t} ids")
        np.testing.assert_array_equal(d, r[:, 7].astype(np.int32), err_msg=f"frame {t} detection index")
        np.testing.assert_array_equal(c, r[:, 6].astype(np.int32))
        np.testing.assert_allclose(s, r[:, 5], rtol=0, atol=1e-7)
        # Kalman posterior boxes: f64 state, two independent op orders -> agree far below 0.01 px
        np.testing.assert_allclose(b, r[:, :4], rtol=0, atol=2e-3)
        n_rows += len(i)
    assert n_rows > 1000


def test_tracker_dense_overlaps_match_oracle():
    """120 objects on an 800x500 patch: association graphs with large connected components, the
    regime where the sparse LAP has to agree with lapjv's dense extended problem."""
    from geotrax_amd.tracker import Tracker
    from oracle.bytetrack_ref import ByteTrackRef

    trk, ref = Tracker("bytetrack"), ByteTrackRef()
    for t, (xyxy, conf, cls) in enumerate(_stream(11, n_obj=120, n_frames=25, w=800, h=500, jitter=2.0)):
        b, i, s, c, d = trk.update(xyxy, conf, cls)
        r = ref.update(xyxy, conf, cls)
        np.testing.assert_array_equal(i, r[:, 4].astype(np.int32), err_msg=f"frame {t}")
        np.testing.assert_array_equal(d, r[:, 7].astype(np.int32))
        np.testing.assert_allclose(b, r[:, :4], atol=2e-3)


def test_tracker_first_frame_ids_follow_detection_order():
    """Golden track file property (data/results-pixel/U_video_cut.txt): on the first frame ids are
    1..N in detection order (descending confidence) and boxes equal the detections."""
    from geotrax_amd.tracker import Tracker

    xyxy, conf, cls = next(iter(_stream(5)))
    b, i, s, c, d = Tracker("bytetrack").update(xyxy, conf, cls)
    hi = conf >= 0.25
    np.testing.assert_array_equal(i, np.arange(1, hi.sum() + 1))
    np.testing.assert_array_equal(d, np.nonzero(hi)[0])
    np.testing.assert_allclose(b, xyxy[hi], atol=1e-3)


def test_tracker_reset_and_empty_input():
    from geotrax_amd.tracker import Tracker

    trk = Tracker("bytetrack")
    out = trk.update(np.zeros((0, 4)), np.zeros(0), np.zeros(0, np.int32))
    assert all(len(o) == 0 for o in out)
    frames = list(_stream(3, n_frames=5))
    a = [Tracker("bytetrack").update(*frames[0])[1]]
    fresh = Tracker("bytetrack")
    a = [fresh.update(*f)[1] for f in frames]
    trk.reset()
    b = [trk.update(*f)[1] for f in frames]
    # after reset the id counter and the frame counter start again: identical to a fresh tracker
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_empty_frames_advance_the_tracker_clock():
    """ultralytics >= 8.4.80 calls tracker.update on frames without detections too: a gap longer than track_buffer
    removes every track (new ids afterwards), a short gap lets the lost tracks be re-found with their old ids.
    Same behaviour in the oracle and the C++ tracker."""
    from geotrax_amd.tracker import Tracker
    from oracle.bytetrack_ref import ByteTrackRef

    frames = list(_stream(7, n_obj=12, n_frames=8, p_miss=0.0, p_low=0.0))
    empty = (np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32))
    for gap, expect_same in ((5, True), (40, False)):
        trk, ref = Tracker("bytetrack"), ByteTrackRef()
        for f in frames[:4]:
            before = trk.update(*f)[1]
            ref.update(*f)
        for _ in range(gap):
            assert len(trk.update(*empty)[1]) == 0 and len(ref.update(*empty)) == 0
        after, r = trk.update(*frames[4])[1], ref.update(*frames[4])
        np.testing.assert_array_equal(after, r[:, 4].astype(np.int32))
        if expect_same:
            assert set(after) <= set(before) and len(after) > 0
        else:                                   # every old track timed out: the frame after the gap opens new, unconfirmed tracks
            after2 = trk.update(*frames[5])[1]
            assert len(after) == 0 and len(after2) > 0 and min(after2) > max(before)


def test_tracker_rejects_unknown_type():
    from geotrax_amd.tracker import Tracker

    with pytest.raises(NotImplementedError):
        Tracker("strongsort")


def _stream_with_feats(seed, dim=128, **kw):
    """_stream plus an appearance vector per detection: the object's own direction + noise (false positives: random ones)."""
    rng = np.random.default_rng(seed + 500)
    protos = {}
    for xyxy, conf, cls in _stream(seed, **kw):
        feats = np.zeros((len(conf), dim), np.float32)
        for j in range(len(conf)):
            key = (int(cls[j]), round(float(xyxy[j, 2] - xyxy[j, 0]) / 8))          # size bucket + class: a stable stand-in for identity
            if key not in protos:
                protos[key] = rng.standard_normal(dim)
            feats[j] = (protos[key] + 0.35 * rng.standard_normal(dim)) * rng.uniform(0.5, 20.0)   # un-normalised, like pooled activations
        yield xyxy, conf, cls, feats


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_botsort_with_reid_matches_oracle(seed):
    """BoT-SORT's appearance branch on detector-derived vectors (`with_reid: true, model: auto`, default.yaml:376-379;
    csrc/tracker.cpp reid_costs) against oracle/bytetrack_ref.py's restatement of BOTSORT.get_dists / BOTrack.update_features."""
    from geotrax_amd.tracker import Tracker
    from oracle.bytetrack_ref import ByteTrackRef

    trk = Tracker("botsort", with_reid=True, proximity_thresh=0.5, appearance_thresh=0.8)
    ref = ByteTrackRef(botsort=True, with_reid=True, proximity_thresh=0.5, appearance_thresh=0.8)
    plain = ByteTrackRef(botsort=True)
    n_rows, differs = 0, False
    for t, (xyxy, conf, cls, feats) in enumerate(_stream_with_feats(seed, n_obj=60, jitter=4.0, p_miss=0.15)):
        b, i, s, c, d = trk.update(xyxy, conf, cls, None, feats=feats)
        r = ref.update(xyxy, conf, cls, None, feats=feats)
        p = plain.update(xyxy, conf, cls, None)
        np.testing.assert_array_equal(i, r[:, 4].astype(np.int32), err_msg=f"frame {t} ids")
        np.testing.assert_array_equal(d, r[:, 7].astype(np.int32), err_msg=f"frame {t} detection index")
        np.testing.assert_allclose(b, r[:, :4], rtol=0, atol=2e-3)
        differs |= len(p) != len(r) or not np.array_equal(p[:, [4, 7]], r[:, [4, 7]])
        n_rows += len(i)
    assert n_rows > 1000
    assert differs                                        # the appearance term did change associations on this stream


def test_with_reid_needs_a_vector_per_detection():
    from geotrax_amd.tracker import Tracker

    trk = Tracker("botsort", with_reid=True)
    xyxy = np.array([[10, 10, 50, 40]], np.float32)
    with pytest.raises(ValueError):
        trk.update(xyxy, np.array([0.9], np.float32), np.array([0], np.int32))
    assert Tracker("bytetrack", with_reid=True).with_reid is False      # BoT-SORT, Deep OC-SORT and TrackTrack have the branch
