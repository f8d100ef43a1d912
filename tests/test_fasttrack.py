"""FastTracker (`tracker.fasttrack`, geotrax/cfg/default.yaml:426-443) as built here: C++ (csrc/tracker.cpp, gtx_tracker type 4)
against oracle/fasttrack_ref.py on seeded streams, its reduction to ByteTrack when the occlusion handling is switched off, and the
behaviours the config's description names: an occluded track stays active on its prediction and keeps its identity, a time-out,
the re-find window, init-IoU suppression. Runs on CPU (the tracker never touches the GPU). Parity against the pinned ultralytics'
implementation is UNPINNED (oracle header)."""
import numpy as np
import pytest

from test_tracker import _stream


def _dense_stream(seed, **kw):
    """Objects packed on a small patch: boxes cover each other often, detections of covered ones drop out."""
    rng = np.random.default_rng(seed + 500)
    for xyxy, conf, cls in _stream(seed, n_obj=60, n_frames=70, w=900, h=600, jitter=1.5, **kw):
        keep = np.ones(len(conf), bool)
        for i in range(len(conf)):                      # a box mostly under a more confident one is not detected half of the time
            for j in range(i):
                iw = min(xyxy[i, 2], xyxy[j, 2]) - max(xyxy[i, 0], xyxy[j, 0])
                ih = min(xyxy[i, 3], xyxy[j, 3]) - max(xyxy[i, 1], xyxy[j, 1])
                if iw > 0 and ih > 0 and iw * ih > 0.6 * (xyxy[i, 2] - xyxy[i, 0]) * (xyxy[i, 3] - xyxy[i, 1]) and rng.random() < 0.5:
                    keep[i] = False
        yield xyxy[keep], conf[keep], cls[keep]


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("kw", [dict(), dict(occ_cover_thresh=0.4, active_occ_to_lost_thresh=4, occ_reappear_window=8, init_iou_suppress=0.5,
                                            reset_velocity_offset_occ=2, reset_pos_offset_occ=1, enlarge_bbox_occ=1.3, dampen_motion_occ=0.8)])
def test_fasttrack_matches_oracle(seed, kw):
    from geotrax_amd.tracker import Tracker
    from oracle.fasttrack_ref import FastTrackRef

    trk, ref = Tracker("fasttrack", **kw), FastTrackRef(**kw)
    n_rows = n_occluded = 0
    for t, (xyxy, conf, cls) in enumerate(_dense_stream(seed)):
        b, i, s, c, d = trk.update(xyxy, conf, cls)
        r = ref.update(xyxy, conf, cls)
        assert len(i) == len(r), f"frame {t}: {len(i)} tracks vs {len(r)}"
        np.testing.assert_array_equal(i, r[:, 4].astype(np.int32), err_msg=f"frame {t} ids")
        np.testing.assert_array_equal(d, r[:, 7].astype(np.int32), err_msg=f"frame {t} detection index")
        np.testing.assert_allclose(b, r[:, :4], rtol=0, atol=2e-3)
        n_rows += len(i)
        n_occluded += int((d < 0).sum())
    assert n_rows > 1500 and n_occluded > 20                # the occluded state is exercised, not only defined


@pytest.mark.parametrize("seed", [0, 3])
def test_fasttrack_without_its_occlusion_handling_is_bytetrack(seed):
    """occ_cover_thresh above 1 (nothing is ever occluded) and init_iou_suppress = 1 (disabled, default.yaml:443): ByteTrack."""
    from geotrax_amd.tracker import Tracker

    ft, bt = Tracker("fasttrack", occ_cover_thresh=1.5, init_iou_suppress=1.0), Tracker("bytetrack")
    for xyxy, conf, cls in _dense_stream(seed):
        a, b = ft.update(xyxy, conf, cls), bt.update(xyxy, conf, cls)
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)


def _pass_behind(t, speed=10.0, occluder=(400, 300, 470, 380)):
    """A small vehicle drives behind a parked truck: undetected while the truck covers 60 % of it."""
    x = 250 + speed * t
    b = [x, 325, x + 50, 350]
    cov = max(0.0, min(b[2], occluder[2]) - max(b[0], occluder[0])) / 50.0
    dets = [list(occluder)] + ([] if cov >= 0.6 else [b])
    return np.asarray(dets, np.float32), np.full(len(dets), 0.9, np.float32), np.zeros(len(dets), np.int32), cov >= 0.6, x


def test_an_occluded_track_stays_active_and_keeps_its_identity():
    from geotrax_amd.tracker import Tracker

    ft, bt = Tracker("fasttrack", occ_cover_thresh=0.5), Tracker("bytetrack")
    hidden_frames, ft_rows, ids_after = 0, 0, {"ft": set(), "bt": set()}
    for t in range(40):
        xyxy, conf, cls, hidden, x = _pass_behind(t)
        bx, i, s, c, d = ft.update(xyxy, conf, cls)
        _, i2, *_ = bt.update(xyxy, conf, cls)
        if hidden:
            hidden_frames += 1
            assert 2 in i and d[list(i).index(2)] == -1            # reported on its prediction, no detection behind it
            assert 2 not in i2                                      # ByteTrack reports nothing for it
            ft_rows += 1
            k = list(i).index(2)
            assert abs(bx[k, 0] - x) < 40                           # the rolled-back, dampened prediction stays near the vehicle
        elif t > 25:
            ids_after["ft"].update(int(v) for v in i)
            ids_after["bt"].update(int(v) for v in i2)
    assert hidden_frames >= 4 and ft_rows == hidden_frames
    assert ids_after["ft"] == {1, 2}                                # same identity on the other side


def test_occlusion_times_out_and_the_refind_window_is_its_own():
    from geotrax_amd.tracker import Tracker

    ft = Tracker("fasttrack", occ_cover_thresh=0.5, active_occ_to_lost_thresh=3, occ_reappear_window=6, track_buffer=30)
    truck = np.asarray([[400, 300, 700, 380]], np.float32)
    car = np.asarray([[300, 325, 350, 350]], np.float32)
    one = lambda b: (b, np.full(len(b), 0.9, np.float32), np.zeros(len(b), np.int32))
    for t in range(8):                                              # the car drives up to the truck and vanishes behind it
        c = car + np.float32([12 * t, 0, 12 * t, 0])
        ft.update(*one(np.concatenate([truck, c])))
    reported = []
    for t in range(12):
        _, i, _, _, d = ft.update(*one(truck))
        reported.append(2 in i)
    assert reported[:3] == [True, True, True] and not any(reported[3:])      # active for active_occ_to_lost_thresh frames, then lost
    # 8 frames after it was last seen the window (6) has closed: the car comes back as a new track although track_buffer is 30
    c = car + np.float32([420, 0, 420, 0])
    for _ in range(3):
        _, i, *_ = ft.update(*one(np.concatenate([truck, c])))
    assert 2 not in i and 3 in i


def test_init_iou_suppression():
    from geotrax_amd.tracker import Tracker

    one = lambda b, s: (np.asarray(b, np.float32), np.asarray(s, np.float32), np.zeros(len(b), np.int32))
    for kind, expect in (("fasttrack", {1}), ("bytetrack", {1, 2})):
        trk = Tracker(kind)
        trk.update(*one([[100, 100, 200, 160]], [0.9]))
        for _ in range(3):                                          # a second, weaker box on top of the tracked one (IoU 0.75 with it)
            _, i, *_ = trk.update(*one([[100, 100, 200, 160], [110, 104, 205, 163]], [0.9, 0.6]))
        assert set(int(v) for v in i) == expect, kind
