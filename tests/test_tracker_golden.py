"""Golden replay of the tracker: the reference-held track file
data/results-pixel/U_video_cut.txt (committed as tests/golden/U_video_cut.txt.gz: 19 817 rows, 150 frames,
147 track ids, produced by the reference's default config = BoT-SORT, data/README.md:14-19) is the only
reference-made vector that touches K5. Its per-frame rows (raw xywh, class, confidence) are fed back as the
frame's detections, in descending-confidence order like NMS output; the tracker must reproduce the file's
identity partition: every golden id maps to exactly one of our ids and vice versa (no fragmentation, no merge),
in the same order of first appearance. Runs on CPU (the tracker is host C++)."""
import gzip
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).parent / "golden"


def _golden():
    t = np.loadtxt(gzip.open(GOLD / "U_video_cut.txt.gz"), delimiter=",")
    T = np.loadtxt(GOLD / "U_video_cut_vid_transf.txt", delimiter=",")
    return t, {int(r[0]): r[1:].reshape(3, 3) for r in T}


def _frame_dets(t, f):
    r = t[t[:, 0] == f]
    r = r[np.argsort(-r[:, 11], kind="stable")]
    xyxy = np.stack([r[:, 2] - r[:, 4] / 2, r[:, 3] - r[:, 5] / 2, r[:, 2] + r[:, 4] / 2, r[:, 3] + r[:, 5] / 2], 1).astype(np.float32)
    return r, xyxy, r[:, 11].astype(np.float32), r[:, 10].astype(np.int32)


def _camera_warp(Hs, f):
    """Stand-in for the GMC of frame f: previous-frame -> current-frame pixels from the golden homographies
    (H_f maps frame f -> reference frame): affine part of H_f^-1 H_{f-1}."""
    if f not in Hs:
        return None
    prev = Hs.get(f - 1, np.eye(3))
    M = np.linalg.inv(Hs[f]) @ prev
    return (M / M[2, 2])[:2]


def _replay(update, t, Hs, with_gmc):
    pairs, late, prev_ids = set(), [], set()
    for f in np.unique(t[:, 0]).astype(int):
        r, xyxy, conf, cls = _frame_dets(t, f)
        ids, det_idx = update(xyxy, conf, cls, _camera_warp(Hs, f) if with_gmc else None)
        assert len(set(ids)) == len(ids)
        for i, d in zip(ids, det_idx):
            pairs.add((int(i), int(r[d, 1])))
        late += [(f, int(r[m, 1]), int(r[m, 1]) in prev_ids) for m in set(range(len(r))) - set(int(d) for d in det_idx)]
        prev_ids = set(r[:, 1].astype(int))
    return pairs, late


@pytest.mark.parametrize("with_gmc", [False, True])
@pytest.mark.parametrize("impl", ["hip_host", "oracle"])
def test_botsort_reproduces_the_golden_identity_partition(impl, with_gmc):
    t, Hs = _golden()
    if impl == "hip_host":
        from geotrax_amd.tracker import Tracker

        trk = Tracker("botsort")

        def update(xyxy, conf, cls, gmc):
            out = trk.update(xyxy, conf, cls, gmc)
            return out[1], out[4]
    else:
        from oracle.bytetrack_ref import ByteTrackRef

        ref = ByteTrackRef(botsort=True)

        def update(xyxy, conf, cls, gmc):
            r = ref.update(xyxy, conf, cls, gmc)
            return r[:, 4].astype(int), r[:, 7].astype(int)

    pairs, late = _replay(update, t, Hs, with_gmc)
    ours, gold = [p[0] for p in pairs], [p[1] for p in pairs]
    n_gold = len(np.unique(t[:, 1]))
    assert n_gold == 147
    assert len(pairs) == len(set(ours)) == len(set(gold)) == n_gold      # a bijection: no fragmentation, no merged tracks
    # ids are handed out in order of first appearance in both runs
    by_ours = [g for _, g in sorted(pairs)]
    assert by_ours == sorted(by_ours)
    # frame 0: ids 1..N in detection (confidence) order, as the golden file has them
    r0 = _frame_dets(t, 0)[0]
    np.testing.assert_array_equal(r0[:, 1], np.arange(1, len(r0) + 1))
    assert {(i, i) for i in range(1, len(r0) + 1)} <= pairs
    # the only rows without a track: the first golden row of each track born after frame 0 (a new track is
    # unconfirmed for one frame, ultralytics STrack.activate: is_activated only on frame 1) and rows that re-open a
    # golden track after a gap (the box restarts unconfirmed for a frame, then the lost track takes it back)
    first = {}
    for f, i in zip(t[:, 0], t[:, 1]):
        first.setdefault(int(i), int(f))
    born_late = {i for i, f in first.items() if f > 0}
    assert {i for f, i, _ in late if first[i] == f} == born_late and len(late) <= len(born_late) + 2
    # never a row of a track that was there the frame before -- exactly so without a warp; the stand-in warp is the
    # difference of two independently estimated homographies (0.2-0.7 px of estimator noise per frame, not the real
    # GMC's output) and costs one box one frame
    assert sum(had_prev for _, _, had_prev in late) <= (1 if with_gmc else 0)


def test_bytetrack_does_not_reproduce_it():
    """Provenance check: the XYAH filter of ByteTrack fragments a few golden tracks, i.e. the golden
    file is BoT-SORT output, which is what the default config says (default.yaml:362)."""
    from geotrax_amd.tracker import Tracker

    t, Hs = _golden()
    trk = Tracker("bytetrack")
    pairs, _ = _replay(lambda *a: (lambda o: (o[1], o[4]))(trk.update(*a[:3])), t, Hs, False)
    assert len({p[1] for p in pairs}) == 147 and len(pairs) > 147


def _iou_cost(a, b):
    x1, y1 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
    x2, y2 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    ua = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]))[:, None] + ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))[None] - inter
    return (1.0 - inter / (ua + 1e-7)).astype(np.float32)


def _lap(cost, limit):
    import ctypes as C

    from geotrax_amd import _lib

    lib = _lib.load()
    n, m = cost.shape
    x, y = np.full(n, -2, np.int32), np.full(m, -2, np.int32)
    _lib.check(lib.gtx_op_linear_assignment(_lib.ptr(np.ascontiguousarray(cost, np.float32)), n, m, C.c_double(limit), _lib.ptr(x), _lib.ptr(y)))
    return x, y


def test_assignment_solver_reaches_scipys_optimum_on_the_golden_frames():
    """The C++ LAP of the trackers (csrc/tracker.cpp, exposed as gtx_op_linear_assignment; host only) against
    scipy.optimize.linear_sum_assignment -- a third-party solver -- on the 1 - IoU matrices between the golden clip's
    consecutive frames (~130 x 130): with lapjv's cost_limit semantics (matching.py:linear_assignment:
    a pair only below match_thresh, an unmatched row / column costs limit / 2) the objective must equal scipy's optimum of the
    extended matrix, and the matching itself where the optimum is unique; without a limit the plain rectangular optimum."""
    from scipy.optimize import linear_sum_assignment

    t, _ = _golden()
    frames = np.unique(t[:, 0]).astype(int)
    rng = np.random.default_rng(0)
    checked = 0
    for f in frames[:-1:7]:
        _, a, _, _ = _frame_dets(t, f)
        _, b, _, _ = _frame_dets(t, f + 1)
        b = b + rng.normal(0, 6.0, b.shape).astype(np.float32)             # jitter: otherwise every IoU is ~1 and the problem trivial
        cost = _iou_cost(a, b)
        n, m = cost.shape
        for limit in (0.8, 0.5):
            x, y = _lap(cost, limit)
            ext = np.full((n + m, n + m), limit / 2.0)
            ext[n:, m:] = 0
            ext[:n, :m] = cost
            r, c = linear_sum_assignment(ext)
            best = ext[r, c].sum()
            matched = x >= 0
            ours = cost[np.nonzero(matched)[0], x[matched]].astype(np.float64).sum() + (limit / 2.0) * ((~matched).sum() + m - matched.sum())
            assert abs(ours - best) < 1e-6, (f, limit, ours, best)
            assert (cost[np.nonzero(matched)[0], x[matched]] < limit).all()
            assert all(y[x[i]] == i for i in np.nonzero(matched)[0]) and (y >= 0).sum() == matched.sum()
            sx = np.full(n, -1)
            for i, j in zip(r, c):
                if i < n and j < m:
                    sx[i] = j
            assert (sx == x).mean() > 0.98                                   # identical up to exact ties
            checked += 1
        x, _ = _lap(cost, 0.0)
        r, c = linear_sum_assignment(cost)
        assert (x >= 0).sum() == min(n, m) and abs(cost[np.arange(n)[x >= 0], x[x >= 0]].astype(np.float64).sum() - cost[r, c].astype(np.float64).sum()) < 1e-6
    assert checked >= 40
    for n, m in ((1, 1), (3, 9), (9, 3), (40, 40)):                          # ragged, both orientations
        cost = rng.random((n, m)).astype(np.float32)
        x, _ = _lap(cost, 0.0)
        r, c = linear_sum_assignment(cost)
        assert abs(cost[np.arange(n)[x >= 0], x[x >= 0]].astype(np.float64).sum() - cost[r, c].astype(np.float64).sum()) < 1e-6
    x, y = _lap(np.zeros((0, 5), np.float32), 0.8)
    assert len(x) == 0 and (y == -1).all()
