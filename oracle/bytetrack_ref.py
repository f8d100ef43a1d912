"""ORACLE -- test infrastructure, not product code.

numpy/scipy restatement of the tracker stage of the reference's hot path: the callback
ultralytics runs inside `model.track(..., persist=True)` (reference call site
geotrax/extract.py:153; parameters geotrax/cfg/default.yaml:361-389 `tracker.bytetrack` /
`tracker.botsort`).

The algorithm lives in third-party code that is not vendored in /root/reference and not
installed here: ultralytics>=8.4.80 (`trackers/byte_tracker.py`, `trackers/bot_sort.py`,
`trackers/utils/{kalman_filter,matching}.py`) and lapx>=0.5.2 (`lap.lapjv`). This file restates
the published ByteTrack / BoT-SORT procedure as those modules implement it, from memory of their
public source; the LAP is solved with scipy.optimize.linear_sum_assignment (the C++ solver is held against scipy directly on the golden
clip's frames, tests/test_tracker_golden.py) on the same extended
cost matrix lapjv(extend_cost=True, cost_limit=t) builds, which has the same optimum.

PARITY UNPINNED against the real packages (neither can be imported here; the reference's tests
mock the model). What pins it indirectly: the golden track file
data/results-pixel/U_video_cut.txt shows ids 1..N assigned in detection (confidence) order on
the first frame and Kalman-posterior boxes afterwards, which is what this procedure produces.

The appearance branch (`with_reid: true, model: auto`: BOTrack.update_features, BOTSORT.get_dists, matching.embedding_distance) is
restated the same way; the vectors themselves are oracle/yolov8_ref.py obj_feats_table (engine/predictor.py get_obj_feats).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
from scipy.optimize import linear_sum_assignment

NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3


class KalmanXYAH:
    """trackers/utils/kalman_filter.py: KalmanFilterXYAH (xywh=False) / KalmanFilterXYWH."""

    def __init__(self, xywh: bool = False):
        self.xywh = xywh
        self.F = np.eye(8)
        for i in range(4):
            self.F[i, 4 + i] = 1.0
        self.H = np.eye(4, 8)
        self.swp, self.swv = 1.0 / 20, 1.0 / 160

    def _std(self, mean, kp, kv, ap, av):
        if self.xywh:
            w, h = mean[2], mean[3]
            return np.array([kp * self.swp * w, kp * self.swp * h, kp * self.swp * w, kp * self.swp * h,
                             kv * self.swv * w, kv * self.swv * h, kv * self.swv * w, kv * self.swv * h])
        h = mean[3]
        return np.array([kp * self.swp * h, kp * self.swp * h, ap, kp * self.swp * h,
                         kv * self.swv * h, kv * self.swv * h, av, kv * self.swv * h])

    def initiate(self, z):
        mean = np.r_[np.asarray(z, dtype=np.float64), np.zeros(4)]
        return mean, np.diag(np.square(self._std(mean, 2, 10, 1e-2, 1e-5)))

    def predict(self, mean, cov):
        q = np.diag(np.square(self._std(mean, 1, 1, 1e-2, 1e-5)))
        return self.F @ mean, self.F @ cov @ self.F.T + q

    def update(self, mean, cov, z):
        r = np.diag(np.square(self._std(mean, 1, 1, 1e-1, 0)[:4]))
        pm, pc = self.H @ mean, self.H @ cov @ self.H.T + r
        chol, lower = scipy.linalg.cho_factor(pc, lower=True, check_finite=False)
        k = scipy.linalg.cho_solve((chol, lower), (cov @ self.H.T).T, check_finite=False).T
        return mean + (np.asarray(z, dtype=np.float64) - pm) @ k.T, cov - k @ pc @ k.T


class Track:
    def __init__(self, xywh, score, cls, idx, feat=None):
        x, y, w, h = (np.float32(v) for v in xywh)
        self._tlwh = np.array([x - w / np.float32(2), y - h / np.float32(2), w, h], dtype=np.float32)
        self.mean = self.cov = None
        self.activated = False
        self.state = NEW
        self.score, self.cls, self.idx = float(score), int(cls), int(idx)
        self.id = 0
        self.frame_id = self.start_frame = self.tracklet_len = 0
        # bot_sort.py BOTrack (with_reid): the detection's normalised appearance vector and the track's 0.9-EMA of them
        self.curr_feat = self.smooth_feat = None
        if feat is not None:
            self.update_features(np.array(feat, dtype=np.float32))

    def update_features(self, feat):
        feat = feat / np.linalg.norm(feat)
        self.curr_feat = feat
        if self.smooth_feat is None:
            self.smooth_feat = feat
        else:
            self.smooth_feat = np.float32(0.9) * self.smooth_feat + np.float32(1 - 0.9) * feat
        self.smooth_feat = self.smooth_feat / np.linalg.norm(self.smooth_feat)


class ByteTrackRef:
    def __init__(self, track_high_thresh=0.25, track_low_thresh=0.1, new_track_thresh=0.25, track_buffer=30,
                 match_thresh=0.8, fuse_score=True, frame_rate=30, botsort=False, with_reid=False, proximity_thresh=0.5,
                 appearance_thresh=0.8):
        self.hi, self.lo, self.new_thr = track_high_thresh, track_low_thresh, new_track_thresh
        self.match_thresh, self.fuse = match_thresh, fuse_score
        self.max_time_lost = int(frame_rate / 30.0 * track_buffer)
        self.kf = KalmanXYAH(xywh=botsort)
        self.botsort = botsort
        # bot_sort.py BOTSORT.get_dists, `with_reid: true, model: auto` (default.yaml:376-379): appearance vectors come with the detections
        self.reid, self.proximity, self.appearance = bool(with_reid) and botsort, proximity_thresh, appearance_thresh
        self.tracked, self.lost, self.removed = [], [], []
        self.frame_id = 0
        self._count = 0

    # ---- geometry helpers (float32 where the upstream numpy code is float32)
    def _tlwh(self, t):
        if t.mean is None:
            return t._tlwh.copy()
        r = t.mean[:4].copy()
        if not self.botsort:
            r[2] *= r[3]
        r[:2] -= r[2:] / 2
        return r

    def _xyxy(self, t):
        r = self._tlwh(t)
        r[2:] += r[:2]
        return r.astype(np.float32)

    def _z(self, tlwh):
        r = np.asarray(tlwh, dtype=np.float32).copy()
        r[:2] += r[2:] / np.float32(2)
        if not self.botsort:
            r[2] /= r[3]
        return r

    def _dists(self, a, b, fuse):
        if not a or not b:
            return np.zeros((len(a), len(b)), dtype=np.float32)
        A = np.stack([self._xyxy(t) for t in a]).astype(np.float32)
        B = np.stack([self._xyxy(t) for t in b]).astype(np.float32)
        iw = (np.minimum(A[:, None, 2], B[None, :, 2]) - np.maximum(A[:, None, 0], B[None, :, 0])).clip(0)
        ih = (np.minimum(A[:, None, 3], B[None, :, 3]) - np.maximum(A[:, None, 1], B[None, :, 1])).clip(0)
        inter = iw * ih
        area = (B[:, 2] - B[:, 0]) * (B[:, 3] - B[:, 1])
        area = area[None, :] + ((A[:, 2] - A[:, 0]) * (A[:, 3] - A[:, 1]))[:, None] - inter
        cost = np.float32(1) - inter / (area + np.float32(1e-7))
        if fuse:
            s = np.array([t.score for t in b], dtype=np.float32)[None, :]
            cost = np.float32(1) - (np.float32(1) - cost) * s
        return cost.astype(np.float32)

    def _get_dists(self, a, b):
        """BOTSORT.get_dists: IoU cost, score fusion, and -- with_reid -- the minimum with half the cosine distance between the
        track's smoothed vector and the detection's, where that is at most 1 - appearance_thresh and the boxes overlap by at
        least proximity_thresh."""
        d = self._dists(a, b, False)
        if not a or not b:
            return d
        mask = d > np.float32(1 - self.proximity)
        if self.fuse:
            d = self._dists(a, b, True)
        if self.reid:
            from scipy.spatial.distance import cdist
            tf = np.asarray([t.smooth_feat for t in a], dtype=np.float32)
            df = np.asarray([t.curr_feat for t in b], dtype=np.float32)
            emb = np.maximum(0.0, cdist(tf, df, "cosine")) / 2.0
            emb[emb > (1 - self.appearance)] = 1.0
            emb[mask] = 1.0
            d = np.minimum(d, emb)
        return d

    @staticmethod
    def _assign(cost, thresh):
        """lap.lapjv(cost, extend_cost=True, cost_limit=thresh) -> matches, unmatched rows/cols."""
        n, m = cost.shape
        if n == 0 or m == 0:
            return [], list(range(n)), list(range(m))
        ext = np.full((n + m, n + m), thresh / 2.0)
        ext[n:, m:] = 0
        ext[:n, :m] = cost
        r, c = linear_sum_assignment(ext)
        x = np.full(n, -1)
        for i, j in zip(r, c):
            if i < n and j < m:
                x[i] = j
        matches = [(i, int(x[i])) for i in range(n) if x[i] >= 0]
        used = {j for _, j in matches}
        return matches, [i for i in range(n) if x[i] < 0], [j for j in range(m) if j not in used]

    def _absorb(self, t, det, reactivate):
        t.mean, t.cov = self.kf.update(t.mean, t.cov, self._z(self._tlwh(det)))
        t.tracklet_len = 0 if reactivate else t.tracklet_len + 1
        t.state, t.activated, t.frame_id = TRACKED, True, self.frame_id
        t.score, t.cls, t.idx = det.score, det.cls, det.idx
        if det.curr_feat is not None:
            t.update_features(det.curr_feat)

    def update(self, xyxy, conf, cls, gmc=None, feats=None):
        """One frame. Returns rows [x1,y1,x2,y2,id,score,cls,idx] of active tracks (float32)."""
        self.frame_id += 1
        xyxy = np.asarray(xyxy, dtype=np.float32).reshape(-1, 4)
        xywh = np.stack([(xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2,
                         xyxy[:, 2] - xyxy[:, 0], xyxy[:, 3] - xyxy[:, 1]], 1).astype(np.float32)
        conf = np.asarray(conf, dtype=np.float32)
        ft = (lambda i: feats[i]) if (self.reid and feats is not None) else (lambda i: None)
        det_hi = [Track(xywh[i], conf[i], cls[i], i, ft(i)) for i in range(len(conf)) if conf[i] >= np.float32(self.hi)]
        det_lo = [Track(xywh[i], conf[i], cls[i], i, ft(i)) for i in range(len(conf))
                  if np.float32(self.lo) < conf[i] < np.float32(self.hi)]
        unconfirmed = [t for t in self.tracked if not t.activated]
        confirmed = [t for t in self.tracked if t.activated]
        ids = {t.id for t in confirmed}
        pool = confirmed + [t for t in self.lost if t.id not in ids]
        for t in pool:
            m = t.mean.copy()
            if t.state != TRACKED:
                if self.botsort:
                    m[6] = m[7] = 0
                else:
                    m[7] = 0
            t.mean, t.cov = self.kf.predict(m, t.cov)
        if self.botsort and gmc is not None:
            Hm = np.asarray(gmc, dtype=np.float64).reshape(2, 3)
            R8 = np.kron(np.eye(4), Hm[:, :2])
            for t in pool + unconfirmed:
                t.mean = R8 @ t.mean
                t.mean[:2] += Hm[:, 2]
                t.cov = R8 @ t.cov @ R8.T

        activated, refind, lost_now, removed_now = [], [], [], []
        matches, u_track, u_det = self._assign(self._get_dists(pool, det_hi) if self.reid else self._dists(pool, det_hi, self.fuse), self.match_thresh)
        for i, j in matches:
            t = pool[i]
            if t.state == TRACKED:
                self._absorb(t, det_hi[j], False)
                activated.append(t)
            else:
                self._absorb(t, det_hi[j], True)
                refind.append(t)
        r_tracked = [pool[i] for i in u_track if pool[i].state == TRACKED]
        matches, u_track2, _ = self._assign(self._dists(r_tracked, det_lo, False), 0.5)
        for i, j in matches:
            t = r_tracked[i]
            if t.state == TRACKED:
                self._absorb(t, det_lo[j], False)
                activated.append(t)
            else:
                self._absorb(t, det_lo[j], True)
                refind.append(t)
        for i in u_track2:
            t = r_tracked[i]
            if t.state != LOST:
                t.state = LOST
                lost_now.append(t)
        left = [det_hi[j] for j in u_det]
        matches, u_unc, u_left = self._assign(self._get_dists(unconfirmed, left) if self.reid else self._dists(unconfirmed, left, self.fuse), 0.7)
        for i, j in matches:
            self._absorb(unconfirmed[i], left[j], False)
            activated.append(unconfirmed[i])
        for i in u_unc:
            unconfirmed[i].state = REMOVED
            removed_now.append(unconfirmed[i])
        for j in u_left:
            t = left[j]
            if t.score < self.new_thr:
                continue
            self._count += 1
            t.id = self._count
            t.mean, t.cov = self.kf.initiate(self._z(t._tlwh))
            t.tracklet_len, t.state = 0, TRACKED
            t.activated = self.frame_id == 1
            t.frame_id = t.start_frame = self.frame_id
            activated.append(t)
        for t in self.lost:
            if self.frame_id - t.frame_id > self.max_time_lost:
                t.state = REMOVED
                removed_now.append(t)

        def joint(a, b):
            seen = {t.id for t in a}
            out = list(a)
            for t in b:
                if t.id not in seen:
                    seen.add(t.id)
                    out.append(t)
            return out

        def sub(a, b):
            bid = {t.id for t in b}
            return [t for t in a if t.id not in bid]

        self.tracked = joint(joint([t for t in self.tracked if t.state == TRACKED], activated), refind)
        self.lost = sub(self.lost, self.tracked) + lost_now
        self.lost = sub(self.lost, self.removed)
        # remove_duplicate_stracks
        pd = self._dists(self.tracked, self.lost, False)
        dupa, dupb = set(), set()
        for p, q in zip(*np.where(pd < np.float32(0.15))):
            tp = self.tracked[p].frame_id - self.tracked[p].start_frame
            tq = self.lost[q].frame_id - self.lost[q].start_frame
            (dupb.add(q) if tp > tq else dupa.add(p))
        self.tracked = [t for i, t in enumerate(self.tracked) if i not in dupa]
        self.lost = [t for i, t in enumerate(self.lost) if i not in dupb]
        self.removed.extend(removed_now)
        if len(self.removed) > 1000:
            self.removed = self.removed[-999:]
        rows = [list(self._xyxy(t)) + [t.id, t.score, t.cls, t.idx] for t in self.tracked if t.activated]
        return np.asarray(rows, dtype=np.float32).reshape(-1, 8)
