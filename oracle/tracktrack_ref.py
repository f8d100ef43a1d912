"""ORACLE -- test infrastructure, not product code.

numpy restatement of the `tracktrack` tracker of the reference's config (geotrax/cfg/default.yaml:445-470: "multi-cue cost
(HMIoU + ReID + confidence + angle) with iterative assignment"; reference call site geotrax/extract.py:153, the tracker
callback inside `model.track(..., persist=True)`).

The implementation the reference runs lives in ultralytics >= 8.4.80, which is neither vendored in /root/reference nor
installed here. This file is written from the config's own description of every parameter and from the published method it
names (TrackTrack, "Focusing on Tracks for Online Multi-Object Tracking", CVPR 2025: track-perspective association by
iterative mutual-minimum assignment under a shrinking threshold, track-aware initialisation; the height-modulated IoU and the
corner-angle cue are Hybrid-SORT's). Where the description leaves a choice open, the CHOICES below say what was taken.
PARITY UNPINNED against the ultralytics port.

CHOICES
  * Kalman filter: ultralytics' KalmanFilterXYWH (the one BoT-SORT uses); lost tracks are predicted with their size velocity
    zeroed, the camera-motion warp is applied like BOTSORT.multi_gmc.
  * One pool: tracked (confirmed or not) and lost tracks meet ALL detections above track_low_thresh in one cost matrix; detections
    below track_high_thresh carry penalty_p. A pair whose boxes do not overlap is infeasible.
  * cost = iou_weight * (1 - HMIoU) + reid_weight * (1 - HMIoU) [no appearance model: "HMIoU fallback"] + conf_weight * |track
    score - detection score| + angle_weight * corner-angle distance. HMIoU = IoU * (overlap of the vertical extents / their union).
  * with_reid (`model: auto`: one vector per detection from the detector): the reid_weight term is the cosine distance
    clip(1 - <track vector, detection vector>, 0, 1) between unit vectors in float32; a track keeps BOTrack's vectors (the matched
    detection's normalised vector and a 0.9-EMA of them, re-normalised), updated on every match.
  * Corner-angle distance: for each of the four corners, the angle between the track's own motion (last observation minus the
    observation delta_t = 3 frames before it, OC-SORT's rule for picking it) and the step from that earlier observation to the
    detection, divided by pi, averaged; 0 for a track with fewer than two observations or for a corner that did not move.
  * Iterative assignment: all mutually-minimal pairs (row minimum and column minimum, ties to the lower index) below the threshold
    are accepted at once, their rows and columns removed, the threshold lowered by reduce_step; repeated until no pair qualifies.
  * penalty_q ("deleted/recovered detections") has nothing to act on: the tracker callback receives the detector's NMS output only.
  * lost_match_thr > 0: tracks that were already lost before this frame and are still unmatched meet the still-unmatched
    high-score detections once more, same cost, threshold lost_match_thr.
  * Track-aware initialisation: unmatched detections with score >= new_track_thresh, in score order; one is dropped when its IoU
    with a track matched in this frame, or with a candidate accepted before it, exceeds tai_thr.
  * min_track_len: a track is reported from the frame on which it has min_track_len observations (on the clip's first frame
    at once, like ByteTrack); an unconfirmed track that misses a frame is removed.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

from oracle.bytetrack_ref import LOST, REMOVED, TRACKED, KalmanXYAH


class _Trk:
    def __init__(self, xyxy, score, cls, idx):
        self.det_xyxy = np.asarray(xyxy, dtype=np.float64)
        self.score, self.cls, self.idx = float(score), int(cls), int(idx)
        self.mean = self.cov = None
        self.state = TRACKED
        self.id = 0
        self.frame_id = self.start_frame = 0
        self.confirmed = False
        self.obs = []                      # (frame, xyxy float64) of every matched detection
        self.curr_feat = self.smooth_feat = None

    def update_features(self, f):
        self.curr_feat = f
        if self.smooth_feat is None:
            self.smooth_feat = f
        else:
            self.smooth_feat = np.float32(0.9) * self.smooth_feat + (np.float32(1) - np.float32(0.9)) * f
        self.smooth_feat = self.smooth_feat / np.sqrt(np.sum(self.smooth_feat * self.smooth_feat, dtype=np.float32))

    def xyxy(self):
        cx, cy, w, h = self.mean[:4]
        return np.array([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2])


def _z(xyxy):
    return np.array([(xyxy[0] + xyxy[2]) / 2, (xyxy[1] + xyxy[3]) / 2, xyxy[2] - xyxy[0], xyxy[3] - xyxy[1]])


def _iou_pair(a, b):
    iw = min(a[2], b[2]) - max(a[0], b[0])
    ih = min(a[3], b[3]) - max(a[1], b[1])
    if iw <= 0 or ih <= 0:
        return 0.0
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)


def _corners(b):
    return np.array([[b[0], b[1]], [b[2], b[1]], [b[0], b[3]], [b[2], b[3]]])


class TrackTrackRef:
    def __init__(self, track_high_thresh=0.6, track_low_thresh=0.25, new_track_thresh=0.7, track_buffer=30, match_thresh=0.7,
                 lost_match_thr=0.0, iou_weight=0.5, reid_weight=0.5, conf_weight=0.1, angle_weight=0.05, penalty_p=0.2,
                 penalty_q=0.4, reduce_step=0.05, tai_thr=0.55, min_track_len=3, frame_rate=30, delta_t=3, with_reid=False, **_ignored):
        self.reid = bool(with_reid)
        self.hi, self.lo, self.new_thr = track_high_thresh, track_low_thresh, new_track_thresh
        self.match_thresh, self.lost_thr = match_thresh, lost_match_thr
        self.w = (iou_weight, reid_weight, conf_weight, angle_weight)
        self.penalty_p, self.reduce_step, self.tai_thr = penalty_p, reduce_step, tai_thr
        self.min_len, self.delta_t = int(min_track_len), int(delta_t)
        self.max_time_lost = int(frame_rate / 30.0 * track_buffer)
        self.kf = KalmanXYAH(xywh=True)
        self.tracked, self.lost = [], []
        self.frame_id = 0
        self._count = 0

    # ---- cost
    def _angle(self, t, det):
        if len(t.obs) < 2:
            return 0.0
        last_f, last = t.obs[-1]
        prev = None
        for dt in range(self.delta_t, 0, -1):                 # OC-SORT k_previous_obs: the observation delta_t frames back, else the nearest
            for f, b in t.obs:
                if f == last_f - dt:
                    prev = b
                    break
            if prev is not None:
                break
        if prev is None:
            prev = t.obs[-2][1]
        v = _corners(last) - _corners(prev)
        u = _corners(det) - _corners(prev)
        tot = 0.0
        for k in range(4):
            nv, nu = np.hypot(*v[k]), np.hypot(*u[k])
            if nv < 1e-9 or nu < 1e-9:
                continue
            c = float(np.clip((v[k] @ u[k]) / (nv * nu), -1.0, 1.0))
            tot += np.arccos(c) / np.pi
        return tot / 4.0

    def _cost(self, tracks, dets, low):
        C = np.full((len(tracks), len(dets)), np.inf)
        for i, t in enumerate(tracks):
            a = t.xyxy()
            for j, d in enumerate(dets):
                b = d.det_xyxy
                iou = _iou_pair(a, b)
                if iou <= 0.0:
                    continue
                hi = (min(a[3], b[3]) - max(a[1], b[1])) / (max(a[3], b[3]) - min(a[1], b[1]))
                dist = 1.0 - iou * hi
                app = dist
                if self.reid and t.smooth_feat is not None:
                    dot = np.float32(0)
                    for k in range(len(d.curr_feat)):           # the C++ loop's float32 accumulation order
                        dot = np.float32(dot + t.smooth_feat[k] * d.curr_feat[k])
                    app = min(1.0, max(0.0, 1.0 - float(dot)))
                c = self.w[0] * dist + self.w[1] * app + self.w[2] * abs(t.score - d.score) + self.w[3] * self._angle(t, b)
                C[i, j] = c + (self.penalty_p if low[j] else 0.0)
        return C

    def _iterate(self, C, thr):
        C = C.copy()
        matches = []
        n, m = C.shape
        while n and m and thr > 0:
            rmin = np.argmin(C, 1)
            cmin = np.argmin(C, 0)
            pairs = [(i, int(rmin[i])) for i in range(n) if np.isfinite(C[i, rmin[i]]) and cmin[rmin[i]] == i and C[i, rmin[i]] < thr]
            if not pairs:
                break
            for i, j in pairs:
                matches.append((i, j))
                C[i, :] = np.inf
                C[:, j] = np.inf
            thr -= self.reduce_step
        return matches

    def _absorb(self, t, d):
        t.mean, t.cov = self.kf.update(t.mean, t.cov, _z(d.det_xyxy))
        t.state, t.frame_id = TRACKED, self.frame_id
        t.score, t.cls, t.idx = d.score, d.cls, d.idx
        t.obs.append((self.frame_id, d.det_xyxy))
        if len(t.obs) > 64:
            t.obs = t.obs[-64:]
        if len(t.obs) >= self.min_len:
            t.confirmed = True
        if d.curr_feat is not None:
            t.update_features(d.curr_feat)

    def update(self, xyxy, conf, cls, gmc=None, feats=None):
        """One frame. Returns rows [x1,y1,x2,y2,id,score,cls,idx] of the reported tracks (float32)."""
        self.frame_id += 1
        xyxy = np.asarray(xyxy, dtype=np.float32).reshape(-1, 4)
        conf = np.asarray(conf, dtype=np.float32)
        dets = [_Trk(xyxy[i], conf[i], cls[i], i) for i in range(len(conf)) if conf[i] > np.float32(self.lo)]
        if self.reid:
            for d in dets:
                f = np.asarray(feats[d.idx], dtype=np.float32)
                ss = np.float32(0)
                for v in f:
                    ss = np.float32(ss + v * v)
                d.curr_feat = d.smooth_feat = f / np.sqrt(ss)
        low = [d.score < float(np.float32(self.hi)) for d in dets]
        pool = list(self.tracked) + list(self.lost)
        was_lost = {id(t) for t in self.lost}
        for t in pool:
            m = t.mean.copy()
            if t.state != TRACKED:
                m[6] = m[7] = 0
            t.mean, t.cov = self.kf.predict(m, t.cov)
        if gmc is not None:
            Hm = np.asarray(gmc, dtype=np.float64).reshape(2, 3)
            R8 = np.kron(np.eye(4), Hm[:, :2])
            for t in pool:
                t.mean = R8 @ t.mean
                t.mean[:2] += Hm[:, 2]
                t.cov = R8 @ t.cov @ R8.T

        C = self._cost(pool, dets, low)
        matches = self._iterate(C, self.match_thresh)
        mt, md = {i for i, _ in matches}, {j for _, j in matches}
        for i, j in matches:
            self._absorb(pool[i], dets[j])
        if self.lost_thr > 0:
            rows = [i for i, t in enumerate(pool) if i not in mt and id(t) in was_lost]
            cols = [j for j in range(len(dets)) if j not in md and not low[j]]
            if rows and cols:
                for a, b in self._iterate(C[np.ix_(rows, cols)], self.lost_thr):
                    self._absorb(pool[rows[a]], dets[cols[b]])
                    mt.add(rows[a]); md.add(cols[b])
        matched_tracks = [pool[i] for i in sorted(mt)]
        for i, t in enumerate(pool):
            if i in mt:
                continue
            if id(t) in was_lost:
                continue
            t.state = LOST if t.confirmed else REMOVED
        # track-aware initialisation
        active = [t.xyxy() for t in matched_tracks]
        born = []
        for j, d in enumerate(dets):
            if j in md or low[j] or d.score < float(np.float32(self.new_thr)):
                continue
            if any(_iou_pair(d.det_xyxy, a) > self.tai_thr for a in active):
                continue
            if any(_iou_pair(d.det_xyxy, b.det_xyxy) > self.tai_thr for b in born):
                continue
            self._count += 1
            d.id = self._count
            d.mean, d.cov = self.kf.initiate(_z(d.det_xyxy))
            d.state, d.frame_id, d.start_frame = TRACKED, self.frame_id, self.frame_id
            d.obs = [(self.frame_id, d.det_xyxy)]
            d.confirmed = self.frame_id == 1 or self.min_len <= 1
            born.append(d)
        for t in self.lost:
            if t.state == LOST and self.frame_id - t.frame_id > self.max_time_lost:
                t.state = REMOVED
        self.tracked = [t for t in pool if t.state == TRACKED] + born
        self.lost = [t for t in pool if t.state == LOST]
        rows = [list(t.xyxy().astype(np.float32)) + [t.id, t.score, t.cls, t.idx]
                for t in self.tracked if t.confirmed and t.frame_id == self.frame_id]
        return np.asarray(rows, dtype=np.float32).reshape(-1, 8)
