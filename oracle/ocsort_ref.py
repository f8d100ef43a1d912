"""ORACLE -- test infrastructure, not product code.

numpy restatement of OC-SORT (Cao, Pang, Weng, Khirodkar, Kitani: "Observation-Centric SORT", CVPR 2023) for the
`tracker.ocsort` block of the reference's config (geotrax/cfg/default.yaml:391-404: tracker_type ocsort with
track_high_thresh, track_low_thresh, new_track_thresh, track_buffer, match_thresh, delta_t, inertia, use_byte), the
tracker the reference selects with `tracker.active: ocsort` (geotrax/utils/config_utils.py:127-194 hands the block to
ultralytics >= 8.4.80, reference call site geotrax/extract.py:153).

The ultralytics module that implements it is not vendored in /root/reference and not installed here, so this file
restates the published algorithm as the authors' public implementation (noahcao/OC_SORT, `trackers/ocsort_tracker/
{ocsort,association,kalmanfilter}.py`) runs it, from memory of that source:
  * per-track 7-state constant-velocity Kalman filter on (u, v, s, r), R[2:,2:] *= 10, P[4:,4:] *= 1000, P *= 10,
    Q[-1,-1] *= 0.01, Q[4:,4:] *= 0.01, Joseph-form update;
  * OCM: association cost -(IoU + inertia * angle term * detection score), the angle between a track's motion
    direction (from observations delta_t frames apart) and the direction to each detection;
  * optional BYTE pass over the low-score detections;
  * OCR: a second attempt for the leftovers against the tracks' LAST OBSERVATIONS;
  * ORU: when a lost track is observed again the filter is rolled back to the frame it was lost and re-run along a
    straight virtual trajectory between the two observations (and, as in the public code, the new observation is
    then applied once more by the regular update);
  * output: last observation of tracks updated this frame with hit_streak >= min_hits (or during the first
    min_hits frames), ids from 1.
Parameter mapping (an assumption, stated in DESIGN.md): det_thresh = track_high_thresh, BYTE floor = track_low_thresh,
iou_threshold = 1 - match_thresh (the config documents match_thresh as the maximum 1 - IoU cost), max_age =
track_buffer, min_hits = 3, new tracks need score >= new_track_thresh; fuse_score has no counterpart.

PARITY UNPINNED: neither ultralytics' port nor the authors' package can be imported here and the reference holds no
OC-SORT output. All arithmetic is float64.

`OCSortRef(cmc=True)` is `tracker.active: deepocsort` (default.yaml:406-427) without its appearance branch: the same
tracker plus the camera-motion compensation of Deep OC-SORT (Maggiolino, Ahmad, Cao, Kitani, ICIP 2023; public code
GerardMaggiolino/Deep-OC-SORT, `apply_affine_correction` of the track and of the (u, v, s, r) filter), applied before the
prediction step with the 2x3 warp handed to update(): Kalman position by (m, t), velocity by m, their covariance blocks by
m . m^T, the frozen prior of a lost track likewise, the last observation and the observations of the last delta_t ages
corner by corner. Two deliberate differences from that code, both stated in DESIGN.md: every observation is moved once
(there the last observation and its entry in the age table are one array and are moved twice), and the observation ORU
starts its virtual trajectory from is moved too (there it stays in the old frame's coordinates).

`OCSortRef(cmc=True, with_reid=True)` adds Deep OC-SORT's appearance branch on vectors handed in per detection (`model: auto`:
the detector's own, default.yaml:421), as the authors' public code runs it: unit-length vectors; the first association's
cost gains -(w * similarity) with similarity = <detection vector, track vector> and w = 0.75 adaptively reduced per row and
per column by how close the two best similarities are (compute_aw_max_metric, bottom 0.5; neither has a config key: the
authors' defaults); a matched track's vector moves by a dynamic-alpha EMA, alpha = alpha_fixed_emb + (1 - alpha_fixed_emb) *
(1 - (score - det_thresh) / (1 - det_thresh)), in the first association and in OCR (not in the BYTE pass); the
unambiguous-IoU shortcut skips the term as it does there. CHOICE for the config's two gates, which the authors' code does not
have (default.yaml:422-423): a pair keeps its similarity only when IoU >= proximity_thresh (theirs: IoU > 0) and
similarity >= appearance_thresh. PARITY UNPINNED against the ultralytics port.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment


def _to_z(b):
    w, h = b[2] - b[0], b[3] - b[1]
    return np.array([b[0] + w / 2.0, b[1] + h / 2.0, w * h, w / float(h + 1e-6)])


def _to_box(x):
    w = np.sqrt(x[2] * x[3])
    h = x[2] / w
    return np.array([x[0] - w / 2.0, x[1] - h / 2.0, x[0] + w / 2.0, x[1] + h / 2.0])


def _direction(b1, b2):
    cx1, cy1 = (b1[0] + b1[2]) / 2.0, (b1[1] + b1[3]) / 2.0
    cx2, cy2 = (b2[0] + b2[2]) / 2.0, (b2[1] + b2[3]) / 2.0
    s = np.array([cy2 - cy1, cx2 - cx1])
    return s / (np.sqrt((cy2 - cy1) ** 2 + (cx2 - cx1) ** 2) + 1e-6)


def _iou(a, b):
    """a [n,4+], b [m,4+] -> [n,m]"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    xx1 = np.maximum(a[:, None, 0], b[None, :, 0])
    yy1 = np.maximum(a[:, None, 1], b[None, :, 1])
    xx2 = np.minimum(a[:, None, 2], b[None, :, 2])
    yy2 = np.minimum(a[:, None, 3], b[None, :, 3])
    wh = np.maximum(0.0, xx2 - xx1) * np.maximum(0.0, yy2 - yy1)
    return wh / ((a[:, None, 2] - a[:, None, 0]) * (a[:, None, 3] - a[:, None, 1]) +
                 (b[None, :, 2] - b[None, :, 0]) * (b[None, :, 3] - b[None, :, 1]) - wh)


def _lap(cost):
    r, c = linear_sum_assignment(cost)
    return list(zip(r.tolist(), c.tolist()))


class _KF:
    """The authors' KalmanFilterNew: filterpy's filter plus freeze / unfreeze (ORU)."""

    def __init__(self):
        self.x = np.zeros(7)
        self.F = np.eye(7)
        self.F[0, 4] = self.F[1, 5] = self.F[2, 6] = 1.0
        self.H = np.eye(4, 7)
        self.R = np.eye(4)
        self.R[2:, 2:] *= 10.0
        self.P = np.eye(7)
        self.P[4:, 4:] *= 1000.0
        self.P *= 10.0
        self.Q = np.eye(7)
        self.Q[-1, -1] *= 0.01
        self.Q[4:, 4:] *= 0.01
        self.observed = False
        self.saved = None
        self.history = []

    def predict(self):
        self.x = self.F @ self.x
        self.P = self.F @ self.P @ self.F.T + self.Q

    def update(self, z):
        self.history.append(None if z is None else np.asarray(z, dtype=np.float64).copy())
        if z is None:
            if self.observed:                                # first frame without an observation: freeze (the prior of this frame)
                self.saved = (self.x.copy(), self.P.copy(), list(self.history))
            self.observed = False
            return
        if not self.observed:
            self._unfreeze()
        self.observed = True
        y = np.asarray(z, dtype=np.float64) - self.H @ self.x
        PHT = self.P @ self.H.T
        S = self.H @ PHT + self.R
        K = PHT @ np.linalg.inv(S)
        self.x = self.x + K @ y
        IKH = np.eye(7) - K @ self.H
        self.P = IKH @ self.P @ IKH.T + K @ self.R @ K.T

    def apply_affine(self, m, t):
        def state(x, P):
            x, P = x.copy(), P.copy()
            x[:2] = m @ x[:2] + t
            x[4:6] = m @ x[4:6]
            P[:2, :2] = m @ P[:2, :2] @ m.T
            P[4:6, 4:6] = m @ P[4:6, 4:6] @ m.T
            return x, P
        self.x, self.P = state(self.x, self.P)
        moved = {}

        def centre(z):                                       # history entries are shared between the two lists: move each array once
            if z is None:
                return None
            if id(z) not in moved:
                z2 = z.copy()
                z2[:2] = m @ z[:2] + t
                moved[id(z)] = z2
            return moved[id(z)]
        last = max((i for i, z in enumerate(self.history) if z is not None), default=None)
        if self.saved is not None and not self.observed:
            sx, sP = state(self.saved[0], self.saved[1])
            sh = list(self.saved[2])
            if last is not None and last < len(sh):
                sh[last] = centre(sh[last])
            self.saved = (sx, sP, sh)
        if last is not None:
            self.history[last] = centre(self.history[last])

    def _unfreeze(self):
        if self.saved is None:
            return
        hist = self.history
        self.x, self.P, self.history = self.saved[0].copy(), self.saved[1].copy(), list(self.saved[2])[:-1]
        self.observed = True
        seen = [i for i, d in enumerate(hist) if d is not None]
        i1, i2 = seen[-2], seen[-1]
        x1, y1, s1, r1 = hist[i1]
        x2, y2, s2, r2 = hist[i2]
        w1, h1, w2, h2 = np.sqrt(s1 * r1), np.sqrt(s1 / r1), np.sqrt(s2 * r2), np.sqrt(s2 / r2)
        gap = i2 - i1
        dx, dy, dw, dh = (x2 - x1) / gap, (y2 - y1) / gap, (w2 - w1) / gap, (h2 - h1) / gap
        for i in range(gap):
            w, h = w1 + (i + 1) * dw, h1 + (i + 1) * dh
            self.update(np.array([x1 + (i + 1) * dx, y1 + (i + 1) * dy, w * h, w / float(h)]))
            if i != gap - 1:
                self.predict()


class _Trk:
    def __init__(self, box, score, cls, idx, tid, delta_t):
        self.kf = _KF()
        self.kf.x[:4] = _to_z(box)
        self.id = tid
        self.time_since_update = self.hits = self.hit_streak = self.age = 0
        self.last_observation = np.array([-1.0, -1.0, -1.0, -1.0, -1.0])
        self.observations = {}
        self.velocity = None
        self.delta_t = delta_t
        self.score, self.cls, self.idx = score, cls, idx
        self.emb = None

    def update_emb(self, emb, alpha):
        if self.emb is None:
            self.emb = emb
            return
        self.emb = alpha * self.emb + (1.0 - alpha) * emb
        self.emb = self.emb / np.sqrt(np.sum(self.emb * self.emb))

    def update(self, det):
        """det: [x1,y1,x2,y2,score,cls,idx] or None"""
        if det is None:
            self.kf.update(None)
            return
        box = np.asarray(det[:5], dtype=np.float64)
        if self.last_observation.sum() >= 0:
            prev = None
            for i in range(self.delta_t):
                if self.age - (self.delta_t - i) in self.observations:
                    prev = self.observations[self.age - (self.delta_t - i)]
                    break
            if prev is None:
                prev = self.last_observation
            self.velocity = _direction(prev, box)
        self.last_observation = box
        self.observations[self.age] = box
        self.time_since_update = 0
        self.hits += 1
        self.hit_streak += 1
        self.score, self.cls, self.idx = float(det[4]), int(det[5]), int(det[6])
        self.kf.update(_to_z(box))

    def apply_affine(self, g):
        g = np.asarray(g, dtype=np.float64).reshape(2, 3)
        m, t = g[:, :2], g[:, 2]

        def corners(b):
            b = b.copy()
            b[:2], b[2:4] = m @ b[:2] + t, m @ b[2:4] + t
            return b
        if self.last_observation.sum() >= 0:
            self.last_observation = corners(self.last_observation)
        self.observations = {a: (corners(b) if a >= self.age - self.delta_t else b) for a, b in self.observations.items()}
        self.kf.apply_affine(m, t)

    def predict(self):
        if self.kf.x[6] + self.kf.x[2] <= 0:
            self.kf.x[6] *= 0.0
        self.kf.predict()
        self.age += 1
        if self.time_since_update > 0:
            self.hit_streak = 0
        self.time_since_update += 1
        return _to_box(self.kf.x)

    def k_previous(self, k):
        if not self.observations:
            return np.array([-1.0, -1.0, -1.0, -1.0, -1.0])
        for i in range(k):
            if self.age - (k - i) in self.observations:
                return self.observations[self.age - (k - i)]
        return self.observations[max(self.observations)]


class OCSortRef:
    def __init__(self, track_high_thresh=0.25, track_low_thresh=0.1, new_track_thresh=0.25, track_buffer=30, match_thresh=0.8,
                 delta_t=3, inertia=0.2, use_byte=False, min_hits=3, cmc=False, with_reid=False, proximity_thresh=0.5, appearance_thresh=0.9,
                 alpha_fixed_emb=0.95, **_ignored):
        self.reid = bool(with_reid) and bool(cmc)
        self.prox, self.app_thr, self.alpha_fixed = float(proximity_thresh), float(appearance_thresh), float(alpha_fixed_emb)
        self.det_thresh, self.low, self.new_thr = track_high_thresh, track_low_thresh, new_track_thresh
        self.max_age, self.iou_thr = int(track_buffer), 1.0 - match_thresh
        self.delta_t, self.inertia, self.use_byte, self.min_hits = int(delta_t), float(inertia), bool(use_byte), int(min_hits)
        self.cmc = bool(cmc)
        self.trackers = []
        self.frame_count = 0
        self._count = 0

    @staticmethod
    def _aw(emb, w=0.75, bottom=0.5):
        """Deep OC-SORT compute_aw_max_metric: w reduced per row / column by the ratio of the second best to the best similarity."""
        wm = np.full_like(emb, w)
        for axis, n_other in ((1, emb.shape[1]), (0, emb.shape[0])):
            if n_other < 2:
                continue
            s = -np.sort(-emb, axis=axis)
            b0 = np.take(s, 0, axis=axis)
            b1 = np.take(s, 1, axis=axis)
            with np.errstate(divide="ignore", invalid="ignore"):
                f = np.where(b0 == 0, 0.0, 1.0 - np.maximum(b1 / np.where(b0 == 0, 1.0, b0) - bottom, 0.0) / (1.0 - bottom))
            wm = wm * (f[:, None] if axis == 1 else f[None, :])
        return wm * emb

    def _associate(self, dets, trks, velocities, prev_obs, det_emb=None):
        n, m = len(dets), len(trks)
        if m == 0:
            return [], list(range(n)), []
        if n == 0:
            return [], [], list(range(m))
        cx1, cy1 = (dets[:, 0] + dets[:, 2]) / 2.0, (dets[:, 1] + dets[:, 3]) / 2.0
        cx2, cy2 = (prev_obs[:, 0] + prev_obs[:, 2]) / 2.0, (prev_obs[:, 1] + prev_obs[:, 3]) / 2.0
        dx, dy = cx1[None, :] - cx2[:, None], cy1[None, :] - cy2[:, None]          # [trk, det]
        norm = np.sqrt(dx ** 2 + dy ** 2) + 1e-6
        dx, dy = dx / norm, dy / norm
        cosang = np.clip(velocities[:, 1][:, None] * dx + velocities[:, 0][:, None] * dy, -1, 1)
        ang = (np.pi / 2.0 - np.abs(np.arccos(cosang))) / np.pi
        valid = (prev_obs[:, 4] >= 0).astype(np.float64)[:, None]
        angle_cost = (valid * ang * self.inertia).T * dets[:, 4][:, None]          # [det, trk]
        iou = _iou(dets, trks)
        a = (iou > self.iou_thr).astype(np.int32)
        if a.sum(1).max() == 1 and a.sum(0).max() == 1:
            pairs = list(zip(*[v.tolist() for v in np.where(a)]))
        else:
            emb_cost = 0.0
            if self.reid and det_emb is not None:
                trk_emb = np.stack([t.emb for t in self.trackers])
                sim = det_emb @ trk_emb.T                                              # [det, trk]
                sim = np.where((iou <= 0) | (iou < self.prox) | (sim < self.app_thr), 0.0, sim)
                emb_cost = self._aw(sim)
            pairs = _lap(-(iou + angle_cost + emb_cost))
        md, mt = {p[0] for p in pairs}, {p[1] for p in pairs}
        u_d = [d for d in range(n) if d not in md]
        u_t = [t for t in range(m) if t not in mt]
        matches = []
        for d, t in pairs:
            if iou[d, t] < self.iou_thr:
                u_d.append(d)
                u_t.append(t)
            else:
                matches.append((d, t))
        return matches, u_d, u_t

    def update(self, xyxy, conf, cls, gmc=None, feats=None):
        """One frame. Returns rows [x1,y1,x2,y2,id,score,cls,idx] (float32), newest track first."""
        self.frame_count += 1
        xyxy = np.asarray(xyxy, dtype=np.float64).reshape(-1, 4)
        conf = np.asarray(conf, dtype=np.float64).reshape(-1)
        cls = np.asarray(cls).reshape(-1)
        rows = np.concatenate([xyxy, conf[:, None], cls[:, None].astype(np.float64), np.arange(len(conf))[:, None].astype(np.float64)], 1) \
            if len(conf) else np.zeros((0, 7))
        second = rows[(conf > self.low) & (conf < self.det_thresh)]
        dets = rows[conf > self.det_thresh]
        if self.cmc and gmc is not None:
            for trk in self.trackers:
                trk.apply_affine(gmc)
        trks = np.zeros((len(self.trackers), 5))
        keep = []
        for t, trk in enumerate(self.trackers):
            trks[t, :4] = trk.predict()
            keep.append(not np.any(np.isnan(trks[t, :4])))
        self.trackers = [trk for trk, k in zip(self.trackers, keep) if k]
        trks = trks[np.array(keep, dtype=bool)] if len(keep) else trks
        velocities = np.array([trk.velocity if trk.velocity is not None else (0.0, 0.0) for trk in self.trackers]).reshape(-1, 2)
        last_boxes = np.array([trk.last_observation for trk in self.trackers]).reshape(-1, 5)
        prev_obs = np.array([trk.k_previous(self.delta_t) for trk in self.trackers]).reshape(-1, 5)

        det_emb = alpha = None
        if self.reid:
            f = np.asarray(feats, dtype=np.float32)
            f = (f.reshape(len(conf), -1)[dets[:, 6].astype(int)] if len(conf) else np.zeros((0, 1), np.float32)).astype(np.float64)
            det_emb = f / np.sqrt(np.sum(f * f, axis=1, keepdims=True)) if len(f) else f
            trust = (dets[:, 4] - self.det_thresh) / (1.0 - self.det_thresh)
            alpha = self.alpha_fixed + (1.0 - self.alpha_fixed) * (1.0 - trust)
        matches, u_d, u_t = self._associate(dets, trks, velocities, prev_obs, det_emb)
        for d, t in matches:
            self.trackers[t].update(dets[d])
            if self.reid:
                self.trackers[t].update_emb(det_emb[d], alpha[d])
        if self.use_byte and len(second) and len(u_t):
            iou_left = _iou(second, trks[u_t])
            if iou_left.max() > self.iou_thr:
                done = []
                for a, b in _lap(-iou_left):
                    if iou_left[a, b] < self.iou_thr:
                        continue
                    self.trackers[u_t[b]].update(second[a])
                    done.append(u_t[b])
                u_t = [t for t in u_t if t not in done]
        if len(u_d) and len(u_t):
            iou_left = _iou(dets[u_d], last_boxes[u_t])
            if iou_left.max() > self.iou_thr:
                dd, dt = [], []
                for a, b in _lap(-iou_left):
                    if iou_left[a, b] < self.iou_thr:
                        continue
                    self.trackers[u_t[b]].update(dets[u_d[a]])
                    if self.reid:
                        self.trackers[u_t[b]].update_emb(det_emb[u_d[a]], alpha[u_d[a]])
                    dd.append(u_d[a])
                    dt.append(u_t[b])
                u_d = [d for d in u_d if d not in dd]
                u_t = [t for t in u_t if t not in dt]
        for t in u_t:
            self.trackers[t].update(None)
        for d in sorted(u_d):
            if dets[d, 4] < self.new_thr:
                continue
            self._count += 1
            self.trackers.append(_Trk(dets[d, :5], float(dets[d, 4]), int(dets[d, 5]), int(dets[d, 6]), self._count, self.delta_t))
            self.trackers[-1].last_observation = np.array([-1.0, -1.0, -1.0, -1.0, -1.0])
            if self.reid:
                self.trackers[-1].emb = det_emb[d]
        out = []
        i = len(self.trackers)
        for trk in reversed(self.trackers):
            d = _to_box(trk.kf.x) if trk.last_observation.sum() < 0 else trk.last_observation[:4]
            if trk.time_since_update < 1 and (trk.hit_streak >= self.min_hits or self.frame_count <= self.min_hits):
                out.append(list(d) + [trk.id, trk.score, trk.cls, trk.idx])
            i -= 1
            if trk.time_since_update > self.max_age:
                self.trackers.pop(i)
        return np.asarray(out, dtype=np.float32).reshape(-1, 8)
