"""ORACLE -- test infrastructure, not product code.

numpy restatement of `tracker.fasttrack` of the reference's config (geotrax/cfg/default.yaml:426-443): "occlusion-aware ByteTrack
variant with Kalman rollback and init-IoU suppression; ReID-free". The implementation the reference runs lives in the pinned
ultralytics (>= 8.4.80, trackers/) which is neither vendored nor installed here, and no other description of it is reachable:
this file is written FROM THE CONFIG'S OWN DESCRIPTION of the eight parameters, on top of the ByteTrack restatement
(oracle/bytetrack_ref.py). PARITY UNPINNED: where the description leaves a choice open the choice made is stated below, and a
geo-trax run of this tracker may differ in those places (tools/score_run.py scores one when an operator supplies it).

What is added to ByteTrack's update (same three associations, same Kalman filter, same list bookkeeping):
  * occlusion test (`occ_cover_thresh`): a confirmed track that found no detection in either association is OCCLUDED when another
    active track's box covers at least that fraction of its own (predicted) box;
  * an occluded track is not marked lost: it stays active on its prediction for up to `active_occ_to_lost_thresh` consecutive
    frames and is reported with the index -1 (no detection). CHOICE: it is reported (ByteTrack reports every activated tracked
    track), so a vehicle that passes behind another keeps its row;
  * at occlusion onset the filter is rolled back (`reset_velocity_offset_occ`, `reset_pos_offset_occ`): the velocity becomes the
    one the track had that many frames ago -- the last measurements before the onset belong to a box that was already being
    eaten by the occluder -- and the position the one of `reset_pos_offset_occ` frames ago carried forward by that velocity to
    the present (CHOICE: the description says "restoring Kalman position"; a position left in the past would lag the vehicle by
    that many frames). The box height is scaled once by `enlarge_bbox_occ` (aspect kept: the search region widens both ways);
  * the restored velocity is multiplied by `dampen_motion_occ` (CHOICE: once, at the onset -- "applied while occluded" read as the
    state the track keeps while occluded; multiplied in again every frame the track would stand still after three frames);
  * a track that leaves the occluded state by time-out goes lost like any other but stays re-findable for
    `occ_reappear_window` frames instead of `track_buffer`;
  * `init_iou_suppress`: an unmatched high-score detection starts no track when its IoU with an active track is at least that.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

from oracle.bytetrack_ref import LOST, REMOVED, TRACKED, ByteTrackRef, Track


class FastTrackRef(ByteTrackRef):
    def __init__(self, reset_velocity_offset_occ=5, reset_pos_offset_occ=3, enlarge_bbox_occ=1.1, dampen_motion_occ=0.5,
                 active_occ_to_lost_thresh=10, occ_cover_thresh=0.7, occ_reappear_window=40, init_iou_suppress=0.7, **kw):
        super().__init__(botsort=False, **kw)
        self.vel_off, self.pos_off = int(reset_velocity_offset_occ), int(reset_pos_offset_occ)
        self.enlarge, self.dampen = float(enlarge_bbox_occ), float(dampen_motion_occ)
        self.occ_max, self.cover, self.window = int(active_occ_to_lost_thresh), float(occ_cover_thresh), int(occ_reappear_window)
        self.suppress = float(init_iou_suppress)
        self.keep = max(self.vel_off, self.pos_off) + 1

    @staticmethod
    def _extra(t):
        if not hasattr(t, "hist"):
            t.hist, t.occluded, t.occ_frames, t.occ_lost = [], False, 0, False

    def _covered(self, t, others):
        """Largest fraction of t's box covered by one other box."""
        a = self._xyxy(t).astype(np.float64)
        area = max((a[2] - a[0]) * (a[3] - a[1]), 1e-12)
        best = 0.0
        for o in others:
            if o is t:
                continue
            b = self._xyxy(o).astype(np.float64)
            iw, ih = min(a[2], b[2]) - max(a[0], b[0]), min(a[3], b[3]) - max(a[1], b[1])
            if iw > 0 and ih > 0:
                best = max(best, iw * ih / area)
        return best

    def _absorb(self, t, det, reactivate):
        super()._absorb(t, det, reactivate)
        self._extra(t)
        t.occluded, t.occ_frames, t.occ_lost = False, 0, False

    def update(self, xyxy, conf, cls, gmc=None):
        self.frame_id += 1
        xyxy = np.asarray(xyxy, dtype=np.float32).reshape(-1, 4)
        xywh = np.stack([(xyxy[:, 0] + xyxy[:, 2]) / 2, (xyxy[:, 1] + xyxy[:, 3]) / 2,
                         xyxy[:, 2] - xyxy[:, 0], xyxy[:, 3] - xyxy[:, 1]], 1).astype(np.float32)
        conf = np.asarray(conf, dtype=np.float32)
        det_hi = [Track(xywh[i], conf[i], cls[i], i) for i in range(len(conf)) if conf[i] >= np.float32(self.hi)]
        det_lo = [Track(xywh[i], conf[i], cls[i], i) for i in range(len(conf)) if np.float32(self.lo) < conf[i] < np.float32(self.hi)]
        unconfirmed = [t for t in self.tracked if not t.activated]
        confirmed = [t for t in self.tracked if t.activated]
        ids = {t.id for t in confirmed}
        pool = confirmed + [t for t in self.lost if t.id not in ids]
        for t in pool:
            self._extra(t)
            m = t.mean.copy()
            if t.state != TRACKED:
                m[7] = 0
            t.mean, t.cov = self.kf.predict(m, t.cov)

        activated, refind, lost_now, removed_now = [], [], [], []
        matches, u_track, u_det = self._assign(self._dists(pool, det_hi, self.fuse), self.match_thresh)
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
            self._absorb(r_tracked[i], det_lo[j], False)
            activated.append(r_tracked[i])
        # tracks without a detection: occluded (stay active) or lost. The occluders are the tracks that did find one.
        seen = activated + refind
        for i in u_track2:
            t = r_tracked[i]
            if self._covered(t, seen) >= self.cover and t.occ_frames < self.occ_max:
                if not t.occluded:                                   # onset: roll the filter back
                    t.occluded = True
                    if t.hist:
                        pv = t.hist[max(len(t.hist) - self.vel_off, 0)] if self.vel_off > 0 else t.mean
                        k = min(self.pos_off, len(t.hist))
                        pp = t.hist[len(t.hist) - k] if k > 0 else t.mean
                        t.mean = t.mean.copy()
                        t.mean[4:8] = pv[4:8]
                        if k > 0:
                            t.mean[0:4] = pp[0:4] + (k + 1) * pv[4:8]    # k stored frames back + this frame's prediction step
                    t.mean[4:8] *= self.dampen
                    t.mean[3] *= self.enlarge
                t.occ_frames += 1
                t.idx = -1
                activated.append(t)                                   # stays in the tracked list
            else:
                if t.occluded:
                    t.occ_lost = True
                t.occluded = False
                t.state = LOST
                lost_now.append(t)
        left = [det_hi[j] for j in u_det]
        matches, u_unc, u_left = self._assign(self._dists(unconfirmed, left, self.fuse), 0.7)
        for i, j in matches:
            self._absorb(unconfirmed[i], left[j], False)
            activated.append(unconfirmed[i])
        for i in u_unc:
            unconfirmed[i].state = REMOVED
            removed_now.append(unconfirmed[i])
        active_now = [t for t in seen + [unconfirmed[i] for i, _ in matches]]
        for j in u_left:
            t = left[j]
            if t.score < self.new_thr:
                continue
            if self.suppress < 1.0 and active_now:
                d = self._dists([t], active_now, False)[0]
                if (np.float32(1) - d).max() >= np.float32(self.suppress):
                    continue
            self._count += 1
            t.id = self._count
            t.mean, t.cov = self.kf.initiate(self._z(t._tlwh))
            t.tracklet_len, t.state = 0, TRACKED
            t.activated = self.frame_id == 1
            t.frame_id = t.start_frame = self.frame_id
            self._extra(t)
            activated.append(t)
        for t in self.lost:
            self._extra(t)
            if self.frame_id - t.frame_id > (self.window if t.occ_lost else self.max_time_lost):
                t.state = REMOVED
                removed_now.append(t)

        def joint(a, b):
            got = {t.id for t in a}
            out = list(a)
            for t in b:
                if t.id not in got:
                    got.add(t.id)
                    out.append(t)
            return out

        def sub(a, b):
            bid = {t.id for t in b}
            return [t for t in a if t.id not in bid]

        self.tracked = joint(joint([t for t in self.tracked if t.state == TRACKED], activated), refind)
        self.lost = sub(self.lost, self.tracked) + lost_now
        self.lost = sub(self.lost, self.removed)
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
        for t in self.tracked:                                       # the filter's history, newest last
            self._extra(t)
            t.hist.append(t.mean.copy())
            if len(t.hist) > self.keep:
                del t.hist[0]
        rows = [list(self._xyxy(t)) + [t.id, t.score, t.cls, t.idx] for t in self.tracked if t.activated]
        return np.asarray(rows, dtype=np.float32).reshape(-1, 8)
