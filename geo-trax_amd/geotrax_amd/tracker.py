"""Tracker object over the C ABI (gtx_tracker_*): the tracking half of ``model.track()``.

Reference behaviour replaced: the tracker callback ultralytics registers for
``model.track(..., persist=True)`` (geotrax/extract.py:153) with the active block of
cfg -> tracker (geotrax/cfg/default.yaml:361-389). Runs on the host (C++), one frame at a time. `tracker_type: ocsort` (default.yaml:391-404) selects the OC-SORT
implementation (csrc/ocsort.cpp), `deepocsort` (default.yaml:406-427) the same tracker with camera-motion compensation by the warp
handed to update() (no appearance branch there); `botsort` with `with_reid: true, model: auto` associates on the detector's own appearance
vectors as well (update(..., feats=), csrc/tracker.cpp reid_costs); `fasttrack` (default.yaml:426-443) ByteTrack with occlusion handling, written from the
config's own description of its parameters (csrc/tracker.cpp type 4, oracle/fasttrack_ref.py); `tracktrack` (default.yaml:445-470) the
multi-cue cost + iterative assignment + track-aware initialisation tracker, written the same way (csrc/tracktrack.cpp, oracle/tracktrack_ref.py).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import TrackerConfig, check, ptr

TRACKER_TYPES = {"bytetrack": 0, "botsort": 1, "ocsort": 2, "deepocsort": 3, "fasttrack": 4, "tracktrack": 5}


class Tracker:
    def __init__(self, tracker_type: str = "bytetrack", track_high_thresh: float = 0.25, track_low_thresh: float = 0.1,
                 new_track_thresh: float = 0.25, track_buffer: int = 30, match_thresh: float = 0.8,
                 fuse_score: bool = True, frame_rate: int = 30, max_tracks: int = 4096, delta_t: int = 3, inertia: float = 0.2,
                 use_byte: bool = False, min_hits: int = 3, reset_velocity_offset_occ: int = 5, reset_pos_offset_occ: int = 3,
                 enlarge_bbox_occ: float = 1.1, dampen_motion_occ: float = 0.5, active_occ_to_lost_thresh: int = 10,
                 occ_cover_thresh: float = 0.7, occ_reappear_window: int = 40, init_iou_suppress: float = 0.7,
                 with_reid: bool = False, proximity_thresh: float = 0.5, appearance_thresh: float = 0.8,
                 lost_match_thr: float = 0.0, iou_weight: float = 0.5, reid_weight: float = 0.5, conf_weight: float = 0.1, angle_weight: float = 0.05,
                 penalty_p: float = 0.2, penalty_q: float = 0.4, reduce_step: float = 0.05, tai_thr: float = 0.55, min_track_len: int = 3,
                 alpha_fixed_emb: float = 0.95, **_ignored):
        if tracker_type not in TRACKER_TYPES:
            raise NotImplementedError(f"tracker '{tracker_type}' is not implemented (available: {sorted(TRACKER_TYPES)})")
        self.lib = _lib.load()
        cfg = TrackerConfig(type=TRACKER_TYPES[tracker_type], track_high_thresh=track_high_thresh,
                            track_low_thresh=track_low_thresh, new_track_thresh=new_track_thresh,
                            track_buffer=track_buffer, match_thresh=match_thresh, fuse_score=int(fuse_score),
                            frame_rate=frame_rate, delta_t=int(delta_t), inertia=float(inertia), use_byte=int(bool(use_byte)),
                            min_hits=int(min_hits), reset_velocity_offset_occ=int(reset_velocity_offset_occ),
                            reset_pos_offset_occ=int(reset_pos_offset_occ), enlarge_bbox_occ=float(enlarge_bbox_occ),
                            dampen_motion_occ=float(dampen_motion_occ), active_occ_to_lost_thresh=int(active_occ_to_lost_thresh),
                            occ_cover_thresh=float(occ_cover_thresh), occ_reappear_window=int(occ_reappear_window),
                            init_iou_suppress=float(init_iou_suppress), with_reid=int(bool(with_reid) and tracker_type in ("botsort", "deepocsort", "tracktrack")),
                            proximity_thresh=float(proximity_thresh), appearance_thresh=float(appearance_thresh),
                            lost_match_thr=float(lost_match_thr), iou_weight=float(iou_weight), reid_weight=float(reid_weight),
                            conf_weight=float(conf_weight), angle_weight=float(angle_weight), penalty_p=float(penalty_p), penalty_q=float(penalty_q),
                            reduce_step=float(reduce_step), tai_thr=float(tai_thr), min_track_len=int(min_track_len), alpha_fixed_emb=float(alpha_fixed_emb))
        self.with_reid = bool(cfg.with_reid)
        h = C.c_void_p()
        check(self.lib.gtx_tracker_create(C.byref(cfg), C.byref(h)))
        self.handle = h
        self.cap = max_tracks
        self._xyxy = np.zeros((max_tracks, 4), np.float32)
        self._id = np.zeros(max_tracks, np.int32)
        self._score = np.zeros(max_tracks, np.float32)
        self._cls = np.zeros(max_tracks, np.int32)
        self._idx = np.zeros(max_tracks, np.int32)
        # addresses of the persistent output buffers, taken once (building ctypes pointers costs ~3 us each per call)
        self._out_ptrs = tuple(a.ctypes.data for a in (self._xyxy, self._id, self._score, self._cls, self._idx))
        self._n = C.c_int()
        self._n_ref = C.byref(self._n)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gtx_tracker_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self.lib.gtx_tracker_reset(self.handle))

    def update(self, xyxy: np.ndarray, conf: np.ndarray, cls: np.ndarray, gmc: np.ndarray | None = None, feats: np.ndarray | None = None):
        """-> (xyxy [k,4] f32, id [k] i32, score [k] f32, cls [k] i32, det_idx [k] i32). feats: [n, dim] float32 appearance vectors,
        one per detection (required when the tracker was made with with_reid)."""
        xyxy = np.ascontiguousarray(xyxy, dtype=np.float32).reshape(-1, 4)
        conf = np.ascontiguousarray(conf, dtype=np.float32)
        cls = np.ascontiguousarray(cls, dtype=np.int32)
        g = None if gmc is None else np.ascontiguousarray(gmc, dtype=np.float64).reshape(6)
        if self.with_reid:
            if feats is None or len(feats) != len(conf):
                raise ValueError("with_reid: update() needs one appearance vector per detection (feats [n, dim])")
            f = np.ascontiguousarray(feats, dtype=np.float32)
            f = f.reshape(len(conf), -1) if len(conf) else f.reshape(0, 0)
            check(self.lib.gtx_tracker_update_feats(self.handle, len(conf), xyxy.ctypes.data, conf.ctypes.data, cls.ctypes.data,
                                                    None if g is None else g.ctypes.data, f.ctypes.data if len(conf) else None,
                                                    f.shape[1] if len(conf) else 0, self.cap, self._n_ref, *self._out_ptrs))
            k = self._n.value
            return (self._xyxy[:k].copy(), self._id[:k].copy(), self._score[:k].copy(), self._cls[:k].copy(), self._idx[:k].copy())
        check(self.lib.gtx_tracker_update(self.handle, len(conf), xyxy.ctypes.data, conf.ctypes.data, cls.ctypes.data,
                                          None if g is None else g.ctypes.data, self.cap, self._n_ref, *self._out_ptrs))
        k = self._n.value
        return (self._xyxy[:k].copy(), self._id[:k].copy(), self._score[:k].copy(), self._cls[:k].copy(),
                self._idx[:k].copy())

    def replay(self, recs: np.ndarray, max_det: int, with_gmc: bool = False):
        """Feeds per-frame records (distributed.pack_frame_record layout, [n_frames, stride] float64, clip order) through
        the tracker in one call (gtx_tracker_replay): the sequential half of a frame-sharded run without a Python round
        trip per frame. -> (rows_per_frame [n_frames], xyxy [k,4], id [k], score [k], cls [k], det_idx [k]), the rows of
        all frames back to back."""
        recs = np.ascontiguousarray(recs, dtype=np.float64)
        if recs.ndim != 2:
            raise ValueError("records must be [n_frames, stride]")
        n, stride = recs.shape
        cap = max(n * max(min(max_det, 4096), 1), 1)
        per = np.zeros(max(n, 1), np.int32)
        xyxy, tid = np.zeros((cap, 4), np.float32), np.zeros(cap, np.int32)
        score, cls, idx = np.zeros(cap, np.float32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        check(self.lib.gtx_tracker_replay(self.handle, recs.ctypes.data, n, stride, int(max_det), int(bool(with_gmc)), cap, per.ctypes.data,
                                          xyxy.ctypes.data, tid.ctypes.data, score.ctypes.data, cls.ctypes.data, idx.ctypes.data))
        k = int(per[:n].sum())
        return per[:n], xyxy[:k], tid[:k], score[:k], cls[:k], idx[:k]
