"""The advertised drop-in: a caller that uses nothing but the call contract of the reference's two
objects -- ``model.track(frame, **cfg, persist=True)`` with its ``.detach().numpy(force=True)`` reads and ``results[0].speed``,
``Stabilizer.set_ref_frame`` / ``stabilize`` / ``transform_cur_boxes`` / ``get_cur_trans_matrix`` (geotrax/extract.py:153-187) --
over this build's ``YOLO`` and ``Stabilizer`` must give what ``python -m geotrax_amd.extract`` gives, byte for byte. The clip
contains frames without detections (which must still reach the tracker and, with BoT-SORT, the GMC: ultralytics'
on_predict_postprocess_end) and frames that cannot be registered (which take stabilo's last known transform)."""
import argparse
import logging
from pathlib import Path

import numpy as np
import pytest

from test_extract_gpu import H, W, _cfg_file, _weights_file

pytestmark = pytest.mark.gpu
logger = logging.getLogger("test_dropin")


class _InterfaceCaller:
    """A caller that knows only the two objects' public call contract (`model.track(frame, **cfg, persist=True)` ->
    `results[0].boxes` / `.speed`; `Stabilizer.set_ref_frame / stabilize / transform_cur_boxes / get_cur_trans_matrix`). Written for
    this test, frame by frame into per-frame records and one table at the end; it shares no code with the product's loops."""

    def __init__(self, model, stabilizer, ultra_kwargs):
        self.model, self.stab, self.kw = model, stabilizer, ultra_kwargs
        self.records, self.homographies, self.empty_frames, self.speeds = [], {}, 0, []

    @staticmethod
    def _host(t, dtype):
        return np.asarray(t.detach().numpy(force=True)).astype(dtype)

    def feed(self, k, frame, is_reference):
        out = self.model.track(frame, **self.kw, persist=True)[0]
        self.speeds.append(out.speed)
        n = len(out.boxes)
        xywh = self._host(out.boxes.xywh, np.float32) if n else None
        if is_reference:
            self.stab.set_ref_frame(frame, xywh)
            warped = xywh
        else:
            self.stab.stabilize(frame, xywh)
            warped = self.stab.transform_cur_boxes() if n else None
            m = self.stab.get_cur_trans_matrix()
            if m is not None:
                self.homographies[k] = np.asarray(m, np.float64).reshape(9)
        if n == 0:
            self.empty_frames += 1
            return
        ids = np.full(n, -1.0) if out.boxes.id is None else self._host(out.boxes.id, np.uint16)
        self.records.append((k, ids, xywh, np.asarray(warped, np.float32), self._host(out.boxes.cls, np.uint8), self._host(out.boxes.conf, np.float32)))

    def tables(self):
        rows = [np.column_stack([np.full(len(i), k), i, b, w, c, s]) for k, i, b, w, c, s in self.records]
        t = np.vstack(rows)
        t = t[t[:, 1] != -1].astype(np.float32)
        h = np.array([[k, *self.homographies[k]] for k in sorted(self.homographies)], np.float64)
        return t, h


def _clip_with_empty_frames(gtx_ctx, tmp_path, wpath, cfg):
    """Ten frames of the moving scene with three of them replaced by flat images on which the detector finds nothing (seeded
    weights fire on texture; the candidates are tried with the detector of the test and the first empty one is taken)."""
    from geotrax_amd.model import YOLO
    from geotrax_amd.synth import make_scene

    scene = make_scene(seed=5, h=H, w=W)
    frames = [scene.render(10 * k, 150) for k in range(10)]
    probe = YOLO(wpath, ctx=gtx_ctx)
    u = {k: v for k, v in cfg["ultralytics"].items()}
    flat = None
    for level in (0, 114, 255, 64, 192, 32):
        cand = np.full((H, W, 3), level, np.uint8)
        if len(probe.predict(cand, **u)[0].boxes) == 0:
            flat = cand
            break
    if probe.detector is not None:
        probe.detector.close()
    assert flat is not None, "no flat image is empty for the seeded detector: pick other candidates"
    for k in (3, 4, 7):                                     # two in a row (lost-track ageing over a gap) and a single one
        frames[k] = flat
    return np.stack(frames)


@pytest.mark.parametrize("tracker", ["bytetrack", "botsort", "botsort+reid"])
def test_interface_caller_over_the_dropin_objects_equals_the_engine(gtx_ctx, tmp_path, tracker):
    reid = tracker.endswith("+reid")         # `with_reid: true, model: auto`: the detector hands BoT-SORT a vector per box on both routes
    tracker = tracker.split("+")[0]
    from geotrax_amd import extract as ex
    from geotrax_amd.config_utils import load_config_all
    from geotrax_amd.model import YOLO
    from geotrax_amd.stabilizer import Stabilizer
    from geotrax_amd.synth import make_scene

    scene = make_scene(seed=5, h=H, w=W)
    wpath, _ = _weights_file(tmp_path, gtx_ctx, scene.render(0, 150))
    cfg_path, cfg = _cfg_file(tmp_path, wpath, tracker=tracker, **({"with_reid": True} if reid else {}))
    if tracker == "botsort":
        assert cfg["tracker"]["botsort"]["gmc_method"] == "sparseOptFlow"     # the reference default (default.yaml:374)
    frames = _clip_with_empty_frames(gtx_ctx, tmp_path, wpath, cfg)
    src = tmp_path / "clip.npy"
    np.save(src, frames)

    def setup():
        args = argparse.Namespace(source=str(src), cfg=cfg_path, output_folder=None, log_path=None, verbose=False, model=None,
                                  class_names=None, conf=None, classes=None, cut_frame_left=None, cut_frame_right=None, interpolate=None)
        model = ex.load_detector(args, logger)
        config = load_config_all(args, logger, model_names=model.names)
        args.cut_frame_left, args.cut_frame_right = 0, None
        return model, config

    # (1) the product: the pipelined engine
    model, config = setup()
    want_tracks, want_transforms = ex.track_with_model(model, config, logger)
    assert len(want_tracks) > 20 and want_tracks.dtype == np.float32 and want_tracks.shape[1] == 12
    rows_on = set(np.unique(want_tracks[:, 0]).astype(int))
    assert not rows_on & {3, 4, 7} and {0, 1, 2} <= rows_on                   # the empty frames write no rows
    if tracker == "bytetrack":
        assert rows_on & {5, 6, 8, 9}                                          # tracking resumes behind them (BoT-SORT's GMC sees flat -> textured jumps: its tracks may not)
    # flat frames cannot be registered: they report the last known transform (stabilo's trans_matrix_last_known)
    assert want_transforms.shape == (9, 10)
    np.testing.assert_array_equal(want_transforms[:, 0], np.arange(1, 10))
    np.testing.assert_array_equal(want_transforms[2, 1:], want_transforms[1, 1:])   # frame 3 <- frame 2
    np.testing.assert_array_equal(want_transforms[3, 1:], want_transforms[1, 1:])   # frame 4 <- frame 2
    np.testing.assert_array_equal(want_transforms[6, 1:], want_transforms[5, 1:])   # frame 7 <- frame 6

    # (2) an interface-only caller over the drop-in objects, a fresh model (fresh tracker and GMC state)
    model, config = setup()
    assert isinstance(model, YOLO)
    caller = _InterfaceCaller(model, Stabilizer(**config['stabilo']), config['ultralytics'])
    for k, frame in enumerate(frames):
        caller.feed(k, frame, is_reference=(k == 0))
    assert caller.empty_frames == 3
    assert all(set(s) == {'preprocess', 'inference', 'postprocess'} and min(s.values()) >= 0 for s in caller.speeds)
    tracks, transforms = caller.tables()
    assert tracks.dtype == want_tracks.dtype and tracks.shape == want_tracks.shape
    assert tracks.tobytes() == want_tracks.tobytes()
    assert transforms.dtype == want_transforms.dtype and transforms.shape == want_transforms.shape
    assert transforms.tobytes() == want_transforms.tobytes()

    # (3) the same loop as a product path (`engine: {pipelined: false}` / GTX_ENGINE=blocking)
    model, config = setup()
    config['main'].setdefault('engine', {})['pipelined'] = False
    assert not ex.pipelined(config)
    t3, h3 = ex.track_with_model_blocking(model, config, logger)
    assert t3.tobytes() == want_tracks.tobytes() and h3.tobytes() == want_transforms.tobytes()


def test_track_calls_the_tracker_on_frames_without_detections(gtx_ctx, tmp_path):
    """model.py used to return before tracker.update on an empty frame; ultralytics (and the engine) update on every frame."""
    from geotrax_amd.model import YOLO
    from geotrax_amd.weights import synthetic_yolov8

    model = YOLO(synthetic_yolov8(seed=1, nc=4, cls_bias=-30.0), ctx=gtx_ctx)
    frame = np.zeros((H, W, 3), np.uint8)
    calls = []
    kw = dict(imgsz=384, conf=0.25, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True, half=False, rect=True,
              tracker=dict(tracker_type="botsort", gmc_method="sparseOptFlow", track_high_thresh=0.25, track_low_thresh=0.1,
                           new_track_thresh=0.25, track_buffer=30, match_thresh=0.8, fuse_score=True))
    for k in range(3):
        res = model.track(frame, persist=True, **kw)
        if k == 0:
            orig = model._tracker.update
            model._tracker.update = lambda *a, **kws: (calls.append(kws.get("gmc")), orig(*a, **kws))[1]
        assert len(res[0].boxes) == 0 and res[0].boxes.id is None
    assert len(calls) == 2 and all(g is not None for g in calls)          # frames 1 and 2 (frame 0 ran before the wrapper): GMC warp handed in
    model.detector.close()
