"""End-to-end parity of the HIP detector against oracle/yolov8_ref.py through the C ABI:
per-layer activations, raw head output, and post-NMS boxes, fp32 and fp16, square and rect."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FRAME_HW = (432, 768)


def _frame(seed=0, hw=FRAME_HW):
    rng = np.random.default_rng(seed)
    h, w = hw
    yy, xx = np.mgrid[0:h, 0:w]
    base = 110 + 50 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
    f = np.stack([base + 20 * rng.standard_normal((h, w)) for _ in range(3)], -1)
    for _ in range(25):
        x, y = rng.integers(0, w - 40), rng.integers(0, h - 20)
        f[y:y + rng.integers(8, 20), x:x + rng.integers(15, 40)] = rng.integers(150, 255, 3)
    return np.clip(f, 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def weights():
    from geotrax_amd.weights import synthetic_yolov8

    return synthetic_yolov8(seed=1, nc=4, scale="s", cls_bias=-3.0)


LAYERS = ["model.0.conv", "model.1.conv", "model.2", "model.3.conv", "model.4", "model.6", "model.8", "model.9",
          "model.12", "model.15", "model.18", "model.21", "model.22.feat0", "model.22.feat1", "model.22.feat2"]


@pytest.mark.parametrize("half,rect,imgsz,split", [(False, False, 384, False), (True, False, 384, False), (False, True, 384, False),
                                                   (True, True, 384, False), (False, False, 384, True), (False, True, 384, True)])
def test_detector_matches_oracle(gtx_ctx, weights, half, rect, imgsz, split):
    """split=True: the fp32-grade path on the fp16 matrix pipe (split-f16x3) under the SAME assertions as the
    exact-fp32 MFMA path: per-layer 2e-4 relative-to-max, identical detections in identical order."""
    from geotrax_amd.detector import Detector
    from oracle.yolov8_ref import YoloV8Ref, detect, letterbox

    frame = _frame(0)
    kw = dict(conf=0.25, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True)
    det = Detector(weights, FRAME_HW, imgsz=imgsz, half=half, rect=rect, fp32_split=split, ctx=gtx_ctx, **kw)
    assert det.fp32_split == split
    got = det.detect(frame)

    ref_model = YoloV8Ref(weights, emulate_half=half)
    x, g = letterbox(frame, imgsz, rect, half=half)
    assert det.net_hw == (g["net_h"], g["net_w"])
    ref_raw = ref_model.forward(x)[0].numpy()

    # fp32: fmaf-chain MFMA vs torch's blocked conv, only summation order differs.
    # fp16: both sides round every stored activation to fp16; a different fp32 summation order
    # flips an fp16 rounding now and then and the flips compound over ~25 layers.
    rel = 2e-4 if not half else 3e-2
    for name in LAYERS:
        a = det.layer_output(name)
        r = ref_model.acts[name][0].permute(1, 2, 0).numpy()
        assert a.shape == r.shape, name
        err = np.abs(a - r).max() / (np.abs(r).max() + 1e-6)
        assert err < rel, f"{name}: rel-to-max error {err:.3e}"

    raw = det.raw_output()
    assert raw.shape == ref_raw.shape
    np.testing.assert_allclose(raw[:, 4:], ref_raw[:, 4:], atol=1e-4 if not half else 2e-2)
    # boxes: network pixels up to imgsz; fp32 path agrees to ~1e-5 relative
    np.testing.assert_allclose(raw[:, :4], ref_raw[:, :4], rtol=2e-5 if not half else 5e-3, atol=2e-3 if not half else 0.5)

    xyxy, conf, cls = detect(ref_model, frame, imgsz, rect, kw["conf"], kw["iou"], kw["classes"], True, kw["max_det"])
    if not half:
        # identical detections in identical (score) order, boxes within 0.01 px
        assert len(got) == len(conf)
        np.testing.assert_array_equal(got.cls, cls)
        np.testing.assert_allclose(got.conf, conf, atol=1e-5)
        np.testing.assert_allclose(got.xyxy, xyxy, atol=1e-2)
    else:
        # fp16: same detections up to threshold- / NMS-borderline ones (random weights give
        # heavily overlapping boxes, so a 1e-3 score change can flip a suppression); match by IoU
        assert abs(len(got) - len(conf)) <= max(3, len(conf) // 10)
        matched = 0
        # (seeded weights also fire inside the letterbox bars; those boxes clip to zero area and
        # cannot be matched by IoU -- they are compared by count only)
        area = (xyxy[:, 2] - xyxy[:, 0]) * (xyxy[:, 3] - xyxy[:, 1])
        xyxy, conf = xyxy[area > 1], conf[area > 1]
        for b, c in zip(xyxy, conf):
            if len(got) == 0:
                break
            ix1, iy1 = np.maximum(got.xyxy[:, 0], b[0]), np.maximum(got.xyxy[:, 1], b[1])
            ix2, iy2 = np.minimum(got.xyxy[:, 2], b[2]), np.minimum(got.xyxy[:, 3], b[3])
            inter = np.clip(ix2 - ix1, 0, None) * np.clip(iy2 - iy1, 0, None)
            a = (got.xyxy[:, 2] - got.xyxy[:, 0]) * (got.xyxy[:, 3] - got.xyxy[:, 1])
            iou = inter / (a + (b[2] - b[0]) * (b[3] - b[1]) - inter + 1e-9)
            matched += iou.max() > 0.7   # boxes are 16-64 px here; fp16 moves a side by up to ~1 px
        assert matched >= 0.9 * len(conf)
    assert len(got) > 0, "test weights/frame should produce detections"


def test_many_candidates_take_the_general_nms_path(gtx_ctx, weights):
    """conf = 0.02 lets > 4096 anchors through: the single-workgroup NMS steps aside and the
    general rank / bit-matrix / wave-pipeline kernels run; non-agnostic NMS adds the class offset.
    Both must reproduce the oracle (torchvision order) exactly on the fp32 path."""
    from geotrax_amd.detector import Detector
    from oracle.yolov8_ref import YoloV8Ref, detect

    frame = _frame(1, (640, 640))
    ref_model = YoloV8Ref(weights)
    probe = Detector(weights, (640, 640), imgsz=640, half=False, rect=False, conf=0.5, fp32_split=False, ctx=gtx_ctx)
    probe.detect(frame)
    top = np.sort(probe.raw_output()[:, 4:].max(1))[::-1]
    probe.close()
    conf_small = float(top[600])      # ~600 candidates: non-agnostic NMS on the single-workgroup path
    for agnostic, conf, split in ((True, 0.02, False), (False, 0.02, False), (False, conf_small, False), (True, 0.02, True)):
        det = Detector(weights, (640, 640), imgsz=640, half=False, rect=False, conf=conf, iou=0.6, max_det=300,
                       classes=[0, 1, 3], agnostic_nms=agnostic, fp32_split=split, ctx=gtx_ctx)
        got = det.detect(frame)
        n_cand = int((det.raw_output()[:, 4:].max(1) > conf).sum())
        assert (n_cand > 4096) == (conf < 0.1), n_cand
        assert n_cand > 300
        xyxy, cf, cls = detect(ref_model, frame, 640, False, conf, 0.6, [0, 1, 3], agnostic, 300)
        assert len(got) == len(cf) > 10
        np.testing.assert_array_equal(got.cls, cls)
        np.testing.assert_allclose(got.conf, cf, atol=1e-5)
        np.testing.assert_allclose(got.xyxy, xyxy, atol=1e-2)
        assert set(np.unique(got.cls)) <= {0, 1, 3}
        det.close()


def test_detector_batch_equals_single(gtx_ctx, weights):
    from geotrax_amd.detector import Detector

    frames = np.stack([_frame(s) for s in range(3)])
    det = Detector(weights, FRAME_HW, imgsz=384, half=True, max_batch=3, ctx=gtx_ctx, conf=0.25, max_det=300)
    singles = [det.detect(f) for f in frames]
    dptr = gtx_ctx.dev_alloc(frames.nbytes)
    try:
        gtx_ctx.dev_upload(dptr, frames)
        batch = det.detect_dev(dptr, 3)
    finally:
        gtx_ctx.dev_free(dptr)
    assert [len(s) for s in singles] == [len(b) for b in batch]
    assert len(singles[2]) == 0 and len(singles[0]) > 0  # frame 2 has no candidate: the empty path is covered
    for s, b in zip(singles, batch):
        np.testing.assert_array_equal(s.xyxy, b.xyxy)
        np.testing.assert_array_equal(s.conf, b.conf)
        np.testing.assert_array_equal(s.cls, b.cls)


@pytest.mark.parametrize("scale,gain", [("n", 1.7), ("m", 1.0), ("l", 1.0), ("x", 1.0)])
@pytest.mark.parametrize("half,split", [(False, True), (True, False), (False, False)])
def test_other_yolov8_scales_match_oracle(gtx_ctx, scale, gain, half, split):
    """"Any Ultralytics-compatible model works" (reference README): the graph is read off the tensor shapes, so the other
    YOLOv8 scales load too -- n / m / x have widths that are multiples of 16 only (16, 48, 80, 144, 400: a half-empty last
    cout tile, 16-channel K chunks, grouped head stages on a common tile). Same per-layer bars as the s model.
    (l / x with a smaller weight gain: the seeded random stack's activations grow with depth and at gain 1.7 pass 1e6,
    beyond what the split path's fp16 halves can carry -- real networks stay far below that.)"""
    from geotrax_amd.detector import Detector
    from geotrax_amd.weights import synthetic_yolov8
    from oracle.yolov8_ref import YoloV8Ref, letterbox

    w = synthetic_yolov8(seed=1, nc=4, scale=scale, cls_bias=-3.0, gain=gain)
    frame = _frame(0)
    det = Detector(w, FRAME_HW, imgsz=384, half=half, rect=False, fp32_split=split, conf=0.25, iou=0.7, max_det=300,
                   classes=[0, 1, 2, 3], agnostic_nms=True, ctx=gtx_ctx)
    det.detect(frame)
    ref = YoloV8Ref(w, emulate_half=half)
    x, _ = letterbox(frame, 384, False, half=half)
    ref_raw = ref.forward(x)[0].numpy()
    rel = 2e-4 if not half else 3e-2
    for name in LAYERS:
        a = det.layer_output(name)
        r = ref.acts[name][0].permute(1, 2, 0).numpy()
        assert a.shape == r.shape, name
        assert np.isfinite(r).all(), f"{name}: the oracle itself overflows at this gain"
        err = np.abs(a - r).max() / (np.abs(r).max() + 1e-6)
        assert err < rel, f"{scale} {name}: rel-to-max error {err:.3e}"
    raw = det.raw_output()
    assert raw.shape == ref_raw.shape
    np.testing.assert_allclose(raw[:, 4:], ref_raw[:, 4:], atol=2e-4 if not half else 3e-2)
    det.close()


def test_saturation_falls_back_to_the_exact_fp32_convolutions(gtx_ctx, monkeypatch, caplog):
    """`half: false` promises fp32's range (default.yaml:245). The seeded `l` stack at weight gain 1.7 pushes activations past
    1e6, beyond what the split-f16x3 path's fp16 halves carry (+-65504): the pass that saturates is re-run through the
    exact-fp32 MFMA kernels and the detector stays there -- with fp32_split=True it returns what the oracle returns, layer by
    layer, raw head output and detections, also for the next frame and through the asynchronous pair; one warning names what
    happened. GTX_SAT_FALLBACK=0 keeps round 3's behaviour (clamped values, flag only) so that the difference is visible."""
    import logging

    from geotrax_amd.detector import Detector
    from geotrax_amd.weights import synthetic_yolov8
    from oracle.yolov8_ref import YoloV8Ref, letterbox

    w = synthetic_yolov8(seed=1, nc=4, scale="l", cls_bias=-3.0, gain=1.7)
    frames = [_frame(0), _frame(1)]
    kw = dict(imgsz=384, half=False, rect=False, conf=0.25, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True, ctx=gtx_ctx)
    ref = YoloV8Ref(w)
    x, _ = letterbox(frames[0], 384, False)
    ref_raw = ref.forward(x)[0].numpy()
    assert max(float(ref.acts[n].abs().max()) for n in LAYERS) > 65504.0          # the case is what it claims to be
    det = Detector(w, FRAME_HW, fp32_split=True, **kw)
    assert det.fp32_split and not det.fell_back()
    with caplog.at_level(logging.WARNING, logger="geotrax_amd.detector"):
        got = det.detect(frames[0])
    assert det.saturated() and det.fell_back()
    assert caplog.text.count("exact-fp32") == 1 and "re-run" in caplog.text
    for name in LAYERS:
        a = det.layer_output(name)
        r = ref.acts[name][0].permute(1, 2, 0).numpy()
        assert np.isfinite(r).all(), name
        err = np.abs(a - r).max() / (np.abs(r).max() + 1e-6)
        assert err < 2e-4, f"{name}: rel-to-max error {err:.3e}"
    # class logits of this stack are O(1e5): compared like the layers, relative to the largest (their sigmoids are 0 or 1 except for
    # a handful of anchors whose logit is near zero, where 2e-4 of the scale decides the score -- so scores and detections are held
    # against the exact-fp32 detector, bit for bit, below)
    lg, ref_lg = det.raw_output(logits=True)[:, 4:], ref.raw_logits[0].numpy() if hasattr(ref, "raw_logits") else None
    if ref_lg is not None:
        assert np.abs(lg - ref_lg).max() / np.abs(ref_lg).max() < 2e-4
    assert (np.abs(det.raw_output()[:, 4:] - ref_raw[:, 4:]) > 2e-4).mean() < 2e-3
    exact = Detector(w, FRAME_HW, fp32_split=False, **kw)
    for f in frames:                                                               # ... and stays there: bit for bit the exact detector
        a, b = det.detect(f), exact.detect(f)
        assert len(a) == len(b) > 0
        np.testing.assert_array_equal(a.xyxy, b.xyxy)
        np.testing.assert_array_equal(a.conf, b.conf)
        np.testing.assert_array_equal(det.raw_output(), exact.raw_output())
    np.testing.assert_array_equal(got.conf, exact.detect(frames[0]).conf)
    with caplog.at_level(logging.WARNING, logger="geotrax_amd.detector"):
        det.detect(frames[1])
    assert caplog.text.count("exact-fp32") == 1                                    # logged once
    det.close()
    # the asynchronous pair: the saturating batch is re-run inside collect() from the frames still resident in HBM
    det = Detector(w, FRAME_HW, fp32_split=True, max_batch=2, **kw)
    both = np.stack(frames)
    dptr = gtx_ctx.dev_alloc(both.nbytes)
    gtx_ctx.dev_upload(dptr, both)
    det.submit_dev(dptr, 2)
    pair = det.collect()
    assert det.fell_back()
    for d, f in zip(pair, frames):
        np.testing.assert_array_equal(d.conf, exact.detect(f).conf)
    g = det.gray_dptr(1)
    assert g[0] and (g[1], g[2]) == (FRAME_HW[0] // 2, FRAME_HW[1] // 2)
    det.submit_dev(dptr, 2)                                                        # later batches go straight to the exact kernels
    np.testing.assert_array_equal(det.collect()[1].conf, pair[1].conf)
    gtx_ctx.dev_free(dptr)
    det.close()
    exact.close()


def test_saturation_flag_without_the_fallback(gtx_ctx):
    """GTX_SAT_FALLBACK=0 (read once per process, hence a child process): the flag is raised, nothing is re-run, and the clamped
    pass does NOT reproduce the exact detector -- which is why the fallback is the default."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys, numpy as np\n"
        f"sys.path[:0] = [r'{root / 'geo-trax_amd'}', r'{root}', r'{root / 'tests'}']\n"
        "from test_detector_gpu import _frame, FRAME_HW\n"
        "from geotrax_amd.detector import Detector\n"
        "from geotrax_amd.weights import synthetic_yolov8\n"
        "w = synthetic_yolov8(seed=1, nc=4, scale='l', cls_bias=-3.0, gain=1.7)\n"
        "kw = dict(imgsz=384, half=False, rect=False, conf=0.25, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True)\n"
        "a = Detector(w, FRAME_HW, fp32_split=True, **kw); b = Detector(w, FRAME_HW, fp32_split=False, **kw)\n"
        "a.detect(_frame(0)); b.detect(_frame(0))\n"
        "ra, rb = a.raw_output(logits=True)[:, 4:], b.raw_output(logits=True)[:, 4:]\n"
        "print('SAT', int(a.saturated()), int(a.fell_back()), float(np.abs(ra - rb).max() / np.abs(rb).max()))\n")
    import os

    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env={**os.environ, "GTX_SAT_FALLBACK": "0"})
    assert p.returncode == 0, p.stderr[-2000:]
    sat, fell, diff = [ln for ln in p.stdout.splitlines() if ln.startswith("SAT")][-1].split()[1:]
    assert (sat, fell) == ("1", "0") and float(diff) > 1e-3


@pytest.mark.parametrize("nc,classes", [(80, [0, 2, 5, 7, 70]), (80, None), (1, None)])
def test_class_counts_other_than_four(gtx_ctx, nc, classes):
    """COCO-style heads (80 classes, a class filter that reaches past bit 63) and single-class heads, class-wise NMS:
    same detections as the oracle (score order up to fp32 near-ties)."""
    from geotrax_amd.detector import Detector
    from geotrax_amd.weights import synthetic_yolov8
    from oracle.yolov8_ref import YoloV8Ref, detect

    w = synthetic_yolov8(seed=1, nc=nc, scale="s", cls_bias=-3.0)
    frame = _frame(0)
    det = Detector(w, FRAME_HW, imgsz=384, half=False, rect=False, fp32_split=True, conf=0.25, iou=0.7, max_det=300, classes=classes,
                   agnostic_nms=False, ctx=gtx_ctx)
    got = det.detect(frame)
    xyxy, conf, cls = detect(YoloV8Ref(w), frame, 384, False, 0.25, 0.7, classes, False, 300)
    assert len(got) == len(conf) > 20
    if classes is not None:
        assert set(got.cls.tolist()) <= set(classes) and 70 in got.cls
    np.testing.assert_allclose(got.conf, conf, atol=1e-5)
    swapped = got.cls != cls                                       # two boxes whose scores differ by < 1e-6 may trade places
    assert swapped.sum() <= 4 and np.all(np.abs(got.conf[swapped] - conf[swapped]) < 1e-5)
    np.testing.assert_allclose(got.xyxy[~swapped], xyxy[~swapped], atol=1e-2)
    det.close()
    with pytest.raises(Exception):
        Detector(synthetic_yolov8(seed=1, nc=200, scale="s"), FRAME_HW, imgsz=384, ctx=gtx_ctx)      # more than 128 classes: refused, not truncated


def test_fused_front_equals_the_two_launches(gtx_ctx, weights, monkeypatch):
    """model.1 (3x3 stride 2) and model.2.cv1 (1x1) run as one launch on the default path (ConvProblem::post_w: the 3x3 tile is
    split and staged in LDS and multiplied there by the 1x1 weights; model.1's output never reaches HBM). Same k-step order and
    MFMAs as the stand-alone 1x1 launch: every layer after it, the raw head output and the detections must equal the
    two-launch detector's bit for bit; model.1's own activations stay reachable (the stand-alone launch runs on demand)."""
    from geotrax_amd.detector import Detector

    frame = _frame(0)
    kw = dict(imgsz=384, half=False, rect=False, fp32_split=True, conf=0.25, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True, ctx=gtx_ctx)
    monkeypatch.setenv("GTX_FUSE_FRONT", "0")
    two = Detector(weights, FRAME_HW, **kw)
    monkeypatch.delenv("GTX_FUSE_FRONT")
    one = Detector(weights, FRAME_HW, **kw)
    try:
        a, b = two.detect(frame), one.detect(frame)
        assert len(two.profile(1, 1)) and sum(f["launches"] for f in two.profile(1, 1)) == sum(f["launches"] for f in one.profile(1, 1)) + 1
        np.testing.assert_array_equal(a.xyxy, b.xyxy)
        np.testing.assert_array_equal(a.conf, b.conf)
        np.testing.assert_array_equal(two.raw_output(), one.raw_output())
        for name in ("model.1.conv", "model.2", "model.4", "model.9", "model.22.feat0"):
            np.testing.assert_array_equal(two.layer_output(name), one.layer_output(name), err_msg=name)
        assert len(a) > 0
    finally:
        two.close()
        one.close()


@pytest.mark.parametrize("imgsz,rect", [(384, False), (416, True), (352, False)])
def test_fused_stem_matches_the_stem_launch(gtx_ctx, weights, monkeypatch, imgsz, rect):
    """model.0 (the stem) is computed inside model.1's launch on the default path (ConvProblem::front_img: a workgroup builds the
    17 x 33 stem patch its tile reads from the RGB0 image with 16x16x16 MFMAs; the stem's output never reaches HBM). Same
    products (three MFMAs each, the same split weights and byte / 255 table) as `stem_split_kernel`, summed in another order: the layers behind
    it agree with the three-launch detector's to 1e-5 of the layer's largest value, the detections to the bar the oracle tests
    use, and the stem's own activations stay reachable. 416 and 352 leave partial 8 x 16 tiles and odd patch origins at the borders."""
    from geotrax_amd.detector import Detector

    frame = _frame(0)
    kw = dict(imgsz=imgsz, half=False, rect=rect, fp32_split=True, conf=0.25, iou=0.7, max_det=300, classes=[0, 1, 2, 3], agnostic_nms=True, ctx=gtx_ctx)
    monkeypatch.setenv("GTX_FUSE_STEM", "0")
    sep = Detector(weights, FRAME_HW, **kw)
    monkeypatch.delenv("GTX_FUSE_STEM")
    one = Detector(weights, FRAME_HW, **kw)
    try:
        a, b = sep.detect(frame), one.detect(frame)
        fam_sep, fam_one = [f["kernel"] for f in sep.profile(1, 1)], [f["kernel"] for f in one.profile(1, 1)]
        assert "stem_split_kernel" in fam_sep and "stem_split_kernel" not in fam_one and "conv_front_split_kernel" in fam_one
        assert sum(f["launches"] for f in sep.profile(1, 1)) == sum(f["launches"] for f in one.profile(1, 1)) + 1
        for name in ("model.0.conv", "model.1.conv"):              # on demand, from the stand-alone launches
            np.testing.assert_array_equal(sep.layer_output(name), one.layer_output(name), err_msg=name)
        for name in ("model.2", "model.4", "model.9", "model.22.feat0"):
            x, y = sep.layer_output(name), one.layer_output(name)
            assert x.shape == y.shape and np.isfinite(y).all()
            assert np.abs(x - y).max() <= 1e-5 * np.abs(x).max(), name
        assert len(a) == len(b) > 0
        np.testing.assert_allclose(a.conf, b.conf, atol=1e-5)
        np.testing.assert_allclose(a.xyxy, b.xyxy, atol=1e-2)
    finally:
        sep.close()
        one.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["split", "exact", "half"])
@pytest.mark.parametrize("conf", [0.25, 0.02])
def test_object_features_are_the_detect_inputs_at_the_boxes_anchors(gtx_ctx, weights, mode, conf):
    """`with_reid: true, model: auto` (default.yaml:376-379): the vector BoT-SORT gets per box is what ultralytics'
    predictor.get_obj_feats makes -- the Detect layer's three input maps, each level's channels averaged in consecutive groups
    down to the narrowest level's width, read at the anchor the kept box came from. Checked against the same reduction in numpy
    on gtx_detector_layer_output, with the anchor found from the full decode (conf 0.02: > 4096 candidates, the general NMS path)."""
    from geotrax_amd.detector import Detector

    hw = (640, 640)
    frame = _frame(1, hw)
    det = Detector(weights, hw, imgsz=640, half=(mode == "half"), fp32_split=(mode == "split"), obj_feats=True, ctx=gtx_ctx,
                   conf=conf, iou=0.7, max_det=300, agnostic_nms=False)
    plain = Detector(weights, hw, imgsz=640, half=(mode == "half"), fp32_split=(mode == "split"), ctx=gtx_ctx,
                     conf=conf, iou=0.7, max_det=300, agnostic_nms=False)
    d, p = det.detect(frame), plain.detect(frame)
    np.testing.assert_array_equal(d.xyxy, p.xyxy)                    # keeping the vectors changes no box
    np.testing.assert_array_equal(d.conf, p.conf)
    assert p.feats is None and d.feats is not None and d.feats.shape == (len(d), 128) and len(d) > 20
    maps = [det.layer_output(n) for n in ("model.15", "model.18", "model.21")]      # [h, w, c]
    s = min(m.shape[-1] for m in maps)
    table = np.concatenate([m.reshape(-1, s, m.shape[-1] // s).astype(np.float32).mean(-1) for m in maps], 0)   # [anchors, 128]
    raw = det.raw_output(0)                                          # [anchors, 4 + nc]: xywh in network pixels, class scores
    assert len(raw) == len(table)
    for j in range(len(d)):
        k = int(d.cls[j])
        cand = np.flatnonzero(np.abs(raw[:, 4 + k] - d.conf[j]) < 1e-6)
        r = raw[cand, :4]                                            # 640 x 640 frame at imgsz 640: network pixels == frame pixels
        boxes = np.clip(np.stack([r[:, 0] - r[:, 2] / 2, r[:, 1] - r[:, 3] / 2, r[:, 0] + r[:, 2] / 2, r[:, 1] + r[:, 3] / 2], 1), 0, 640)
        hit = cand[np.abs(boxes - d.xyxy[j]).max(1) < 2e-3]          # the anchor(s) this box can have come from
        assert len(hit), j
        err = min(np.abs(table[a] - d.feats[j]).max() for a in hit)
        assert err <= 1e-6 * max(1.0, np.abs(d.feats[j]).max()), (j, err)
    det.close(); plain.close()


@pytest.mark.gpu
@pytest.mark.parametrize("imgsz,conf", [(640, 0.25), (1920, 0.25)])
def test_sparse_box_branch_is_the_dense_one_bit_for_bit(gtx_ctx, weights, monkeypatch, imgsz, conf):
    """The default fp32 path evaluates Detect's box branch (cv2[l][0], cv2[l][1]) at the anchors that pass the score gate only
    (csrc/head_sparse.hip): same boxes, scores and order as the dense layers, bit for bit -- at the image border too (zero padding
    of both layers) -- and the debug read-backs (full decode, the head's feature maps) still return the dense layers' values."""
    from geotrax_amd.detector import Detector

    from geotrax_amd.weights import calibrate_cls_bias

    hw = (2160, 3840) if imgsz == 1920 else (640, 640)
    frame = _frame(3, hw)
    kw = dict(imgsz=imgsz, conf=conf, iou=0.7, max_det=1000, agnostic_nms=True, ctx=gtx_ctx)
    monkeypatch.setenv("GTX_SPARSE_BOX", "0")
    if imgsz == 1920:                                # the seeded class bias fires nowhere on this frame: set it for ~400 candidates
        probe = Detector(weights, hw, **kw)
        probe.detect(frame)
        weights = calibrate_cls_bias(weights, probe.raw_output(logits=True)[:, 4:], conf, 400)
        probe.close()
    dense = Detector(weights, hw, **kw)
    monkeypatch.setenv("GTX_SPARSE_BOX", "1")
    sparse = Detector(weights, hw, **kw)
    assert dense.sparse_box() == (False, 0) and sparse.sparse_box() == (True, 0)
    a, b = dense.detect(frame), sparse.detect(frame)
    assert len(a) > 30
    np.testing.assert_array_equal(b.xyxy, a.xyxy)
    np.testing.assert_array_equal(b.conf, a.conf)
    np.testing.assert_array_equal(b.cls, a.cls)
    assert a.xyxy[:, :2].min() < 1.0 or imgsz == 1920                # boxes that touch the border: anchors in the first rows / columns
    if imgsz == 640:
        np.testing.assert_array_equal(sparse.raw_output(0), dense.raw_output(0))
        for name in ("model.22.feat0", "model.22.feat1", "model.22.feat2"):
            np.testing.assert_array_equal(sparse.layer_output(name), dense.layer_output(name), err_msg=name)
        c = sparse.detect(frame)                                     # and the read-backs left the next pass alone
        np.testing.assert_array_equal(c.xyxy, a.xyxy)
    assert sparse.sparse_box() == (True, 0)
    dense.close(); sparse.close()


@pytest.mark.gpu
def test_more_candidates_than_the_sparse_buffer_holds_go_through_the_dense_layers(gtx_ctx, weights, monkeypatch):
    from geotrax_amd.detector import Detector

    hw = (1080, 1920)
    frame = _frame(4, hw)
    kw = dict(imgsz=1920, conf=0.002, iou=0.7, max_det=300, agnostic_nms=False, rect=True, ctx=gtx_ctx)   # nearly every anchor passes
    monkeypatch.setenv("GTX_SPARSE_BOX", "0")
    dense = Detector(weights, hw, **kw)
    monkeypatch.setenv("GTX_SPARSE_BOX", "1")
    sparse = Detector(weights, hw, **kw)
    a, b = dense.detect(frame), sparse.detect(frame)
    assert sparse.sparse_box() == (True, 1)                          # one batch over the buffer
    np.testing.assert_array_equal(b.xyxy, a.xyxy)
    np.testing.assert_array_equal(b.conf, a.conf)
    dense.close(); sparse.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["split", "exact", "half"])
@pytest.mark.parametrize("hw,imgsz", [((1080, 1920), 960), ((720, 960), 640)])
def test_padding_rows_are_computed_once_and_every_result_stays_bit_identical(gtx_ctx, weights, monkeypatch, mode, hw, imgsz):
    """`rect: false` letterboxes a 16:9 frame into a square input: 44 % of its rows are a constant colour, and an activation row out
    of reach of the frame's rows is the same for every frame. The detector computes those rows when it is created and leaves them
    out of its launches afterwards (csrc/detector.cpp plan_pad_skip). Over a sequence of different frames, with two batch slots,
    every detection AND every layer the tests can read back must equal a detector built with GTX_PAD_SKIP=0 bit for bit."""
    from geotrax_amd.detector import Detector

    kw = dict(imgsz=imgsz, conf=0.25, iou=0.7, max_det=300, agnostic_nms=True, half=(mode == "half"), fp32_split=(mode == "split"),
              max_batch=2, ctx=gtx_ctx)
    monkeypatch.setenv("GTX_PAD_SKIP", "0")
    full = Detector(weights, hw, **kw)
    monkeypatch.setenv("GTX_PAD_SKIP", "1")
    skip = Detector(weights, hw, **kw)
    on, skipped, total = skip.pad_skip()
    assert on and 0 < skipped < 0.5 * total and full.pad_skip()[0] is False
    assert skipped > 0.1 * total or hw != (1080, 1920)                 # 16:9 into a square: a tenth of all tile rows and more
    frames = np.stack([_frame(s, hw) for s in range(4)])
    frames[1] = 255 - frames[1]                                      # unlike its neighbours everywhere
    frames[2, : hw[0] // 2] = 0
    dptr = gtx_ctx.dev_alloc(frames[:2].nbytes)
    try:
        for k in (0, 2, 1):                                          # pairs (0,1), (2,3), (1,2): every slot sees different frames
            pair = np.ascontiguousarray(frames[k:k + 2])
            gtx_ctx.dev_upload(dptr, pair)
            a, b = full.detect_dev(dptr, 2), skip.detect_dev(dptr, 2)
            for x, y in zip(a, b):
                np.testing.assert_array_equal(y.xyxy, x.xyxy)
                np.testing.assert_array_equal(y.conf, x.conf)
                np.testing.assert_array_equal(y.cls, x.cls)
            for name in LAYERS:
                for slot in (0, 1):
                    np.testing.assert_array_equal(skip.layer_output(name, slot), full.layer_output(name, slot), err_msg=f"{name} slot {slot} pair {k}")
        one = skip.detect(frames[3])                                 # a single frame after the pairs
        ref = full.detect(frames[3])
        np.testing.assert_array_equal(one.xyxy, ref.xyxy)
    finally:
        gtx_ctx.dev_free(dptr)
    full.close(); skip.close()


@pytest.mark.gpu
def test_nothing_is_skipped_without_padding_rows(gtx_ctx, weights):
    from geotrax_amd.detector import Detector

    sq = Detector(weights, (640, 640), imgsz=640, ctx=gtx_ctx)                     # square frame: no padding
    rc = Detector(weights, (1080, 1920), imgsz=960, rect=True, ctx=gtx_ctx)        # rect: 544 x 960, 2 + 2 padding rows
    assert sq.pad_skip()[1] == 0 and rc.pad_skip()[1] == 0
    sq.close(); rc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scale,gain", [("n", 1.7), ("m", 1.7), ("l", 1.0)])
def test_padding_rows_of_the_other_scales(gtx_ctx, monkeypatch, scale, gain):
    """The same bit-for-bit check on the other YOLOv8 scales (other widths, one to three bottlenecks per C2f, 16-channel chunks where
    the widths are multiples of 16 only), on an odd letterbox: 1080 x 1920 into 640 x 640 puts the frame in rows 140..499."""
    from geotrax_amd.detector import Detector
    from geotrax_amd.weights import synthetic_yolov8

    w = synthetic_yolov8(seed=3, nc=4, scale=scale, cls_bias=-3.0, gain=gain)
    hw = (1080, 1920)
    kw = dict(imgsz=640, conf=0.25, iou=0.7, max_det=300, agnostic_nms=True, ctx=gtx_ctx)
    monkeypatch.setenv("GTX_PAD_SKIP", "0")
    full = Detector(w, hw, **kw)
    monkeypatch.setenv("GTX_PAD_SKIP", "1")
    skip = Detector(w, hw, **kw)
    assert skip.pad_skip()[1] > 0
    for s in (5, 6, 7):
        f = _frame(s, hw)
        a, b = full.detect(f), skip.detect(f)
        np.testing.assert_array_equal(b.xyxy, a.xyxy)
        np.testing.assert_array_equal(b.conf, a.conf)
        for name in ("model.2", "model.4", "model.6", "model.8", "model.9", "model.22.feat0"):
            np.testing.assert_array_equal(skip.layer_output(name), full.layer_output(name), err_msg=f"{scale} {name} frame {s}")
    full.close(); skip.close()
