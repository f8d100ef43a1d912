"""Parity of the HIP RT-DETR detector (csrc/rtdetr.cpp, rtdetr_kernels.hip; geotrax/extract.py:222-225 swaps YOLO for RTDETR)
against oracle/rtdetr_ref.py through the C ABI: probed layers of the backbone / encoder / decoder, the selected queries, every
query's box and scores, and the boxes after the score stage. Bars: YOLOv8's (per layer <= 2e-4 of the layer maximum on both
fp32-grade paths, the same queries, boxes / scores <= 1e-4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FRAME_HW = (432, 768)
MAP_LAYERS = ["model.0.stem1.conv", "model.0.stem2a.conv", "model.0.stem2b.conv", "model.0.stem3.conv", "model.0", "model.1", "model.2.conv", "model.3", "model.5",
              "model.6", "model.7", "model.9", "model.10.conv", "model.11", "model.12.conv", "model.16", "model.17.conv", "model.21", "model.24", "model.27"]


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
    from geotrax_amd.weights import synthetic_rtdetr

    return synthetic_rtdetr(seed=3, nc=4)


def _rel(a, r):
    return float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-12))


def _check_against_oracle(det, ref, frame, imgsz, conf, classes, bar=2e-4, layers=MAP_LAYERS):
    from oracle.rtdetr_ref import postprocess, stretch

    got = det.detect(frame)
    pred = ref.forward(stretch(frame, imgsz))[0].numpy()
    worst = {}
    for name in layers:
        a = det.layer_output(name)
        r = ref.acts[name][0].permute(1, 2, 0).numpy()
        assert a.shape == r.shape, (name, a.shape, r.shape)
        worst[name] = _rel(a, r)
        assert worst[name] <= bar, (name, worst[name])
    # decoder inputs: the three projected levels with the invalid anchors' rows zeroed, their enc_output rows, the class logits
    shapes = [(imgsz // s, imgsz // s) for s in (8, 16, 32)]
    _, valid = ref._anchors(shapes)
    feats = (ref.acts["model.28.feats"] * valid)[0].numpy()
    enc, scores = ref.acts["model.28.enc_output"][0].numpy(), ref.acts["model.28.enc_scores"][0].numpy()
    o = 0
    for l, (h, w) in enumerate(shapes):
        for name, r in (("feats", feats), ("enc_output", enc), ("enc_scores", scores)):
            a = det.layer_output(f"model.28.{name}.{l}").reshape(h * w, -1)[:, :r.shape[1]]
            e = _rel(a, r[o:o + h * w])
            worst[f"model.28.{name}.{l}"] = e
            assert e <= bar, (name, l, e)
        o += h * w
    # the same queries (top-k on the max class logit), in the same order up to swaps of near-ties: two anchors whose scores differ by
    # less than the summation noise may trade places, and the anchors outside (0.01, 0.99) all carry the same row (enc_output of a
    # zeroed row) and the same box (inf -> 1), a tie torch.topk and the kernel may break differently. The decoder is equivariant
    # under a permutation of its queries, so every query is compared with the oracle's query of the same anchor.
    idx = det.layer_output_int("model.28.topk").ravel()
    want_idx = ref.topk[0].numpy()
    inval = ~valid[0, :, 0].numpy()
    key = ref.acts["model.28.enc_scores"][0].max(-1).values.numpy()
    assert sorted(idx[~inval[idx]]) == sorted(want_idx[~inval[want_idx]]) and inval[idx].sum() == inval[want_idx].sum()
    moved = idx != want_idx
    assert np.abs(key[idx[moved]] - key[want_idx[moved]]).max(initial=0) <= 1e-5 * np.abs(key).max(), "queries out of order beyond a near-tie"
    pos = {int(a): j for j, a in enumerate(want_idx) if not inval[a]}
    spare = [j for j, a in enumerate(want_idx) if inval[a]]
    to_ref = np.array([pos[int(a)] if not inval[a] else spare.pop() for a in idx])
    for i in range(ref.ndl):
        a = det.layer_output(f"model.28.decoder.layers.{i}")[0]
        r = ref.acts[f"model.28.decoder.layers.{i}"][0].numpy()[to_ref]
        e = _rel(a, r)
        worst[f"decoder.{i}"] = e
        assert e <= bar, (i, e)
    raw = det.raw_output()
    assert raw.shape == pred.shape
    np.testing.assert_allclose(raw[:, :4], pred[to_ref, :4], atol=1e-4)        # normalised xywh
    np.testing.assert_allclose(raw[:, 4:], pred[to_ref, 4:], atol=1e-4)        # class scores
    xyxy, score, cls, _ = postprocess(pred, frame.shape[:2], conf, classes, det.max_det)
    assert len(got) == len(score) > 0
    np.testing.assert_allclose(got.conf, score, atol=1e-4)                      # descending on both sides
    tol = 0.05 * max(frame.shape[:2]) / 640
    for k in range(len(score)):                                                 # rows of equal score (within the bar) may be swapped
        near = np.flatnonzero(np.abs(score - got.conf[k]) <= 1e-4)
        d = np.abs(xyxy[near] - got.xyxy[k]).max(1)
        assert d.min() <= tol and cls[near[d.argmin()]] == got.cls[k], (k, d.min())
    return got, pred, worst


@pytest.mark.parametrize("split", [False, True])
def test_rtdetr_matches_oracle(gtx_ctx, weights, split):
    from geotrax_amd.detector import Detector
    from oracle.rtdetr_ref import RtDetrRef

    frame = _frame(0)
    det = Detector(weights, FRAME_HW, imgsz=640, conf=0.3, max_det=300, classes=[0, 1, 3], fp32_split=split, ctx=gtx_ctx)
    assert det.rtdetr and det.net_hw == (640, 640) and det.fp32_split == split
    ref = RtDetrRef(weights)
    got, pred, worst = _check_against_oracle(det, ref, frame, 640, 0.3, [0, 1, 3])
    print({k: f"{v:.1e}" for k, v in worst.items()})
    assert 0 < len(got) < 300                      # the threshold and the class filter both cut
    assert not det.saturated()
    det.close()


@pytest.mark.parametrize("split", [True, False])
def test_rtdetr_4k_matches_the_oracle(gtx_ctx, weights, split):
    """The reference configuration: one 3840 x 2160 frame stretched to 1920 x 1920 (75 600 anchors, 3 600 AIFI tokens) on the default
    (split-f16x3) and the exact-fp32 path: the probed layers, the 300 selected queries, their boxes and scores."""
    from geotrax_amd.detector import Detector
    from oracle.rtdetr_ref import RtDetrRef

    hw = (2160, 3840)
    frame = _frame(0, hw)
    det = Detector(weights, hw, imgsz=1920, conf=0.25, max_det=300, fp32_split=split, ctx=gtx_ctx)
    assert det.net_hw == (1920, 1920)
    ref = RtDetrRef(weights)
    got, pred, worst = _check_against_oracle(det, ref, frame, 1920, 0.25, None, layers=["model.0", "model.1", "model.3", "model.7", "model.9", "model.11", "model.16", "model.21", "model.24", "model.27"])
    print({k: f"{v:.1e}" for k, v in worst.items()})
    assert len(got) > 0 and not det.saturated()
    det.close()


def test_rtdetr_batch_of_two_and_other_sizes(gtx_ctx, weights):
    """Two different frames in one pass == the two frames alone, bit for bit; a non-square-friendly size (imgsz 480: 15 x 15 tokens)."""
    from geotrax_amd.detector import Detector
    from oracle.rtdetr_ref import RtDetrRef

    f0, f1 = _frame(1), _frame(2)
    det = Detector(weights, FRAME_HW, imgsz=480, conf=0.25, max_det=100, max_batch=2, ctx=gtx_ctx)
    a0, a1 = det.detect(f0), det.detect(f1)
    both = np.ascontiguousarray(np.stack([f0, f1]))
    dptr = gtx_ctx.dev_alloc(both.nbytes)
    try:
        gtx_ctx.dev_upload(dptr, both)
        b0, b1 = det.detect_dev(dptr, 2)
    finally:
        gtx_ctx.dev_free(dptr)
    for a, b in ((a0, b0), (a1, b1)):
        assert len(a) == len(b) > 0
        assert a.xyxy.tobytes() == b.xyxy.tobytes() and a.conf.tobytes() == b.conf.tobytes() and (a.cls == b.cls).all()
    ref = RtDetrRef(weights)
    _check_against_oracle(det, ref, f0, 480, 0.25, None, layers=["model.0", "model.9", "model.11", "model.27"])
    det.close()


def test_rtdetr_through_the_model_object(gtx_ctx, weights, tmp_path):
    """The reference's dispatch (extract.py:222-225): YOLO(path) on a file whose graph is RT-DETR reports an rtdetr yaml; RTDETR(path)
    loads it; track() runs detector + tracker; a YOLOv8 file is refused by RTDETR()."""
    from geotrax_amd.model import RTDETR, YOLO
    from geotrax_amd.weights import save_weights, synthetic_yolov8

    p = tmp_path / "rtdetr-l.safetensors"
    save_weights(weights, p)
    m = YOLO(str(p), ctx=gtx_ctx)
    assert "rtdetr" in m.model.yaml_file
    m = RTDETR(str(p), ctx=gtx_ctx)
    frame = _frame(0)
    kw = dict(imgsz=640, conf=0.3, iou=0.7, max_det=300, classes=None, agnostic_nms=True, half=False, rect=False,
              tracker=dict(tracker_type="bytetrack", track_high_thresh=0.3, track_low_thresh=0.1, new_track_thresh=0.3, track_buffer=30, match_thresh=0.8, fuse_score=True))
    r0 = m.track(frame, persist=True, **kw)[0]
    r1 = m.track(frame, persist=True, **kw)[0]
    assert len(r0.boxes) > 0 and r1.boxes.id is not None and len(r1.boxes) > 0
    assert set(r0.speed) == {"preprocess", "inference", "postprocess"}
    m.detector.close()
    q = tmp_path / "yolov8s.safetensors"
    save_weights(synthetic_yolov8(seed=0, nc=4), q)
    with pytest.raises(ValueError):
        RTDETR(str(q), ctx=gtx_ctx)


def test_rtdetr_saturating_checkpoint_falls_back_to_the_exact_kernels(gtx_ctx, weights, caplog):
    """A checkpoint whose stem output leaves fp16's range (weights x 3e4: activations ~1e5) cannot be carried in the pair format: the
    pass that clamps is re-run on the exact-fp32 kernels inside collect() and the detector stays there (Detector's rule, rtdetr.cpp
    fall_back_to_exact) -- the same boxes, scores and queries as a detector built with fp32_split=False, bit for bit."""
    import logging

    from geotrax_amd.detector import Detector

    w = dict(weights)
    w["model.0.stem1.conv.weight"] = (weights["model.0.stem1.conv.weight"] * np.float32(3e4)).astype(np.float32)
    frame = _frame(0)
    kw = dict(imgsz=320, conf=0.25, max_det=300, ctx=gtx_ctx)
    exact = Detector(w, FRAME_HW, fp32_split=False, **kw)
    want = exact.detect(frame)
    assert np.abs(exact.layer_output("model.0.stem1.conv")).max() > 65504.0          # the case is what it claims to be
    det = Detector(w, FRAME_HW, fp32_split=True, **kw)
    assert not det.fell_back()
    with caplog.at_level(logging.WARNING, logger="geotrax_amd.detector"):
        got = det.detect(frame)
    assert det.saturated() and det.fell_back() and "exact-fp32" in caplog.text
    assert len(got) == len(want)
    np.testing.assert_array_equal(got.xyxy, want.xyxy)
    np.testing.assert_array_equal(got.conf, want.conf)
    np.testing.assert_array_equal(det.raw_output(), exact.raw_output())
    again = det.detect(_frame(1))                                                    # ... and stays there
    np.testing.assert_array_equal(again.conf, exact.detect(_frame(1)).conf)
    det.close()
    exact.close()


def test_rtdetr_other_head_sizes_match_the_oracle(gtx_ctx):
    """What the tensors and `rtdetr.meta` say instead of the defaults: 80 classes (COCO's head), 100 queries, three decoder layers,
    max_det below the number of queries that clear conf -- probed layers, queries, boxes, scores and the truncated output."""
    from geotrax_amd.detector import Detector
    from geotrax_amd.weights import synthetic_rtdetr
    from oracle.rtdetr_ref import RtDetrRef, postprocess, stretch

    w = synthetic_rtdetr(seed=11, nc=80, nq=100, ndl=3, score_bias=-0.5)
    frame = _frame(3)
    det = Detector(w, FRAME_HW, imgsz=384, conf=0.05, max_det=20, ctx=gtx_ctx)
    assert det.n_queries == 100 and det.nc == 80
    ref = RtDetrRef(w)
    assert (ref.nq, ref.ndl, ref.nc) == (100, 3, 80)
    got, pred, _ = _check_against_oracle(det, ref, frame, 384, 0.05, None, layers=["model.9", "model.11", "model.27"])
    assert len(got) == 20 and len(postprocess(pred, frame.shape[:2], 0.05, None, 300)[1]) > 20      # max_det cut the list
    det.close()


def test_rtdetr_half_precision_maps(gtx_ctx, weights):
    """`ultralytics.half: true` (default.yaml:245) for RT-DETR: fp16 maps and weights on the fp16 MFMA convolutions (fp32 accumulate),
    the token side stays fp32 -- so this path is MORE exact than a half() model upstream, and it is held against the fp32 oracle at
    fp16's accumulated rounding (3e-2 of the layer maximum; YOLOv8's half bar). The decoder's input changes with it, so queries are
    compared as a set: most selected anchors are the fp32 run's."""
    from geotrax_amd.detector import Detector
    from oracle.rtdetr_ref import RtDetrRef, stretch

    frame = _frame(0)
    det = Detector(weights, FRAME_HW, imgsz=640, conf=0.3, max_det=300, half=True, ctx=gtx_ctx)
    assert det.rtdetr and not det.fp32_split
    got = det.detect(frame)
    ref = RtDetrRef(weights)
    pred = ref.forward(stretch(frame, 640))[0].numpy()
    for name in ["model.0", "model.1", "model.3", "model.7", "model.9", "model.11", "model.16", "model.21", "model.27"]:
        a = det.layer_output(name)
        r = ref.acts[name][0].permute(1, 2, 0).numpy()
        assert a.shape == r.shape and _rel(a, r) <= 3e-2, (name, _rel(a, r))
    idx = set(det.layer_output_int("model.28.topk").ravel().tolist())
    assert len(idx & set(ref.topk[0].numpy().tolist())) >= 240           # of 300
    raw = det.raw_output()
    assert raw.shape == pred.shape and np.isfinite(raw).all() and (raw[:, :4] >= 0).all() and (raw[:, :4] <= 1).all()
    n_ref = int((pred[:, 4:].max(1) > 0.3).sum())
    assert abs(len(got) - n_ref) <= max(5, n_ref // 4)
    det.close()
