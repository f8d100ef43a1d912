"""GPU global motion compensation, method 'ecc' (gtx_ecc_*, csrc/ecc.hip) against oracle/ecc_ref.py: the prepared image bit for
bit, the fitted warp, the iteration count and the correlation coefficient; the two error conditions; the object's sequence rules;
BoT-SORT with `gmc_method: ecc` through the model object and through the engine. Reference path: BOTSORT.update -> GMC.apply ->
apply_ecc (extract.py:153, default.yaml:374)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene_frames(hw, ts, seed=3):
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=seed, h=hw[0], w=hw[1])
    return sc, [sc.render(t) for t in ts]


@pytest.mark.parametrize("warp", ["exact", "fixed"])
@pytest.mark.parametrize("hw", [(216, 384), (361, 641), (540, 960)])
def test_prepared_image_and_warp_match_the_oracle(gtx_ctx, hw, warp):
    """cvtColor -> GaussianBlur(3x3, 1.5) -> resize(1/2) is integer arithmetic: equal bit for bit (odd frame sizes drop the last row /
    column). The fit runs the same float32 / float64 operations in the same order with its float64 sums taken in another order:
    the same number of iterations (converged, or the cap of 300 set here), the coefficient to 1e-9, the map to 2e-6 (its float32
    entries differ by an ulp or two at most)."""
    from geotrax_amd.gmc import make_gmc
    from oracle.ecc_ref import EccRef, prepare

    _, fr = _scene_frames(hw, (0, 40, 80))
    g, o = make_gmc(hw, method="ecc", ctx=gtx_ctx, max_iters=300, warp=warp), EccRef(max_iters=300, warp=warp)   # the reference's 5000 would only
    # repeat a limit cycle: with the 1/32-pixel source positions of OpenCV <= 4.10 ("fixed") a few pixels keep flipping, rho moves by more than
    # 1e-6 and the loop runs to its cap; with floating-point positions ("exact", OpenCV >= 4.11, the default) these fits end after ~10 iterations
    np.testing.assert_array_equal(g.apply(fr[0]), np.eye(2, 3))
    assert not g.valid and g.last["iters"] == 0
    o.apply(fr[0])
    np.testing.assert_array_equal(g.image(0), prepare(fr[0]).astype(np.float32))
    np.testing.assert_array_equal(g.image(1), prepare(fr[0]).astype(np.float32))
    for k in (1, 2):
        A, Ao = g.apply(fr[k]), o.apply(fr[k])
        np.testing.assert_array_equal(g.image(0), prepare(fr[k]).astype(np.float32))
        assert g.valid and g.last["status"] == o.last["status"] == 0
        assert g.last["iters"] == o.last["iters"] and 2 <= g.last["iters"] <= (40 if warp == "exact" else 300)
        assert abs(g.last["rho"] - o.last["rho"]) < 1e-9 and g.last["rho"] > 0.5
        np.testing.assert_allclose(A, Ao, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(g.image(1), prepare(fr[0]).astype(np.float32))      # the template is still the first frame
    g.close()


def test_warp_recovers_the_synthetic_camera_at_4k(gtx_ctx):
    """3840 x 2160: frame t against frame 0 of the synthetic clip. The camera of the clip is a small homography; the Euclidean fit
    must put the image corners within 1.5 half-resolution pixels of it, and a second look at the same frame gives the same warp."""
    from geotrax_amd.gmc import make_gmc

    hw = (2160, 3840)
    sc, fr = _scene_frames(hw, (0, 30))
    g = make_gmc(hw, method="ecc", ctx=gtx_ctx)                          # the defaults: 5000 iterations / 1e-6, floating-point source positions
    g.apply(fr[0])
    A = g.apply(fr[1])
    assert g.valid and g.last["rho"] > 0.5 and g.last["iters"] < 60      # the vehicles move against the background: the coefficient is well below 1
    G = sc.camera(30) @ np.linalg.inv(sc.camera(0))                      # frame 0 -> frame 30, full-resolution pixels
    ys, xs = np.meshgrid(np.linspace(100, hw[0] - 100, 5), np.linspace(100, hw[1] - 100, 7), indexing="ij")
    P = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    q = G @ P
    q = q[:2] / q[2]
    half = lambda p: (p - 0.5) / 2.0                                      # pixel centres of the 2 x 2 reduction
    got = A @ np.vstack([half(P[:2]), np.ones(P.shape[1])])
    assert np.abs(got - half(q)).max() < 1.5
    np.testing.assert_array_equal(g.apply(fr[1]), A)
    g.close()


def test_error_conditions_and_sequence_rules(gtx_ctx):
    from geotrax_amd._lib import GtxError
    from geotrax_amd.gmc import EccGMC
    from oracle.ecc_ref import EccRef

    hw = (128, 192)
    g = EccGMC(hw, ctx=gtx_ctx, max_iters=50)
    with pytest.raises(GtxError):
        g.collect()                                                       # nothing submitted
    flat = np.full((hw[0], hw[1], 3), 100, np.uint8)
    g.apply(flat)
    np.testing.assert_array_equal(g.apply(flat), np.eye(2, 3))            # zero variance: NaN correlation -> cv2.error upstream, identity kept
    assert g.last["status"] == 1 and g.last["iters"] == 1 and not g.valid
    g.reset_params()
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    o = EccRef(max_iters=50)
    g.apply(a)
    o.apply(a)
    A, Ao = g.apply(255 - a), o.apply(255 - a)                            # anti-correlated: the second cv2.error
    assert g.last["status"] == o.last["status"] == 2 and g.last["iters"] == o.last["iters"]
    np.testing.assert_allclose(A, Ao, rtol=0, atol=2e-6)
    with pytest.raises(GtxError):
        g.apply(np.zeros((100, 100, 3), np.uint8))                        # wrong frame size
    with pytest.raises(NotImplementedError):
        g.submit_gray_dev(0, hw[0] // 2, hw[1] // 2)                      # the blur comes before the reduction: frames only
    # frames queued ahead come back in order; the iteration cap is honoured
    _, fr = _scene_frames(hw, (0, 10, 20, 30))
    g.reset_params()
    for f in fr:
        g.submit_frame(f)
    got = [g.collect() for _ in fr]
    g.reset_params()
    want = [g.apply(f) for f in fr]
    for x, y in zip(got, want):
        np.testing.assert_array_equal(x, y)
    # the switch upstream does not have: frame-to-frame templates
    g2, o2 = EccGMC(hw, ctx=gtx_ctx, max_iters=100, replace_template=True), EccRef(max_iters=100, replace_template=True)
    for f in fr:
        np.testing.assert_allclose(g2.apply(f), o2.apply(f), rtol=0, atol=2e-6)
    np.testing.assert_array_equal(g2.image(1), g2.image(0))              # the last frame is the template now
    g2.close()
    g1 = EccGMC(hw, ctx=gtx_ctx, max_iters=3)
    g1.apply(fr[0])
    g1.apply(fr[3])
    assert g1.last["iters"] == 3 and g1.last["status"] == 0
    g.close()
    g1.close()


def test_ecc_through_the_model_object_and_the_engine(gtx_ctx):
    """BoT-SORT with `gmc_method: ecc`: model.track (frame at a time) and the engine (batches of two on two detector streams, the
    frames handed to the GMC on the detectors' streams) must give the same warps and the same tracks, frame for frame."""
    from geotrax_amd.engine import ExtractEngine
    from geotrax_amd.model import YOLO
    from geotrax_amd.tracker import Tracker
    from geotrax_amd.weights import synthetic_yolov8

    hw = (432, 768)
    sc, fr = _scene_frames(hw, range(0, 48, 4))
    w = synthetic_yolov8(seed=1, nc=4)
    tk = dict(tracker_type="botsort", gmc_method="ecc", track_high_thresh=0.25, track_low_thresh=0.1, new_track_thresh=0.25, track_buffer=30,
              match_thresh=0.8, fuse_score=True)
    det_kw = dict(imgsz=384, conf=0.25, iou=0.7, max_det=300, classes=None, agnostic_nms=True, half=False, rect=True)
    m = YOLO(w, ctx=gtx_ctx)
    one = []
    for f in fr:
        r = m.track(f, persist=True, tracker=tk, **det_kw)[0]
        one.append((m._gmc.last.copy(), None if r.boxes._id is None else r.boxes._id.astype(np.int64), r.boxes._xyxy.copy()))
    assert type(m._gmc).__name__ == "EccGMC" and sum(l["iters"] > 0 for l, _, _ in one) == len(fr) - 1
    m.detector.close()
    tr = Tracker("botsort", track_high_thresh=0.25, track_low_thresh=0.1, new_track_thresh=0.25, track_buffer=30, match_thresh=0.8, fuse_score=True)
    eng = ExtractEngine(w, hw, det_kw, tr, None, batch=2, det_streams=2, gmc="ecc")
    res = list(eng.run([fr[i:i + 2] for i in range(0, len(fr), 2)]))
    assert len(res) == len(fr)
    for r, (last, ids, xyxy) in zip(res, one):
        if ids is None:
            assert r.ids is None
        else:
            np.testing.assert_array_equal(np.asarray(r.ids, np.int64), ids)
            np.testing.assert_array_equal(r.xyxy, xyxy)
    assert all(r.gmc is not None for r in res) and np.array_equal(res[0].gmc, np.eye(2, 3)) and not np.array_equal(res[-1].gmc, np.eye(2, 3))
    eng.close()
