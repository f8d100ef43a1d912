"""GPU global motion compensation (gtx_gmc_*) against oracle/gmc_ref.py stage by stage and against the
synthetic camera (absolute accuracy). Reference path: BOTSORT.update -> GMC.apply (extract.py:153,
default.yaml:362-374)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HW = (360, 640)


@pytest.fixture(scope="module")
def frames():
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=3, h=HW[0], w=HW[1])
    return sc, [sc.render(t) for t in (0, 50, 100)]


def test_corners_flow_and_warp_match_the_oracle(gtx_ctx, frames):
    from geotrax_amd.gmc import GMC
    from oracle.gmc_ref import GmcRef
    from oracle.yolov8_ref import bgr2gray_half

    sc, fr = frames
    g, o = GMC(HW, ctx=gtx_ctx), GmcRef(seed=0)
    A0 = g.apply(fr[0])
    np.testing.assert_array_equal(A0, np.eye(2, 3))                  # first frame: identity
    assert not g.valid
    o.apply(bgr2gray_half(fr[0]))
    for k in (1, 2):
        A = g.apply(fr[k])
        Ao = o.apply(bgr2gray_half(fr[k]))
        # corners: integer structure tensor, float64 eigenvalue -> same set in the same order
        cur, _ = g.points(0)
        np.testing.assert_array_equal(cur, o.prev_pts)
        prev, _ = g.points(1)
        np.testing.assert_array_equal(prev, o.last["prev"])
        nxt, st = g.points(2)
        np.testing.assert_array_equal(st, o.last["status"])
        assert st.sum() > 300
        np.testing.assert_allclose(nxt[st], o.last["next"][st], atol=2e-3)   # f64 LK, different summation order
        np.testing.assert_allclose(A, Ao, rtol=0, atol=2e-4)
        assert g.valid and g.stats[1] == st.sum() and g.stats[2] > 0.8 * st.sum()


def test_warp_recovers_the_synthetic_camera(gtx_ctx, frames):
    from geotrax_amd.gmc import GMC

    sc, fr = frames
    g = GMC(HW, ctx=gtx_ctx)
    g.apply(fr[0])
    A = g.apply(fr[2])
    G = sc.camera(100) @ np.linalg.inv(sc.camera(0))                 # frame 0 -> frame 100 pixels
    ys, xs = np.meshgrid(np.linspace(0, HW[0] - 1, 5), np.linspace(0, HW[1] - 1, 7), indexing="ij")
    P = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    q = G @ P
    assert np.abs(A @ P - q[:2] / q[2]).max() < 0.35                 # px; a similarity fitted to a tiny homography


def test_state_rules(gtx_ctx, frames):
    from geotrax_amd._lib import GtxError
    from geotrax_amd.gmc import GMC

    sc, fr = frames
    g = GMC(HW, ctx=gtx_ctx)
    with pytest.raises(GtxError):
        g.collect()                                                   # nothing submitted
    g.apply(fr[0])
    flat = np.full((HW[0], HW[1], 3), 100, np.uint8)                  # no corners -> nothing to track -> identity
    A = g.apply(flat)
    A2 = g.apply(fr[1])
    np.testing.assert_array_equal(A2, np.eye(2, 3))
    assert not g.valid
    g.reset_params()
    np.testing.assert_array_equal(g.apply(fr[1]), np.eye(2, 3))      # first frame again
    with pytest.raises(ValueError):
        GMC(HW, method="ecc", ctx=gtx_ctx)                            # orb / sift / ecc: gmc.make_gmc -> FeatureGMC / EccGMC (below, tests/test_ecc_gpu.py)
    with pytest.raises(GtxError):
        g.apply(np.zeros((100, 100, 3), np.uint8))                    # wrong frame size


@pytest.mark.parametrize("method", ["orb", "sift"])
def test_feature_gmc_matches_the_oracle_and_the_camera(gtx_ctx, frames, method):
    """`gmc_method: orb` / `sift` (default.yaml:374): the stabilizer's ORB kernels / csrc/sift.hip + the L2 matcher, ratio 0.9,
    apply_features' spatial filters and the partial-affine RANSAC -- against oracle/gmc_ref.py GmcFeatureRef (same matches, same
    filtered pairs, the same warp) and against the synthetic camera."""
    from geotrax_amd.gmc import make_gmc
    from oracle.gmc_ref import GmcFeatureRef
    from oracle.yolov8_ref import bgr2gray_half

    sc, fr = frames
    hw = HW if method == "orb" else (HW[0] // 2 * 2, HW[1] // 2 * 2)
    g, o = make_gmc(hw, method=method, ctx=gtx_ctx), GmcFeatureRef(hw, method=method)
    A0 = g.apply(fr[0])
    np.testing.assert_array_equal(A0, np.eye(2, 3))
    assert not g.valid
    o.apply(bgr2gray_half(fr[0]))
    for k in (1, 2):
        A = g.apply(fr[k])
        Ao = o.apply(bgr2gray_half(fr[k]))
        assert g.valid and g.stats[1] > 30 and g.stats[2] > 0.5 * g.stats[1]
        if method == "orb":                                              # integer stages: the same pairs survive the filters
            assert g.stats[1] == int(o.last["keep"].sum())
            np.testing.assert_allclose(A, Ao, rtol=0, atol=1e-9)
        else:                                                            # float32 pyramids: a keypoint or two may differ
            assert abs(int(g.stats[1]) - int(o.last["keep"].sum())) <= max(3, int(0.05 * g.stats[1]))
            P = np.array([[0, 0, 1], [hw[1], 0, 1], [0, hw[0], 1], [hw[1], hw[0], 1.0]]).T
            assert np.abs(A @ P - Ao @ P).max() < 0.25
    G = sc.camera(100) @ np.linalg.inv(sc.camera(50))                    # frame 50 -> frame 100 pixels
    ys, xs = np.meshgrid(np.linspace(0, HW[0] - 1, 5), np.linspace(0, HW[1] - 1, 7), indexing="ij")
    P = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    q = G @ P
    assert np.abs(A @ P - q[:2] / q[2]).max() < 0.8                      # px; a similarity fitted to a tiny homography from integer keypoints of the half-resolution pyramid
    g.close()


def test_feature_gmc_through_the_model_object(gtx_ctx):
    """BoT-SORT with `gmc_method: orb`: model.track hands the detector's gray image in HBM to the feature GMC; a method outside the reference's list is refused by name."""
    from geotrax_amd.model import YOLO
    from geotrax_amd.synth import make_scene
    from geotrax_amd.weights import synthetic_yolov8

    sc = make_scene(seed=3, h=HW[0], w=HW[1])
    m = YOLO(synthetic_yolov8(seed=1, nc=4), ctx=gtx_ctx)
    kw = dict(imgsz=384, conf=0.25, iou=0.7, max_det=300, classes=None, agnostic_nms=True, half=False, rect=True,
              tracker=dict(tracker_type="botsort", gmc_method="orb", track_high_thresh=0.25, track_low_thresh=0.1, new_track_thresh=0.25, track_buffer=30,
                           match_thresh=0.8, fuse_score=True))
    warps = []
    for t in (0, 20, 40):
        m.track(sc.render(t), persist=True, **kw)
        warps.append(m._gmc.valid)
    assert warps == [False, True, True] and type(m._gmc).__name__ == "FeatureGMC"
    m.detector.close()
    with pytest.raises(NotImplementedError):
        YOLO(synthetic_yolov8(seed=1, nc=4), ctx=gtx_ctx)._make_tracker(dict(tracker_type="botsort", gmc_method="akaze"))
