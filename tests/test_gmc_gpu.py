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
    with pytest.raises(NotImplementedError):
        GMC(HW, method="orb", ctx=gtx_ctx)
    with pytest.raises(GtxError):
        g.apply(np.zeros((100, 100, 3), np.uint8))                    # wrong frame size
