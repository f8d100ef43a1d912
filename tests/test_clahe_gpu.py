"""CLAHE (stabilo `clahe: true`, the reference's `stable` preset): the HIP kernels (gtx_op_clahe, and the stabilizer's
pre-processing step) against oracle/clahe_ref.py, bit for bit, and the stabilizer with the preset's settings against the
oracle chain. Parity unpinned against cv2 itself (not installed here): see the oracle's header."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1080, 1920), (2160, 3840), (203, 317), (64, 64), (135, 241), (540, 960), (77, 160)])
def test_clahe_kernel_equals_oracle(gtx_ctx, shape):
    from geotrax_amd import ops
    from oracle.clahe_ref import clahe

    rng = np.random.default_rng(shape[0])
    yy, xx = np.mgrid[:shape[0], :shape[1]]
    g = (110 + 60 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + rng.normal(0, 12, shape)).clip(0, 255).astype(np.uint8)
    g[: shape[0] // 5, : shape[1] // 4] = 7                      # a flat dark corner: clipping redistributes almost everything
    got = ops.clahe(g, ctx=gtx_ctx)
    np.testing.assert_array_equal(got, clahe(g))
    assert got.std() > g.std()                                   # it does equalise


def test_stabilizer_with_clahe_matches_oracle_chain(gtx_ctx):
    """`stable`-preset switches (clahe, full resolution, stricter ratio) on a small clip: same keypoints, descriptors and
    matches as the oracle chain with the CLAHE oracle in front, and the known camera motion is recovered."""
    from geotrax_amd.stabilizer import Stabilizer
    from geotrax_amd.synth import make_scene
    from oracle.stabilo_ref import StabilizerRef

    hw = (540, 960)
    cfg = dict(downsample_ratio=1.0, max_features=800, ref_multiplier=2.0, filter_ratio=0.8, ransac_threshold=2.0, mask_use=True,
               mask_margin_ratio=0.15, fast_threshold=20, n_levels=8, scale_factor=1.2, seed=0, clahe=True)
    sc = make_scene(seed=5, h=hw[0], w=hw[1])
    f0, f1 = sc.render(0), sc.render(30)
    st = Stabilizer(hw, downsample_ratio=1.0, max_features=800, filter_ratio=0.8, clahe=True, ctx=gtx_ctx)
    ref = StabilizerRef(cfg, hw, n_hyp=2048)
    st.set_ref_frame(f0, sc.boxes(0))
    ref.set_ref_frame(f0, sc.boxes(0))
    st.stabilize(f1, sc.boxes(30))
    H_ref, _ = ref.stabilize(f1, sc.boxes(30))
    for which, o in (("ref", ref.ref), ("cur", ref.cur)):
        g = st.keypoints(which)
        assert len(g["bin"]) == len(o["bin"]) > 300
        np.testing.assert_array_equal(g["xy"], o["xy"])
        np.testing.assert_array_equal(g["desc"], o["desc"])
    q, t, d = st.matches()
    np.testing.assert_array_equal(q, ref.m[0])
    np.testing.assert_array_equal(t, ref.m[1])
    H = st.get_cur_trans_matrix()
    assert H is not None and H_ref is not None
    ys, xs = np.meshgrid(np.linspace(0, hw[0] - 1, 9), np.linspace(0, hw[1] - 1, 16), indexing="ij")
    p = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    a, b = H @ p, np.linalg.inv(sc.camera(30)) @ p
    assert np.abs(a[:2] / a[2] - b[:2] / b[2]).max() < 1.0       # the scene's own camera model, independent of the oracle


def test_reference_frame_takes_8000_features(gtx_ctx):
    """max_features 4000 x ref_multiplier 2 (the `stable` preset) is inside the stabilizer's capacity now."""
    from geotrax_amd.stabilizer import Stabilizer
    from geotrax_amd.synth import make_scene

    hw = (1080, 1920)
    sc = make_scene(seed=2, h=hw[0], w=hw[1])
    st = Stabilizer(hw, downsample_ratio=1.0, max_features=4000, filter_ratio=0.8, clahe=True, ctx=gtx_ctx)
    st.set_ref_frame(sc.render(0), sc.boxes(0))
    st.stabilize(sc.render(10), sc.boxes(10))
    n_ref, n_cur = st.get_cur_num_keypoints()
    assert n_ref > 4000 and n_cur > 2000 and st.get_cur_trans_matrix() is not None
