"""GPU stabilizer (gtx_stabilizer_*) against oracle/stabilo_ref.py stage by stage (keypoints,
orientation bins, descriptors and matches are integer work: bit-exact) and against the ground
truth homography of seeded synthetic sequences (absolute accuracy, independent of the oracle)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CFG = dict(downsample_ratio=0.5, max_features=600, ref_multiplier=2.0, filter_ratio=0.9, ransac_threshold=2.0,
           mask_use=True, mask_margin_ratio=0.15, fast_threshold=20, n_levels=8, scale_factor=1.2, seed=0)
HW = (720, 1280)


def _grid_err(Ha, Hb, hw):
    ys, xs = np.meshgrid(np.linspace(0, hw[0] - 1, 9), np.linspace(0, hw[1] - 1, 16), indexing="ij")
    p = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    a, b = Ha @ p, Hb @ p
    return np.abs(a[:2] / a[2] - b[:2] / b[2]).max()


@pytest.fixture(scope="module")
def seq():
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=3, h=HW[0], w=HW[1])
    return sc, {t: sc.render(t) for t in (0, 40, 149)}


def _make(gtx_ctx, **over):
    from geotrax_amd.stabilizer import Stabilizer

    kw = dict(downsample_ratio=CFG["downsample_ratio"], max_features=CFG["max_features"], ref_multiplier=CFG["ref_multiplier"],
              filter_ratio=CFG["filter_ratio"], ransac_epipolar_threshold=CFG["ransac_threshold"], mask_use=CFG["mask_use"],
              mask_margin_ratio=CFG["mask_margin_ratio"], seed=CFG["seed"], ctx=gtx_ctx)
    kw.update(over)
    return Stabilizer(HW, **kw)


def test_stages_bit_exact_against_oracle(gtx_ctx, seq):
    from oracle.stabilo_ref import StabilizerRef

    sc, fr = seq
    st = _make(gtx_ctx)
    ref = StabilizerRef(CFG, HW, n_hyp=2048)                 # the oracle's own sampling table (oracle.stabilo_ref.brief_pattern)
    np.testing.assert_array_equal(st.pattern(), ref.pattern)
    b0, b1 = sc.boxes(0), sc.boxes(40)
    st.set_ref_frame(fr[0], b0)
    ref.set_ref_frame(fr[0], b0)
    st.stabilize(fr[40], b1)
    H_ref, n_inl = ref.stabilize(fr[40], b1)
    for which, o in (("ref", ref.ref), ("cur", ref.cur)):
        g = st.keypoints(which)
        assert len(g["bin"]) == len(o["bin"]) > 300, which
        np.testing.assert_array_equal(g["level"], o["level"])
        np.testing.assert_array_equal(g["xy"], o["xy"])          # same pixels, same fp32 scaling
        np.testing.assert_array_equal(g["bin"], o["bin"])
        np.testing.assert_array_equal(g["desc"], o["desc"])
    q, t, d = st.matches()
    np.testing.assert_array_equal(q, ref.m[0])
    np.testing.assert_array_equal(t, ref.m[1])
    np.testing.assert_array_equal(d, ref.m[2])
    assert len(q) > 100
    H = st.get_cur_trans_matrix()
    assert H is not None and H_ref is not None
    # f64 homography: same hypotheses, same inlier set, two linear-algebra back ends
    assert _grid_err(H, H_ref, HW) < 1e-3
    assert abs(st.get_cur_inliers_count() - n_inl) <= 2
    assert st.get_cur_num_keypoints() == (len(ref.ref["bin"]), len(ref.cur["bin"]))
    assert st.get_cur_num_matches() == len(q)


def test_mask_excludes_vehicle_boxes(gtx_ctx, seq):
    sc, fr = seq
    st = _make(gtx_ctx)
    boxes = sc.boxes(0)
    st.set_ref_frame(fr[0], boxes)
    xy = st.keypoints("ref")["xy"]
    inside = np.zeros(len(xy), bool)
    for cx, cy, w, h in boxes:
        inside |= (np.abs(xy[:, 0] - cx) < w / 2) & (np.abs(xy[:, 1] - cy) < h / 2)
    assert not inside.any()
    st2 = _make(gtx_ctx, mask_use=False)
    st2.set_ref_frame(fr[0], boxes)
    xy2 = st2.keypoints("ref")["xy"]
    inside2 = np.zeros(len(xy2), bool)
    for cx, cy, w, h in boxes:
        inside2 |= (np.abs(xy2[:, 0] - cx) < w / 2) & (np.abs(xy2[:, 1] - cy) < h / 2)
    assert inside2.any()   # vehicle corners are strong features: the mask matters


def test_mask_with_hundreds_of_boxes_over_one_tile_equals_oracle(gtx_ctx, seq):
    """The foreground test runs against a per-tile list of the boxes that reach the tile (256 entries); a tile under more
    boxes than that tests against all of them. 700 small boxes packed into a corner of the frame, plus the scene's own:
    keypoints, bins and descriptors equal the oracle's (which draws every box into a mask image), none lies in a box."""
    from oracle.stabilo_ref import StabilizerRef

    sc, fr = seq
    rng = np.random.default_rng(9)
    dense = np.stack([rng.uniform(40, 420, 700), rng.uniform(40, 300, 700), rng.uniform(6, 14, 700), rng.uniform(6, 14, 700)], 1).astype(np.float32)
    boxes = np.concatenate([sc.boxes(0), dense]).astype(np.float32)
    st = _make(gtx_ctx)
    ref = StabilizerRef(CFG, HW, n_hyp=2048)
    st.set_ref_frame(fr[0], boxes)
    ref.set_ref_frame(fr[0], boxes)
    g, o = st.keypoints("ref"), ref.ref
    assert len(g["bin"]) == len(o["bin"]) > 200
    np.testing.assert_array_equal(g["xy"], o["xy"])
    np.testing.assert_array_equal(g["bin"], o["bin"])
    np.testing.assert_array_equal(g["desc"], o["desc"])
    xy = g["xy"]
    inside = np.zeros(len(xy), bool)
    for cx, cy, w, h in boxes:
        inside |= (np.abs(xy[:, 0] - cx) < w / 2) & (np.abs(xy[:, 1] - cy) < h / 2)
    assert not inside.any()


@pytest.mark.parametrize("t", [40, 149])
def test_recovers_ground_truth_homography(gtx_ctx, seq, t):
    """cur -> ref mapping must equal inv(G_t) of the synthetic camera within 1.0 px over a 9x16
    grid (the BASELINE.md §5 bar; the RANSAC inlier threshold is 2 px). Integer-pixel keypoints on
    a 640x360 working image with 600 features leave ~0.5-0.9 px at the frame corners."""
    sc, fr = seq
    st = _make(gtx_ctx)
    st.set_ref_frame(fr[0], sc.boxes(0))
    st.stabilize(fr[t], sc.boxes(t))
    H = st.get_cur_trans_matrix()
    assert H is not None
    assert _grid_err(H, np.linalg.inv(sc.camera(t)), HW) < 1.0
    assert st.get_cur_inliers_count() > 50
    # boxes of static vehicles come back to their frame-0 position
    static = np.abs(sc.veh_vel).sum(1) == 0
    if static.any():
        warped = st.transform_cur_boxes()
        np.testing.assert_allclose(warped[static, :2], sc.boxes(0)[static, :2], atol=1.0)


def test_homographies_of_the_synthetic_clip_stay_inside_the_golden_envelope(gtx_ctx, seq):
    """The reference's golden `_vid_transf` table (tests/golden, 149 frames of hovering-drone footage) has a shape every
    sane stabilization of such footage shares: det ~ 1, h33 = 1, rotation and scale within a few 1e-3, a perspective part
    that moves no pixel by more than a pixel or two, translations of a few pixels that drift smoothly. The synthetic clip
    is built to that envelope; what this build estimates on it must stay inside it (with the estimator's own noise)."""
    from pathlib import Path

    from geotrax_amd import agreement as A

    gold = A.homography_envelope(np.loadtxt(Path(__file__).parent / "golden" / "U_video_cut_vid_transf.txt", delimiter=","))
    sc, fr = seq
    st = _make(gtx_ctx)
    st.set_ref_frame(fr[0], sc.boxes(0))
    rows = []
    for t in sorted(k for k in fr if k > 0):
        st.stabilize(fr[t], sc.boxes(t))
        Hm = st.get_cur_trans_matrix()
        assert Hm is not None
        rows.append(np.r_[t, Hm.ravel()])
    e = A.homography_envelope(np.asarray(rows))
    w = HW[1]
    assert e["det_min"] > 0.98 and e["h33_dev_max"] < 1e-9
    assert e["rotation_abs_max"] < max(3 * gold["rotation_abs_max"], 5e-3) and e["scale_dev_max"] < max(3 * gold["scale_dev_max"], 5e-3)
    assert e["perspective_abs_max"] * w * w < 3.0                       # the projective part moves no pixel by more than ~3 px
    assert e["translation_abs_max"][0] < gold["translation_abs_max"][0] + 1.5 and e["translation_abs_max"][1] < gold["translation_abs_max"][1] + 1.5


def test_identical_frame_gives_identity_and_state_errors(gtx_ctx, seq):
    from geotrax_amd._lib import GtxError
    from geotrax_amd.stabilizer import Stabilizer

    sc, fr = seq
    st = _make(gtx_ctx)
    with pytest.raises(GtxError):
        st.stabilize(fr[0], None)                      # no reference frame yet
    st.set_ref_frame(fr[0], None)
    st.stabilize(fr[0], None)
    H = st.get_cur_trans_matrix()
    assert _grid_err(H, np.eye(3), HW) < 1e-6
    flat = np.full((HW[0], HW[1], 3), 127, np.uint8)   # no texture -> no keypoints -> no transform of its own
    st.stabilize(flat, None)
    assert st.get_cur_trans_matrix(raw=True) is None and not st.registered and st.get_cur_num_matches() == 0
    np.testing.assert_array_equal(st.get_cur_trans_matrix(), H)      # stabilo's trans_matrix_last_known
    st.set_ref_frame(fr[0], None)                      # a new reference frame forgets it: None (extract.py:185 skips the row)
    st.stabilize(flat, None)
    assert st.get_cur_trans_matrix() is None
    with pytest.raises(NotImplementedError):
        Stabilizer(HW, detector_name="brisk")                         # orb, sift and rsift are built (round 6)


@pytest.mark.parametrize("opts", [dict(transformation_type="affine"), dict(filter_type="none"),
                                  dict(transformation_type="affine", filter_type="none")])
def test_affine_model_and_unfiltered_matches_equal_the_oracle(gtx_ctx, seq, opts):
    """stabilo `transformation_type: affine` (default.yaml:121) and `filter_type: none` (default.yaml:117): same matches
    as the oracle (with `none`: every query keypoint's nearest neighbour), same estimate from the same hypotheses; the
    affine matrix has last row (0, 0, 1) exactly and, the synthetic camera being nearly a similarity, still lands within
    1.5 px of the ground truth."""
    from oracle.stabilo_ref import StabilizerRef

    sc, fr = seq
    st = _make(gtx_ctx, **opts)
    ref = StabilizerRef(dict(CFG, **opts), HW, n_hyp=2048)
    b0, b1 = sc.boxes(0), sc.boxes(40)
    st.set_ref_frame(fr[0], b0)
    ref.set_ref_frame(fr[0], b0)
    st.stabilize(fr[40], b1)
    H_ref, n_inl = ref.stabilize(fr[40], b1)
    q, t, d = st.matches()
    np.testing.assert_array_equal(q, ref.m[0])
    np.testing.assert_array_equal(t, ref.m[1])
    np.testing.assert_array_equal(d, ref.m[2])
    if opts.get("filter_type") == "none":
        assert len(q) == st.get_cur_num_keypoints()[1]
    H = st.get_cur_trans_matrix()
    assert H is not None and H_ref is not None
    assert _grid_err(H, H_ref, HW) < 1e-3
    assert abs(st.get_cur_inliers_count() - n_inl) <= 2
    if opts.get("transformation_type") == "affine":
        np.testing.assert_array_equal(H[2], [0.0, 0.0, 1.0])
        assert _grid_err(H, np.linalg.inv(sc.camera(40)), HW) < 1.5
    else:
        assert _grid_err(H, np.linalg.inv(sc.camera(40)), HW) < 1.0


@pytest.mark.parametrize("name", ["rsift", "sift"])
def test_sift_detectors_recover_the_ground_truth_homography(gtx_ctx, seq, name):
    """stabilo `detector_name: rsift` / `sift` (default.yaml:109): the registration stage's SIFT + L2 matcher + RANSAC on the
    half-resolution frames, one blocking call per frame. Known camera to 1 px on the 9 x 16 grid, stabilo's counters filled,
    the last-known-transform rule kept, box warp as with ORB."""
    sc, fr = seq
    st = _make(gtx_ctx, detector_name=name, max_features=3000, filter_ratio=0.75)
    st.set_ref_frame(fr[0], sc.boxes(0))
    for t in (40, 149):
        st.stabilize(fr[t], sc.boxes(t))
        H = st.get_cur_trans_matrix()
        assert H is not None and st.registered
        assert _grid_err(H, np.linalg.inv(sc.camera(t)), HW) < 1.0
        n_ref, n_cur = st.get_cur_num_keypoints()
        assert n_ref > 200 and n_cur > 200 and st.get_cur_num_matches() > 50 and st.get_cur_inliers_count() > 30
    st.stabilize(np.full_like(fr[0], 90), None)                       # nothing to register: the last known transform is reported
    assert not st.registered and np.array_equal(st.get_cur_trans_matrix(), H)
    with pytest.raises(NotImplementedError):
        st.set_ref_gray_dev(0, HW[0] // 2, HW[1] // 2)                 # host frames only on this path


def test_unsupported_stabilizer_choices_raise(gtx_ctx):
    _make(gtx_ctx, matcher_name="flann").close()                      # served by the exact matcher, with a warning
    for kw in (dict(detector_name="brisk"), dict(detector_name="akaze"), dict(matcher_name="annoy"), dict(filter_type="distance"),
               dict(detector_name="sift", transformation_type="affine")):
        with pytest.raises(NotImplementedError):
            _make(gtx_ctx, **kw)
    with pytest.raises(ValueError):
        _make(gtx_ctx, transformation_type="similarity")
