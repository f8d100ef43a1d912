"""The co-designed oracles and the HIP kernels held against an implementation nobody here wrote: scikit-image 0.18.3 /
scipy, run once in the build container by tests/golden/make_independent.py (fixture: tests/golden/independent_skimage.npz).

scikit-image is not a dependency of the reference (geo-trax runs OpenCV through stabilo / ultralytics), so these checks do
not pin parity with the reference's own stack; they rule out a misreading of the *published algorithms* shared by an oracle
and the kernel it was written next to: the FAST-9/16 segment test, what an ORB + ratio test + RANSAC
homography of these frames is, bilinear warping / resizing, BT.601. Inputs are rebuilt from seeds and checked by CRC32.

CPU tests: oracle vs fixture. `-m gpu` tests: the HIP path (through the C ABI) vs the same fixture.
Reference call sites: geotrax/extract.py:153,176-187 (stabilizer / detector), geotrax/cfg/default.yaml:100-145,
geotrax/visualize.py:285-289 (warp), geotrax/extract.py:146 (decoded frames).
"""
import ctypes as C
import importlib.util
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden"
HW = (720, 1280)
STAB_CFG = dict(downsample_ratio=0.5, max_features=600, ref_multiplier=2.0, filter_ratio=0.9, ransac_threshold=2.0,
                mask_use=True, mask_margin_ratio=0.15, fast_threshold=20, n_levels=8, scale_factor=1.2, seed=0)


@pytest.fixture(scope="module")
def mi():
    spec = importlib.util.spec_from_file_location("make_independent", GOLD / "make_independent.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def fx():
    return np.load(GOLD / "independent_skimage.npz")


@pytest.fixture(scope="module")
def inp(mi, fx):
    d = mi.inputs()
    for k in ("f0", "f1", "f40", "f149", "fast0", "fast1", "warp_src", "resize_src", "i420", "d_query", "d_train", "pts_p", "pts_q"):
        assert mi.crc(d[k]) == int(fx[k + "_crc"]), f"input {k} is not the one the fixture was made from: regenerate with make_independent.py"
    return d


def _grid(hw):
    ys, xs = np.meshgrid(np.linspace(0, hw[0] - 1, 9), np.linspace(0, hw[1] - 1, 16), indexing="ij")
    return np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])


def _proj(M, P):
    M = np.asarray(M, dtype=np.float64)
    if M.shape == (2, 3):
        return M @ P
    q = M @ P
    return q[:2] / q[2]


def _fast_set(fx, k):
    h, w = (int(v) for v in fx[k + "_shape"])
    return np.unpackbits(fx[k + "_bits"])[:h * w].reshape(h, w).astype(bool)


def _warp_interior(mi, fx, shape):
    """Destination pixels whose four source taps lie inside the image: skimage 0.18's constant-mode border blending is not
    OpenCV's, so the border band is left to the oracle / kernel comparison at 4K (tests/test_warp_gpu.py)."""
    h, w = shape[:2]
    Hinv = np.linalg.inv(fx["warp_H"])
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    W = Hinv[2, 0] * xx + Hinv[2, 1] * yy + Hinv[2, 2]
    X = (Hinv[0, 0] * xx + Hinv[0, 1] * yy + Hinv[0, 2]) / W
    Y = (Hinv[1, 0] * xx + Hinv[1, 1] * yy + Hinv[1, 2]) / W
    return (X >= 1) & (X <= w - 2) & (Y >= 1) & (Y <= h - 2)


# ------------------------------------------------------------------------------------------------ oracle vs scikit-image (CPU)

@pytest.mark.parametrize("k", ["fast0", "fast1"])
def test_oracle_fast_segment_test_equals_skimage_corner_fast(inp, fx, k):
    """oracle.stabilo_ref.fast_score > 0 must be exactly the set corner_fast(n=9) answers on (inside the 31-pixel border ORB
    keeps free): same 16-pixel circle, same 'nine contiguous pixels all brighter than p + t or all darker than p - t'."""
    from oracle.stabilo_ref import BORDER, fast_score

    g = inp[k]
    h, w = g.shape
    sk = _fast_set(fx, k)
    inner = np.zeros_like(sk)
    inner[BORDER:h - BORDER, BORDER:w - BORDER] = True
    ours = fast_score(g, STAB_CFG["fast_threshold"]) > 0
    assert ours.sum() > 1000
    np.testing.assert_array_equal(ours, sk & inner)


@pytest.mark.parametrize("t", [40, 149])
def test_oracle_homography_agrees_with_skimage_orb_ransac(inp, fx, t):
    """StabilizerRef (the oracle's ORB + ratio test + MSAC) and scikit-image's ORB + match_descriptors + ransac on the same two
    frames: <= 1 px apart on the 9 x 16 grid, and each <= 1 px from the clip's known camera (SURVEY 8d bar)."""
    from oracle.stabilo_ref import StabilizerRef

    sc = inp["scene"]
    ref = StabilizerRef(STAB_CFG, HW, n_hyp=2048)
    ref.set_ref_frame(inp["f0"], sc.boxes(0))
    H, n_inl = ref.stabilize(inp["f%d" % t], sc.boxes(t))
    P = _grid(HW)
    Hs, Ht = fx["orb_H_%d" % t], np.linalg.inv(sc.camera(t))
    assert int(fx["orb_H_%d_stats" % t][2]) > 500 and n_inl > 100
    assert np.abs(_proj(Hs, P) - _proj(Ht, P)).max() < 1.0           # the third-party estimate itself is a valid yardstick
    assert np.abs(_proj(H, P) - _proj(Ht, P)).max() < 1.0
    assert np.abs(_proj(H, P) - _proj(Hs, P)).max() < 1.0


def test_oracle_gmc_agrees_with_skimage_similarity(inp, fx, mi):
    """GmcRef (corners + pyramidal LK + similarity RANSAC) vs ORB + ransac(SimilarityTransform) for frames 0 -> 1."""
    from oracle.gmc_ref import GmcRef

    g = GmcRef()
    g.apply(mi.gray_u8(inp["f0"], True))
    A = g.apply(mi.gray_u8(inp["f1"], True))
    P = _grid(HW)
    S = fx["orb_S_1"][:2]
    G = inp["scene"].camera(1)
    assert np.abs(_proj(S, P) - _proj(G, P)).max() < 0.25
    assert np.abs(_proj(A, P) - _proj(S, P)).max() < 0.25
    assert np.abs(_proj(A, P) - _proj(G, P)).max() < 0.25


def _ecc_full(A, P):
    """An ECC warp (half-resolution pixels, first frame -> frame) applied to full-resolution points: the 2 x 2 reduction puts
    half-resolution pixel i at full-resolution coordinate 2 i + 0.5."""
    h = np.vstack([(P[:2] - 0.5) / 2.0, np.ones(P.shape[1])])
    return 2.0 * (np.asarray(A, np.float64) @ h) + 0.5


def test_oracle_ecc_agrees_with_skimage_similarity(inp, fx):
    """EccRef (cv2.findTransformECC restated: dense, iterative, Euclidean) vs ORB + ransac(SimilarityTransform) (sparse, feature
    based) for frames 0 -> 1 of the synthetic clip, and both vs the clip's camera: two unrelated estimators of one motion."""
    from oracle.ecc_ref import EccRef

    e = EccRef()                                                       # the defaults: 5000 / 1e-6, floating-point source positions
    e.apply(inp["f0"])
    A = e.apply(inp["f1"])
    assert e.last["status"] == 0 and e.last["rho"] > 0.9
    P = _grid(HW)
    assert np.abs(_ecc_full(A, P) - _proj(fx["orb_S_1"][:2], P)).max() < 0.3
    assert np.abs(_ecc_full(A, P) - _proj(inp["scene"].camera(1), P)).max() < 0.3


def test_oracle_warp_agrees_with_skimage_warp(inp, fx, mi):
    """oracle.warp_ref (OpenCV's 1/32-pixel coordinate quantisation) vs transform.warp(order=1) (exact float coordinates): the
    quantisation moves a value by at most |gradient| / 64, i.e. <= 1 grey level except on the sharpest edges of the crop (2)."""
    from oracle.warp_ref import warp_perspective

    out = warp_perspective(inp["warp_src"], fx["warp_H"])
    inside = _warp_interior(mi, fx, out.shape)
    d = np.abs(out.astype(np.int16) - fx["warp_out"].astype(np.int16)).max(-1)[inside]
    assert inside.mean() > 0.9 and d.max() <= 2 and (d > 1).mean() < 1e-3, (d.max(), (d > 1).mean())


def test_oracle_general_letterbox_resize_agrees_with_skimage_resize(inp, fx, mi):
    """The non-2x bilinear of the letterbox (yolov8_ref.resize_linear_u8, 'restated from memory' of OpenCV's 11-bit fixed
    point) vs transform.resize(order=1, anti_aliasing=False): same half-pixel-centre mapping, <= 1 grey level everywhere."""
    from oracle.yolov8_ref import resize_linear_u8

    out = resize_linear_u8(inp["resize_src"], *mi.RESIZE_TO)
    d = np.abs(out.astype(np.int16) - fx["resize_out"].astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 0.25, (d.max(), (d > 0).mean())


def test_oracle_yuv_agrees_with_skimage_ycbcr(inp, fx):
    from oracle.yuv_ref import i420_to_bgr

    yh, yw = (int(v) for v in inp["yuv_hw"])
    d = np.abs(i420_to_bgr(inp["i420"], yh, yw).astype(np.int16) - fx["yuv_bgr"].astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 0.1, (d.max(), (d > 0).mean())


def test_oracle_matcher_equals_skimage_match_descriptors(inp, fx):
    """The Hamming 2-NN + Lowe ratio stage on its own (oracle.stabilo_ref.match; the HIP matcher equals it bit for bit in
    tests/test_stabilizer_gpu.py): the same (query, train) pairs as skimage.feature.match_descriptors(metric='hamming',
    cross_check=False, max_ratio=0.9) on seeded 256-bit descriptors -- nearest neighbour with ties to the lowest index, strict
    ratio test."""
    from oracle.stabilo_ref import match

    q, t, d = match(inp["d_query"], inp["d_train"], 0.9)
    pairs = fx["match_pairs"]
    assert 200 < len(pairs) < 560                                      # the ratio test rejects the damaged / unrelated queries
    np.testing.assert_array_equal(np.stack([q, t], 1), pairs)
    bits = np.unpackbits(inp["d_query"][q] ^ inp["d_train"][t], axis=1).sum(1)
    np.testing.assert_array_equal(d, bits)


def test_oracle_estimator_agrees_with_skimage_ransac_on_given_matches(inp, fx):
    """The estimation stage on its own: oracle.stabilo_ref.ransac_homography (MSAC-scored hypotheses + IRLS refit) and
    skimage.measure.ransac(ProjectiveTransform) on the same 500 seeded matches (30 % outliers): within 0.3 px of each other
    and of the homography the matches were made with, on the 9 x 16 grid."""
    from oracle.stabilo_ref import ransac_homography

    H, n_inl = ransac_homography(inp["pts_p"], inp["pts_q"], (HW[1], HW[0]), 2.0, 2048, 0)
    P = _grid(HW)
    assert H is not None and abs(n_inl - int(fx["ransac_inliers"])) <= 10 and n_inl > 300
    assert np.abs(_proj(fx["ransac_H"], P) - _proj(inp["H_true"], P)).max() < 0.3
    assert np.abs(_proj(H, P) - _proj(inp["H_true"], P)).max() < 0.3
    assert np.abs(_proj(H, P) - _proj(fx["ransac_H"], P)).max() < 0.3


# ------------------------------------------------------------------------------------------------ HIP path vs scikit-image

def _stabilizer(gtx_ctx):
    from geotrax_amd.stabilizer import Stabilizer

    return Stabilizer(HW, downsample_ratio=0.5, max_features=600, ref_multiplier=2.0, filter_ratio=0.9, ransac_epipolar_threshold=2.0,
                      mask_use=True, mask_margin_ratio=0.15, seed=0, ctx=gtx_ctx)


@pytest.mark.gpu
def test_gpu_keypoints_are_skimage_fast_corners(gtx_ctx, inp, fx):
    """Every level-0 keypoint the HIP stabilizer keeps is a pixel scikit-image's FAST-9/16 fires on (the pyramid's level 0 is
    the half-resolution gray image the fixture's `fast0` was computed from)."""
    st = _stabilizer(gtx_ctx)
    st.set_ref_frame(inp["f0"], None)
    kp = st.keypoints("ref")
    l0 = kp["level"] == 0
    assert l0.sum() > 100
    px = np.rint(kp["xy"][l0] * 0.5).astype(int)                      # level-0 pixel = full-resolution coordinate x downsample_ratio
    sk = _fast_set(fx, "fast0")
    assert sk[px[:, 1], px[:, 0]].all()


@pytest.mark.gpu
@pytest.mark.parametrize("t", [40, 149])
def test_gpu_homography_agrees_with_skimage_orb_ransac(gtx_ctx, inp, fx, t):
    sc = inp["scene"]
    st = _stabilizer(gtx_ctx)
    st.set_ref_frame(inp["f0"], sc.boxes(0))
    st.stabilize(inp["f%d" % t], sc.boxes(t))
    H = st.get_cur_trans_matrix()
    assert H is not None
    P = _grid(HW)
    Hs, Ht = fx["orb_H_%d" % t], np.linalg.inv(sc.camera(t))
    assert np.abs(_proj(H, P) - _proj(Ht, P)).max() < 1.0
    assert np.abs(_proj(H, P) - _proj(Hs, P)).max() < 1.0


@pytest.mark.gpu
def test_gpu_gmc_agrees_with_skimage_similarity(gtx_ctx, inp, fx):
    from geotrax_amd.gmc import GMC

    g = GMC(HW, ctx=gtx_ctx)
    g.apply(inp["f0"])
    A = g.apply(inp["f1"])
    P = _grid(HW)
    assert g.valid and np.abs(_proj(A, P) - _proj(fx["orb_S_1"][:2], P)).max() < 0.25


@pytest.mark.gpu
def test_gpu_ecc_agrees_with_skimage_similarity(gtx_ctx, inp, fx):
    from geotrax_amd.gmc import make_gmc

    g = make_gmc(HW, method="ecc", ctx=gtx_ctx)
    g.apply(inp["f0"])
    A = g.apply(inp["f1"])
    P = _grid(HW)
    assert g.valid and np.abs(_ecc_full(A, P) - _proj(fx["orb_S_1"][:2], P)).max() < 0.3
    g.close()


@pytest.mark.gpu
def test_gpu_warp_agrees_with_skimage_warp(gtx_ctx, inp, fx, mi):
    from geotrax_amd.warp import warp_perspective

    out = warp_perspective(inp["warp_src"], fx["warp_H"], ctx=gtx_ctx)
    inside = _warp_interior(mi, fx, out.shape)
    d = np.abs(out.astype(np.int16) - fx["warp_out"].astype(np.int16)).max(-1)[inside]
    assert d.max() <= 2 and (d > 1).mean() < 1e-3, (d.max(), (d > 1).mean())


@pytest.mark.gpu
def test_gpu_general_letterbox_agrees_with_skimage_resize(gtx_ctx, inp, fx, mi):
    """preprocess_kernel's general (non-2x) bilinear path, the one 1080p / 2.7K sources take (det_kernels.hip)."""
    from geotrax_amd import ops

    nh, nw = mi.RESIZE_TO
    img, _ = ops.preprocess(inp["resize_src"], nh, nw, want_gray=False, ctx=gtx_ctx)
    bgr = np.rint(img[..., 2::-1] * 255.0).astype(np.int16)            # RGB0 / 255 -> BGR bytes
    d = np.abs(bgr - fx["resize_out"].astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 0.25, (d.max(), (d > 0).mean())


@pytest.mark.gpu
def test_gpu_yuv_agrees_with_skimage_ycbcr(gtx_ctx, inp, fx):
    from geotrax_amd import _lib

    yh, yw = (int(v) for v in inp["yuv_hw"])
    data = inp["i420"]
    src, dst = gtx_ctx.dev_alloc(data.nbytes), gtx_ctx.dev_alloc(yh * yw * 3)
    try:
        gtx_ctx.dev_upload(src, data)
        _lib.check(gtx_ctx.lib.gtx_yuv420_to_bgr_dev(gtx_ctx.handle, C.c_void_p(src), yh, yw, C.c_void_p(dst)))
        out = np.empty((yh, yw, 3), np.uint8)
        gtx_ctx.dev_download(out, dst)
    finally:
        gtx_ctx.dev_free(src)
        gtx_ctx.dev_free(dst)
    d = np.abs(out.astype(np.int16) - fx["yuv_bgr"].astype(np.int16))
    assert d.max() <= 1 and (d > 0).mean() < 0.1
