"""Frame warp (geotrax/visualize.py:285-289): the HIP kernel behind gtx_warp_frame /
gtx_warp_frame_dev against oracle/warp_ref.py (numpy restatement of cv2.warpPerspective: INTER_LINEAR, constant-0
border, 1/32-pixel coordinate quantisation) at 4K, through the C ABI, plus the properties that hold whatever the
resampler: identity, integer translations, composition with the box warp of the extract stage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
H4, W4 = 2160, 3840


def _frame(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    base = 110 + 60 * np.sin(xx / 41.0) * np.cos(yy / 29.0)
    f = np.stack([base + 25 * rng.standard_normal((h, w)) for _ in range(3)], -1)
    return np.clip(f, 0, 255).astype(np.uint8)


def _camera(seed, w, h, strength=1.0):
    rng = np.random.default_rng(seed)
    a = 2e-3 * strength * rng.standard_normal()
    return np.array([[np.cos(a) * (1 + 1e-3 * strength), -np.sin(a), 6.0 * strength * rng.standard_normal()],
                     [np.sin(a), np.cos(a) * (1 - 1e-3 * strength), 6.0 * strength * rng.standard_normal()],
                     [1e-7 * strength * rng.standard_normal(), 1e-7 * strength * rng.standard_normal(), 1.0]])


@pytest.mark.parametrize("case", ["golden-envelope", "strong", "zoom-out-rotate"])
def test_warp_4k_matches_oracle(gtx_ctx, case):
    from geotrax_amd.warp import warp_perspective
    from oracle.warp_ref import warp_perspective as ref

    rng = np.random.default_rng(3)
    f = _frame(rng, H4, W4)
    if case == "golden-envelope":          # the size of the reference's golden homographies: few px drift, 1e-3 rotation, 1e-7 perspective
        Hm = _camera(1, W4, H4)
    elif case == "strong":
        Hm = _camera(2, W4, H4, strength=8.0)
    else:                                  # 0.45x zoom + 30 degree rotation about the frame centre: tile footprints exceed the LDS
        c, s, z = np.cos(0.52), np.sin(0.52), 0.45   # budget, so the direct-gather path runs; large constant-0 border area
        T = np.array([[1, 0, W4 / 2], [0, 1, H4 / 2], [0, 0, 1.0]])
        Hm = T @ np.array([[z * c, -z * s, 0], [z * s, z * c, 0], [2e-6, -1e-6, 1.0]]) @ np.linalg.inv(T)
    got = warp_perspective(f, Hm, ctx=gtx_ctx)
    want = ref(f, Hm)
    diff = got.astype(np.int16) - want.astype(np.int16)
    # the two inverses of H (library: adjugate; oracle: LAPACK) agree to ~1e-16: a handful of coordinates in 16.6 M may fall on the other side of a 1/32 px rounding boundary
    assert np.count_nonzero(diff) <= 32 and np.abs(diff).max() <= 3, (np.count_nonzero(diff), np.abs(diff).max())
    assert got.std() > 10


def test_warp_properties_and_ragged_sizes(gtx_ctx):
    from geotrax_amd.warp import FrameWarper, warp_perspective
    from oracle.warp_ref import warp_perspective as ref

    rng = np.random.default_rng(5)
    for (h, w) in ((2160, 3840), (333, 517), (64, 130), (9, 5)):   # 517*3 and 5*3 are not multiples of 4 or 16: unaligned rows
        f = _frame(rng, h, w)
        np.testing.assert_array_equal(warp_perspective(f, np.eye(3), ctx=gtx_ctx), f)                      # identity
        tx, ty = 7, -3
        sh = warp_perspective(f, np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1.0]]), ctx=gtx_ctx)              # integer shift: exact copy
        want = np.zeros_like(f)
        if w > tx and h > -ty:
            want[:h + ty, tx:] = f[-ty:, :w - tx]
        np.testing.assert_array_equal(sh, want)
        Hm = _camera(h + w, w, h, strength=3.0)
        np.testing.assert_array_equal(warp_perspective(f, Hm, ctx=gtx_ctx), ref(f, Hm))
    # the streaming object (device buffers kept) gives the same frames; no transform -> frame returned unchanged
    f = _frame(rng, 540, 960)
    wp = FrameWarper((540, 960), ctx=gtx_ctx)
    try:
        for s in range(3):
            Hm = _camera(s, 960, 540, strength=2.0)
            np.testing.assert_array_equal(wp(f, Hm), warp_perspective(f, Hm, ctx=gtx_ctx))
        assert wp(f, None) is f
        with pytest.raises(ValueError):
            wp(f[:100], np.eye(3))
    finally:
        wp.close()
    with pytest.raises(Exception):
        warp_perspective(f, np.zeros((3, 3)), ctx=gtx_ctx)                                                 # singular


def test_warped_frame_agrees_with_the_stabilized_boxes(gtx_ctx):
    """The extract stage maps boxes through H (box corners -> axis-aligned hull, K10); the visualisation warps the
    frame through the same H. A bright rectangle drawn at a box must land on the warped box."""
    from geotrax_amd.geometry import warp_boxes
    from geotrax_amd.warp import warp_perspective

    h, w = 1080, 1920
    f = np.full((h, w, 3), 20, np.uint8)
    boxes = np.array([[400, 300, 120, 60], [1500, 800, 90, 44], [960, 540, 200, 100]], np.float32)    # xywh
    for cx, cy, bw, bh in boxes:
        f[int(cy - bh / 2):int(cy + bh / 2), int(cx - bw / 2):int(cx + bw / 2)] = 240
    Hm = _camera(9, w, h, strength=6.0)
    out = warp_perspective(f, Hm, ctx=gtx_ctx)
    for (cx, cy, bw, bh) in warp_boxes(Hm, boxes):
        ys, xs = np.nonzero(out[int(cy - bh / 2) - 6:int(cy + bh / 2) + 6, int(cx - bw / 2) - 6:int(cx + bw / 2) + 6, 0] > 128)
        assert len(xs) > 0.8 * bw * bh
        assert abs((xs.min() + xs.max()) / 2 + int(cx - bw / 2) - 6 - cx) < 1.5 and abs((ys.min() + ys.max()) / 2 + int(cy - bh / 2) - 6 - cy) < 1.5
