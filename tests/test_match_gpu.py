"""MFMA brute-force 2-NN matcher (gtx_op_match_2nn) against a numpy restatement of
cv2.BFMatcher(NORM_L2).knnMatch(k=2) on RootSIFT-like descriptors (registration.py:59-85)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rootsift_like(rng, n, clusters=None, noise=0.05):
    """Non-negative, L1-normalised then square-rooted rows (unit L2 norm), like RootSIFT."""
    base = rng.gamma(0.6, 1.0, (n, 128)).astype(np.float32) if clusters is None else clusters[rng.integers(0, len(clusters), n)]
    d = np.abs(base + noise * rng.standard_normal((n, 128)).astype(np.float32) * base.mean())
    d /= d.sum(1, keepdims=True) + 1e-8
    return np.sqrt(d).astype(np.float32)


def _knn2_numpy(q, t):
    d = np.sqrt(np.maximum((q * q).sum(1)[:, None] + (t * t).sum(1)[None] - 2.0 * (q.astype(np.float64) @ t.astype(np.float64).T), 0.0))
    order = np.argsort(d, axis=1, kind="stable")[:, :2]
    return order, np.take_along_axis(d, order, 1)


@pytest.mark.parametrize("nq,nt", [(1, 2), (37, 5), (300, 1000), (1000, 4099), (129, 128 * 40 + 3)])
def test_match_2nn_equals_bruteforce(gtx_ctx, nq, nt):
    from geotrax_amd import ops

    rng = np.random.default_rng(nq * 131 + nt)
    centres = rng.gamma(0.6, 1.0, (max(nt // 3, 1), 128)).astype(np.float32)
    t = _rootsift_like(rng, nt, centres)
    q = _rootsift_like(rng, nq, centres)
    i1, i2, d1, d2 = ops.match_2nn(q, t, ctx=gtx_ctx)
    d_all = np.sqrt(np.maximum(2.0 - 2.0 * (q.astype(np.float64) @ t.astype(np.float64).T), 0.0))
    srt = np.sort(d_all, axis=1)
    order, dist = _knn2_numpy(q, t)
    # The search runs on fp16 dot products: a winner may differ from the float64 one only where the
    # next candidate is within the fp16 noise (4e-3 in distance); distances are exact fp32 either way.
    gap1 = (srt[:, 1] - srt[:, 0]) if nt >= 2 else np.full(nq, np.inf)
    clear1 = gap1 > 4e-3
    np.testing.assert_array_equal(i1[clear1], order[clear1, 0])
    np.testing.assert_allclose(d1[clear1], dist[clear1, 0], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(d1, dist[:, 0], rtol=0, atol=4e-3)
    assert clear1.mean() > 0.5                                   # the data is not degenerate
    if nt >= 2:
        gap2 = (srt[:, 2] - srt[:, 1]) if nt >= 3 else np.full(nq, np.inf)
        clear2 = clear1 & (gap2 > 4e-3)
        np.testing.assert_array_equal(i2[clear2], order[clear2, 1])
        np.testing.assert_allclose(d2[clear2], dist[clear2, 1], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(d2, dist[:, 1], rtol=0, atol=4e-3)
        assert (d1 <= d2).all()
        # Lowe ratio decisions (ratio 0.55, default.yaml) agree wherever the margin exceeds the fp16 noise
        r_gpu, r_ref = d1 < 0.55 * d2, dist[:, 0] < 0.55 * dist[:, 1]
        clear = np.abs(dist[:, 0] - 0.55 * dist[:, 1]) > 4e-3
        np.testing.assert_array_equal(r_gpu[clear], r_ref[clear])


def test_match_2nn_degenerate_sizes(gtx_ctx):
    from geotrax_amd import ops

    rng = np.random.default_rng(0)
    q = _rootsift_like(rng, 5)
    i1, i2, d1, d2 = ops.match_2nn(q, q[:1], ctx=gtx_ctx)            # one train row: no second neighbour
    assert (i1 == 0).all() and (i2 == -1).all() and (d2 > 1e38).all()
    i1, i2, d1, d2 = ops.match_2nn(q, np.zeros((0, 128), np.float32), ctx=gtx_ctx)
    assert (i1 == -1).all() and (i2 == -1).all()
    out = ops.match_2nn(np.zeros((0, 128), np.float32), q, ctx=gtx_ctx)
    assert len(out[0]) == 0
    i1, i2, d1, d2 = ops.match_2nn(q, q, ctx=gtx_ctx)                # self match: distance 0 to itself
    np.testing.assert_array_equal(i1, np.arange(5))
    np.testing.assert_allclose(d1, 0.0, atol=1e-6)
