"""CPU restatement of BoT-SORT's global motion compensation, method 'ecc' -- TEST INFRASTRUCTURE ONLY
(imported by tests/; never by the product path).

Path restated: the tracker callback of `model.track(..., persist=True)` (geotrax/extract.py:153) with
cfg -> tracker -> {botsort, deepocsort, tracktrack} -> gmc_method: ecc (geotrax/cfg/default.yaml:374,419,467), i.e.
ultralytics.trackers.utils.gmc.GMC(method='ecc', downscale=2).apply_ecc:

    gray = cvtColor(frame, BGR2GRAY); gray = GaussianBlur(gray, (3, 3), 1.5); gray = resize(gray, (w // 2, h // 2))
    first frame: prevFrame = gray, return eye(2, 3)
    else: findTransformECC(prevFrame, gray, H = eye(2, 3, float32), MOTION_EUCLIDEAN, (EPS | COUNT, 5000, 1e-6), None, 1)

Two properties of that code are kept, because a drop-in reproduces what the reference computes: `prevFrame` is never
replaced (every frame is registered against the FIRST frame of the sequence, and that warp is what the tracker applies as
"previous -> current"), and the translation is NOT scaled back by the downscale factor (it is in half-resolution pixels).
Both come from the BoT-SORT authors' gmc.py, which ultralytics carries unchanged. When findTransformECC raises (a NaN
correlation, or the correlation about to be minimised) upstream logs a warning and returns the matrix as the failed call
left it (the array is updated in place): so does this.

ultralytics and OpenCV are absent from /root/reference (pyproject pins ultralytics>=8.4.80; OpenCV comes with it, version
open), so this follows OpenCV's published source -- video/src/ecc.cpp (Evangelidis &
Psarakis, PAMI 2008: forward additive ECC) and imgproc's warpAffine -- operation by operation and is **PARITY UNPINNED**
against OpenCV itself (held against an estimator nobody here wrote: scikit-image's ORB + ransac(SimilarityTransform) on a
consecutive frame pair and the clip's own camera, <= 0.3 px apart on a 9 x 16 grid, tests/test_independent.py):
  * cvtColor BGR2GRAY: (1868 B + 9617 G + 4899 R + 8192) >> 14;
  * GaussianBlur 3x3, sigma 1.5 on uint8: the bit-exact fixed-point kernel (79, 98, 79) / 256, rows then columns,
    BORDER_REFLECT_101, rounded once at the end ((v + 2^15) >> 16);
  * resize to exactly half: INTER_LINEAR becomes the 2 x 2 area mean, (a + b + c + d + 2) >> 2;
  * ECC: float32 images, gradients by filter2D with (-0.5, 0, 0.5) (BORDER_REFLECT_101), warpAffine(INTER_LINEAR |
    WARP_INVERSE_MAP, constant border 0) in one of the two forms below, the validity mask by warpAffine(INTER_NEAREST) of ones, mean / std / dot
    products accumulated in float64, the 3 x 3 Hessian and the projections stored as float32, its inverse by cofactors in
    float64, the Euclidean update theta += dp0, t += (dp1, dp2).
Two forms of warpAffine's bilinear path exist upstream and both are restated (`warp=`): "exact" -- OpenCV >= 4.11, whose new
kernels take the source position m0 x + m1 y + m2 as it is and interpolate p00 + a (p01 - p00) ..., the build an installation
of the pinned ultralytics resolves to today -- and "fixed" -- OpenCV through 4.10: the position in fixed point, rounded to
1/32 pixel, table weights. The difference matters for more than the last digits: under "fixed" the samples stop changing
smoothly near the optimum, the coefficient dithers in its sixth decimal and many fits never meet the 1e-6 criterion (they run
to the 5000-iteration cap); under "exact" the same fits end after 5-30 iterations. The mask (INTER_NEAREST) is fixed point in both.
"""
from __future__ import annotations

import numpy as np

MAX_ITERS = 5000
EPS = 1e-6
AB_BITS, INTER_BITS = 10, 5
AB_SCALE = 1 << AB_BITS
INTER_TAB = 1 << INTER_BITS


def _reflect101(i, n):
    p = 2 * (n - 1)
    i = np.abs(i) % p
    return np.where(i >= n, p - i, i)


def gray_bgr(frame: np.ndarray) -> np.ndarray:
    f = np.asarray(frame, np.uint8).astype(np.int32)
    return ((f[..., 0] * 1868 + f[..., 1] * 9617 + f[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


def gaussian_blur3(gray: np.ndarray) -> np.ndarray:
    """cv2.GaussianBlur(gray, (3, 3), 1.5) on uint8: fixed-point kernel (79, 98, 79) / 256 in both directions."""
    g = gray.astype(np.int64)
    h, w = g.shape
    xi = [_reflect101(np.arange(w) + d, w) for d in (-1, 0, 1)]
    hz = 79 * g[:, xi[0]] + 98 * g[:, xi[1]] + 79 * g[:, xi[2]]
    yi = [_reflect101(np.arange(h) + d, h) for d in (-1, 0, 1)]
    v = 79 * hz[yi[0]] + 98 * hz[yi[1]] + 79 * hz[yi[2]]
    return np.clip((v + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


def half_size(gray: np.ndarray) -> np.ndarray:
    """cv2.resize(gray, (w // 2, h // 2)) for an exact factor of two (odd sizes: the last row / column is dropped)."""
    h2, w2 = gray.shape[0] // 2, gray.shape[1] // 2
    g = gray[:2 * h2, :2 * w2].astype(np.int32)
    return ((g[0::2, 0::2] + g[0::2, 1::2] + g[1::2, 0::2] + g[1::2, 1::2] + 2) >> 2).astype(np.uint8)


def prepare(frame_bgr: np.ndarray) -> np.ndarray:
    """The image apply_ecc hands to findTransformECC (uint8, half resolution)."""
    return half_size(gaussian_blur3(gray_bgr(frame_bgr)))


def gradients(img: np.ndarray):
    """filter2D(img, -1, (-0.5, 0, 0.5)) and its transpose, float32, BORDER_REFLECT_101."""
    h, w = img.shape
    xm, xp = _reflect101(np.arange(w) - 1, w), _reflect101(np.arange(w) + 1, w)
    ym, yp = _reflect101(np.arange(h) - 1, h), _reflect101(np.arange(h) + 1, h)
    half, zero = np.float32(0.5), np.float32(0.0)
    gx = (-half * img[:, xm] + zero * img) + half * img[:, xp]
    gy = (-half * img[ym] + zero * img) + half * img[yp]
    return gx.astype(np.float32), gy.astype(np.float32)


def _sat_int(v):
    """saturate_cast<int>(double): round half to even (cvRound)."""
    return np.rint(v).astype(np.int64)


def warp_coords(M: np.ndarray, hs: int, ws: int):
    """warpAffine's source coordinates for WARP_INVERSE_MAP: integer pixel and the 1/32 fractions (INTER_LINEAR), and the
    nearest pixel (INTER_NEAREST). M: 2x3 float32 (used as float64, as warpAffine converts it)."""
    m = M.astype(np.float64)
    x = np.arange(ws, dtype=np.float64)
    y = np.arange(hs, dtype=np.float64)
    ad, bd = _sat_int(m[0, 0] * x * AB_SCALE), _sat_int(m[1, 0] * x * AB_SCALE)
    out = []
    for rd in (AB_SCALE // INTER_TAB // 2, AB_SCALE // 2):
        X0 = _sat_int((m[0, 1] * y + m[0, 2]) * AB_SCALE) + rd
        Y0 = _sat_int((m[1, 1] * y + m[1, 2]) * AB_SCALE) + rd
        out.append((X0[:, None] + ad[None, :], Y0[:, None] + bd[None, :]))
    (Xl, Yl), (Xn, Yn) = out
    Xl >>= AB_BITS - INTER_BITS
    Yl >>= AB_BITS - INTER_BITS
    return (Xl >> INTER_BITS, Yl >> INTER_BITS, Xl & (INTER_TAB - 1), Yl & (INTER_TAB - 1)), (Xn >> AB_BITS, Yn >> AB_BITS)


def _fetch(img, sy, sx):
    h, w = img.shape
    ok = (sy >= 0) & (sy < h) & (sx >= 0) & (sx < w)
    v = img[np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1)]
    return np.where(ok, v, np.float32(0.0)).astype(np.float32)


def warp_linear(img: np.ndarray, lin) -> np.ndarray:
    """remapBilinear on float32: table weights (1 - fy)(1 - fx), (1 - fy) fx, fy (1 - fx), fy fx as float32 products."""
    sx, sy, fx, fy = lin
    ax = (fx.astype(np.float32) * np.float32(1.0 / INTER_TAB)).astype(np.float32)
    ay = (fy.astype(np.float32) * np.float32(1.0 / INTER_TAB)).astype(np.float32)
    one = np.float32(1.0)
    w00, w01 = (one - ay) * (one - ax), (one - ay) * ax
    w10, w11 = ay * (one - ax), ay * ax
    return ((_fetch(img, sy, sx) * w00 + _fetch(img, sy, sx + 1) * w01) + _fetch(img, sy + 1, sx) * w10) + _fetch(img, sy + 1, sx + 1) * w11


def warp_coords_exact(M: np.ndarray, hs: int, ws: int):
    """OpenCV >= 4.11: source position in floating point, floor + fraction (float32 fraction)."""
    m = M.astype(np.float64)
    x = np.arange(ws, dtype=np.float64)[None, :]
    y = np.arange(hs, dtype=np.float64)[:, None]
    sx = (m[0, 0] * x + m[0, 1] * y) + m[0, 2]
    sy = (m[1, 0] * x + m[1, 1] * y) + m[1, 2]
    ix, iy = np.floor(sx), np.floor(sy)
    return ix.astype(np.int64), iy.astype(np.int64), (sx - ix).astype(np.float32), (sy - iy).astype(np.float32)


def warp_linear_exact(img: np.ndarray, ex) -> np.ndarray:
    sx, sy, ax, ay = ex
    p00, p01, p10, p11 = _fetch(img, sy, sx), _fetch(img, sy, sx + 1), _fetch(img, sy + 1, sx), _fetch(img, sy + 1, sx + 1)
    v0 = p00 + ax * (p01 - p00)
    v1 = p10 + ax * (p11 - p10)
    return (v0 + ay * (v1 - v0)).astype(np.float32)


def _masked_mean_std(a, mask, n):
    s = a[mask].astype(np.float64)
    mean = s.sum() / n
    var = max((s * s).sum() / n - mean * mean, 0.0)
    return mean, np.sqrt(var)


def _inv3(hm: np.ndarray):
    """cv::invert(DECOMP_LU) of a 3x3 float32 matrix: cofactors in float64, float32 result; None when singular."""
    a = hm.astype(np.float64)
    d = (a[0, 0] * (a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]) - a[0, 1] * (a[1, 0] * a[2, 2] - a[1, 2] * a[2, 0])
         + a[0, 2] * (a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]))
    if d == 0.0:
        return None
    d = 1.0 / d
    t = np.empty((3, 3))
    t[0, 0] = (a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]) * d
    t[0, 1] = (a[0, 2] * a[2, 1] - a[0, 1] * a[2, 2]) * d
    t[0, 2] = (a[0, 1] * a[1, 2] - a[0, 2] * a[1, 1]) * d
    t[1, 0] = (a[1, 2] * a[2, 0] - a[1, 0] * a[2, 2]) * d
    t[1, 1] = (a[0, 0] * a[2, 2] - a[0, 2] * a[2, 0]) * d
    t[1, 2] = (a[0, 2] * a[1, 0] - a[0, 0] * a[1, 2]) * d
    t[2, 0] = (a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]) * d
    t[2, 1] = (a[0, 1] * a[2, 0] - a[0, 0] * a[2, 1]) * d
    t[2, 2] = (a[0, 0] * a[1, 1] - a[0, 1] * a[1, 0]) * d
    return t.astype(np.float32)


def find_transform_ecc(template_u8: np.ndarray, image_u8: np.ndarray, M: np.ndarray, max_iters: int = MAX_ITERS, eps: float = EPS, warp: str = "exact"):
    """cv2.findTransformECC(template, image, M, MOTION_EUCLIDEAN, (EPS | COUNT, max_iters, eps), None, 1). M (2x3 float32) is
    updated in place. Returns (rho, iterations run, status): status 0 = finished, 1 = NaN correlation, 2 = the correlation
    would be minimised (both are cv2.error upstream)."""
    tmpl = template_u8.astype(np.float32)
    img = image_u8.astype(np.float32)
    hs, ws = tmpl.shape
    gx, gy = gradients(img)
    X = np.broadcast_to(np.arange(ws, dtype=np.float32)[None, :], (hs, ws))
    Y = np.broadcast_to(np.arange(hs, dtype=np.float32)[:, None], (hs, ws))
    ones = np.ones(img.shape, np.float32)
    rho, last_rho = -1.0, -eps
    it = 0
    while it < max_iters and abs(rho - last_rho) >= eps:
        it += 1
        lin, (nx, ny) = warp_coords(M, hs, ws)
        if warp == "exact":
            ex = warp_coords_exact(M, hs, ws)
            iw, gxw, gyw = warp_linear_exact(img, ex), warp_linear_exact(gx, ex), warp_linear_exact(gy, ex)
        else:
            iw, gxw, gyw = warp_linear(img, lin), warp_linear(gx, lin), warp_linear(gy, lin)
        mask = _fetch(ones, ny, nx) > 0
        n = int(mask.sum())
        img_mean, img_std = _masked_mean_std(iw, mask, n) if n else (0.0, 0.0)
        tmp_mean, tmp_std = _masked_mean_std(tmpl, mask, n) if n else (0.0, 0.0)
        iw = np.where(mask, iw - np.float32(img_mean), iw).astype(np.float32)
        tz = np.where(mask, tmpl - np.float32(tmp_mean), np.float32(0.0)).astype(np.float32)
        tmp_norm = np.sqrt(n * tmp_std * tmp_std)
        img_norm = np.sqrt(n * img_std * img_std)
        h0, h1 = M[0, 0], M[1, 0]                                         # cos(theta), sin(theta), float32
        hat_x = -(X * h1) - (Y * h0)
        hat_y = (X * h0) - (Y * h1)
        J = [(gxw * hat_x) + (gyw * hat_y), gxw, gyw]                     # float32
        Jd = [j.astype(np.float64) for j in J]
        hess = np.array([[np.sum(Jd[i] * Jd[j]) for j in range(3)] for i in range(3)]).astype(np.float32)
        hinv = _inv3(hess)
        iwd, tzd = iw.astype(np.float64), tz.astype(np.float64)
        correlation = float(np.sum(tzd * iwd))
        last_rho = rho
        with np.errstate(all="ignore"):
            rho = correlation / (img_norm * tmp_norm) if img_norm * tmp_norm != 0 else float("nan")
        if np.isnan(rho):
            return rho, it, 1
        if hinv is None:                                                  # a singular Hessian inverts to zeros in OpenCV: no update, rho repeats
            hinv = np.zeros((3, 3), np.float32)
        img_proj = np.array([np.sum(Jd[i] * iwd) for i in range(3)]).astype(np.float32)
        tmp_proj = np.array([np.sum(Jd[i] * tzd) for i in range(3)]).astype(np.float32)
        iph = (hinv.astype(np.float64) @ img_proj.astype(np.float64)).astype(np.float32)
        lambda_n = img_norm * img_norm - float(np.dot(img_proj.astype(np.float64), iph.astype(np.float64)))
        lambda_d = correlation - float(np.dot(tmp_proj.astype(np.float64), iph.astype(np.float64)))
        if lambda_d <= 0.0:
            return -1.0, it, 2
        lam = lambda_n / lambda_d
        # error = lambda * templateZM - imageWarped projected onto the Jacobian; the projection is linear in its argument. The
        # float32 image `error` upstream forms first costs one rounding per pixel, which the float64 sums of 10^5+ terms absorb.
        err_proj = (lam * tmp_proj.astype(np.float64) - img_proj.astype(np.float64)).astype(np.float32)
        dp = (hinv.astype(np.float64) @ err_proj.astype(np.float64)).astype(np.float32)
        theta = np.arcsin(np.float64(M[1, 0])) + np.float64(dp[0])
        M[0, 2] = np.float32(np.float64(M[0, 2]) + np.float64(dp[1]))
        M[1, 2] = np.float32(np.float64(M[1, 2]) + np.float64(dp[2]))
        M[0, 0] = M[1, 1] = np.float32(np.cos(theta))
        M[1, 0] = np.float32(np.sin(theta))
        M[0, 1] = -M[1, 0]
    return rho, it, 0


class EccRef:
    """GMC(method='ecc', downscale=2).apply(frame_bgr) -> 2x3 float64 (see the module text for the two upstream properties kept)."""

    def __init__(self, max_iters: int = MAX_ITERS, eps: float = EPS, replace_template: bool = False, warp: str = "exact"):
        assert warp in ("exact", "fixed")
        self.warp = warp
        self.template = None
        self.max_iters, self.eps = max_iters, eps
        self.replace_template = replace_template         # True: NOT upstream -- every frame becomes the template of the next one
        self.last = {}

    def apply(self, frame_bgr: np.ndarray) -> np.ndarray:
        g = prepare(frame_bgr)
        H = np.eye(2, 3, dtype=np.float32)
        if self.template is None:
            self.template = g
            return H.astype(np.float64)
        rho, iters, status = find_transform_ecc(self.template, g, H, self.max_iters, self.eps, self.warp)
        self.last = dict(rho=rho, iters=iters, status=status)
        if self.replace_template:
            self.template = g
        return H.astype(np.float64)
