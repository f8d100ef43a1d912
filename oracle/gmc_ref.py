"""CPU restatement of BoT-SORT's global motion compensation, method 'sparseOptFlow' -- TEST
INFRASTRUCTURE ONLY (imported by tests/ and bench.py's cpu_baseline; never by the product path).

Path restated: the tracker callback of `model.track(..., persist=True)` (geotrax/extract.py:153) with
cfg -> tracker -> botsort -> gmc_method: sparseOptFlow (geotrax/cfg/default.yaml:362-374), i.e.
ultralytics.trackers.utils.gmc.GMC.apply_sparseoptflow: gray, 1/2 scale, cv2.goodFeaturesToTrack
(maxCorners 1000, qualityLevel 0.01, minDistance 1, blockSize 3), cv2.calcOpticalFlowPyrLK against the
previous frame (21x21 window, 3 pyramid levels above the base, 30 iterations / 0.01), then
cv2.estimateAffinePartial2D(RANSAC); the 2x3 result (translation scaled back by 2) is applied to every
track's Kalman state. ultralytics and OpenCV are absent from /root/reference (pyproject pins
ultralytics>=8.4.80), so this follows the published algorithms (Shi-Tomasi, Bouguet's pyramidal LK) and
is **PARITY UNPINNED** against OpenCV itself (held against scikit-image's ORB + ransac(SimilarityTransform) on a
consecutive frame pair: <= 0.25 px apart on a 9 x 16 grid, tests/test_independent.py). Deliberate choices: corner response from exact integer
Sobel/box sums, LK in float64 with Scharr derivatives (OpenCV: 14-bit fixed point), similarity fit =
2-point hypotheses scored by inlier count + least squares on the inliers (OpenCV: + LM refinement).
"""
from __future__ import annotations

import numpy as np

MAX_CORNERS = 1000
QUALITY = 0.01
WIN = 21
HALF = (WIN - 1) * 0.5
MAX_LEVEL = 3
MAX_ITERS = 30
EPS = 0.01
MIN_EIG_THR = 1e-4
RANSAC_THR = 3.0
N_HYP = 512


def _reflect101(i, n):
    p = 2 * (n - 1)
    i = np.abs(i) % p
    return np.where(i >= n, p - i, i)


def _shift(a, dy, dx):
    h, w = a.shape
    return a[np.ix_(_reflect101(np.arange(h) + dy, h), _reflect101(np.arange(w) + dx, w))]


def corner_response(gray: np.ndarray) -> np.ndarray:
    """Minimum eigenvalue of the 3x3-summed Sobel structure tensor (cornerMinEigenVal, blockSize 3,
    ksize 3) up to a constant factor; integer sums, float64 eigenvalue."""
    g = gray.astype(np.int64)
    dx = (_shift(g, -1, 1) - _shift(g, -1, -1)) + 2 * (_shift(g, 0, 1) - _shift(g, 0, -1)) + (_shift(g, 1, 1) - _shift(g, 1, -1))
    dy = (_shift(g, 1, -1) - _shift(g, -1, -1)) + 2 * (_shift(g, 1, 0) - _shift(g, -1, 0)) + (_shift(g, 1, 1) - _shift(g, -1, 1))

    def box(a):
        return sum(_shift(a, i, j) for i in (-1, 0, 1) for j in (-1, 0, 1))

    a, b, c = box(dx * dx), box(dx * dy), box(dy * dy)
    return 0.5 * (a + c).astype(np.float64) - np.sqrt(0.25 * ((a - c).astype(np.float64) ** 2) + b.astype(np.float64) ** 2)


def good_features(gray: np.ndarray) -> np.ndarray:
    """[n,2] float32 (x, y): 3x3 local maxima above 0.01 * max, strongest first (ties: larger pixel
    index first), at most 1000, the 1-pixel border excluded."""
    lam = corner_response(gray)
    h, w = lam.shape
    thr = lam.max() * QUALITY
    v = np.where(lam > thr, lam, 0.0)
    dil = np.full((h, w), -np.inf)
    pad = np.pad(v, 1, mode="constant", constant_values=-np.inf)
    for i in range(3):
        for j in range(3):
            dil = np.maximum(dil, pad[i:i + h, j:j + w])
    ok = (v != 0) & (v == dil)
    ok[0, :] = ok[-1, :] = False
    ok[:, 0] = ok[:, -1] = False
    ys, xs = np.nonzero(ok)
    pix = ys * w + xs
    order = np.lexsort((-pix, -v[ys, xs]))[:MAX_CORNERS]
    return np.stack([xs[order], ys[order]], 1).astype(np.float32)


def pyr_down(img: np.ndarray) -> np.ndarray:
    """cv2.pyrDown on u8: separable [1 4 6 4 1], reflect-101, every second pixel, (sum + 128) >> 8."""
    a = img.astype(np.int64)
    h, w = a.shape
    oh, ow = (h + 1) // 2, (w + 1) // 2
    k = (1, 4, 6, 4, 1)
    rows = sum(k[i] * a[:, _reflect101(2 * np.arange(ow) + i - 2, w)] for i in range(5))
    out = sum(k[i] * rows[_reflect101(2 * np.arange(oh) + i - 2, h), :] for i in range(5))
    return ((out + 128) >> 8).astype(np.uint8)


def build_pyramid(gray: np.ndarray) -> list[np.ndarray]:
    pyr = [gray]
    for _ in range(MAX_LEVEL):
        pyr.append(pyr_down(pyr[-1]))
    return pyr


def _scharr(img: np.ndarray):
    g = img.astype(np.float64)
    ix = 3 * (_shift(g, -1, 1) - _shift(g, -1, -1)) + 10 * (_shift(g, 0, 1) - _shift(g, 0, -1)) + 3 * (_shift(g, 1, 1) - _shift(g, 1, -1))
    iy = 3 * (_shift(g, 1, -1) - _shift(g, -1, -1)) + 10 * (_shift(g, 1, 0) - _shift(g, -1, 0)) + 3 * (_shift(g, 1, 1) - _shift(g, -1, 1))
    return ix / 32.0, iy / 32.0


def _bilinear(img, x0, y0, fx, fy):
    """WIN x WIN patch whose top-left sample sits at (x0 + fx, y0 + fy); caller guarantees bounds."""
    p = img[y0:y0 + WIN + 1, x0:x0 + WIN + 1]
    return (p[:-1, :-1] * (1 - fx) * (1 - fy) + p[:-1, 1:] * fx * (1 - fy) + p[1:, :-1] * (1 - fx) * fy + p[1:, 1:] * fx * fy)


def lk_track(prev_pyr, cur_pyr, pts: np.ndarray):
    """Pyramidal Lucas-Kanade, one point at a time. -> (next points [n,2] f32, status [n] bool)."""
    grads = [_scharr(p) for p in prev_pyr]
    prevf = [p.astype(np.float64) for p in prev_pyr]
    curf = [p.astype(np.float64) for p in cur_pyr]
    out = np.zeros((len(pts), 2), np.float64)
    status = np.ones(len(pts), bool)
    for n, (px, py) in enumerate(pts.astype(np.float64)):
        nx = ny = 0.0
        ok = True
        for L in range(MAX_LEVEL, -1, -1):
            sc = 1.0 / (1 << L)
            qx, qy = px * sc, py * sc
            if L == MAX_LEVEL:
                nx, ny = qx, qy
            else:
                nx, ny = nx * 2.0, ny * 2.0
            I, (Ix, Iy), J = prevf[L], grads[L], curf[L]
            h, w = I.shape
            tx, ty = qx - HALF, qy - HALF
            x0, y0 = int(np.floor(tx)), int(np.floor(ty))
            if x0 < 0 or y0 < 0 or x0 + WIN + 1 > w or y0 + WIN + 1 > h:
                if L == 0:
                    ok = False
                continue
            fx, fy = tx - x0, ty - y0
            Iw, Ixw, Iyw = _bilinear(I, x0, y0, fx, fy), _bilinear(Ix, x0, y0, fx, fy), _bilinear(Iy, x0, y0, fx, fy)
            A11, A12, A22 = float((Ixw * Ixw).sum()), float((Ixw * Iyw).sum()), float((Iyw * Iyw).sum())
            D = A11 * A22 - A12 * A12
            min_eig = (A22 + A11 - np.sqrt((A11 - A22) ** 2 + 4.0 * A12 * A12)) / (2.0 * WIN * WIN)
            if min_eig < MIN_EIG_THR or D < 1.1920929e-7:
                if L == 0:
                    ok = False
                continue
            D = 1.0 / D
            pdx = pdy = 0.0
            for j in range(MAX_ITERS):
                ux, uy = nx - HALF, ny - HALF
                jx0, jy0 = int(np.floor(ux)), int(np.floor(uy))
                if jx0 < 0 or jy0 < 0 or jx0 + WIN + 1 > w or jy0 + WIN + 1 > h:
                    if L == 0:
                        ok = False
                    break
                Jw = _bilinear(J, jx0, jy0, ux - jx0, uy - jy0)
                diff = Jw - Iw
                b1, b2 = float((diff * Ixw).sum()), float((diff * Iyw).sum())
                dx, dy = (A12 * b2 - A22 * b1) * D, (A12 * b1 - A11 * b2) * D
                nx, ny = nx + dx, ny + dy
                if dx * dx + dy * dy <= EPS * EPS:
                    break
                if j > 0 and abs(dx + pdx) < 0.01 and abs(dy + pdy) < 0.01:
                    nx, ny = nx - dx * 0.5, ny - dy * 0.5
                    break
                pdx, pdy = dx, dy
        out[n] = (nx, ny)
        status[n] = ok
    return out.astype(np.float32), status


def _hash(x: int) -> int:
    x &= 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def _similarity_from(p, q):
    """Least-squares [a -b tx; b a ty] mapping p -> q (float64, closed form)."""
    mp, mq = p.mean(0), q.mean(0)
    pc, qc = p - mp, q - mq
    den = (pc * pc).sum()
    if den <= 1e-12:
        return None
    a = (pc[:, 0] * qc[:, 0] + pc[:, 1] * qc[:, 1]).sum() / den
    b = (pc[:, 0] * qc[:, 1] - pc[:, 1] * qc[:, 0]).sum() / den
    return np.array([[a, -b, mq[0] - (a * mp[0] - b * mp[1])], [b, a, mq[1] - (b * mp[0] + a * mp[1])]])


def estimate_affine_partial(p: np.ndarray, q: np.ndarray, seed: int = 0):
    """RANSAC similarity p -> q: 512 two-point hypotheses (counter-hash sampling), most inliers within
    3 px wins (first on ties), then three rounds of least squares on the inliers. -> 2x3 f64 or None."""
    n = len(p)
    if n < 2:
        return None
    p, q = p.astype(np.float64), q.astype(np.float64)
    best, best_M = -1, None
    for hyp in range(N_HYP):
        i = _hash(seed ^ _hash(2 * hyp)) % n
        j = _hash(seed ^ _hash(2 * hyp + 1)) % n
        if i == j:
            continue
        dpx, dpy = p[j, 0] - p[i, 0], p[j, 1] - p[i, 1]
        den = dpx * dpx + dpy * dpy
        if den < 1e-12:
            continue
        dqx, dqy = q[j, 0] - q[i, 0], q[j, 1] - q[i, 1]
        a, b = (dpx * dqx + dpy * dqy) / den, (dpx * dqy - dpy * dqx) / den
        tx, ty = q[i, 0] - (a * p[i, 0] - b * p[i, 1]), q[i, 1] - (b * p[i, 0] + a * p[i, 1])
        ex = a * p[:, 0] - b * p[:, 1] + tx - q[:, 0]
        ey = b * p[:, 0] + a * p[:, 1] + ty - q[:, 1]
        cnt = int((ex * ex + ey * ey < RANSAC_THR * RANSAC_THR).sum())
        if cnt > best:
            best, best_M = cnt, np.array([[a, -b, tx], [b, a, ty]])
    if best_M is None:
        return None
    M = best_M
    for _ in range(3):
        e = p @ M[:, :2].T + M[:, 2] - q
        inl = (e * e).sum(1) < RANSAC_THR * RANSAC_THR
        if inl.sum() < 2:
            break
        M2 = _similarity_from(p[inl], q[inl])
        if M2 is None:
            break
        M = M2
    return M


class GmcRef:
    """GMC(method='sparseOptFlow', downscale=2).apply(gray_half) -> 2x3 float64."""

    def __init__(self, seed: int = 0):
        self.prev_pyr = None
        self.prev_pts = None
        self.seed = seed
        self.last = {}

    def apply(self, gray_half: np.ndarray) -> np.ndarray:
        H = np.eye(2, 3)
        pts = good_features(gray_half)
        pyr = build_pyramid(gray_half)
        if self.prev_pyr is not None and self.prev_pts is not None and len(self.prev_pts):
            nxt, st = lk_track(self.prev_pyr, pyr, self.prev_pts)
            p, q = self.prev_pts[st], nxt[st]
            self.last = dict(prev=self.prev_pts, next=nxt, status=st)
            if len(p) > 4:
                M = estimate_affine_partial(p, q, self.seed)
                if M is not None:
                    H = M.copy()
                    H[:, 2] *= 2.0
        self.prev_pyr, self.prev_pts = pyr, pts
        return H


# --------------------------------------------------------------------------- feature-based methods (gmc_method: orb / sift)
def filter_matches(prev_xy, cur_xy, frame_hw):
    """ultralytics GMC.apply_features after the ratio test: |displacement| < 0.25 x (width, height) per axis, then
    (displacement - mean) < 2.5 x std per axis -- one-sided, as upstream writes it."""
    d = np.asarray(prev_xy, np.float64) - np.asarray(cur_xy, np.float64)
    keep = (np.abs(d[:, 0]) < 0.25 * frame_hw[1]) & (np.abs(d[:, 1]) < 0.25 * frame_hw[0])
    idx = np.flatnonzero(keep)
    if len(idx) == 0:
        return keep
    dk = d[idx]
    inl = ((dk - dk.mean(0)) < 2.5 * dk.std(0)).all(1)
    out = np.zeros(len(d), bool)
    out[idx[inl]] = True
    return out


class GmcFeatureRef:
    """GMC(method='orb' | 'sift', downscale=2).apply(frame) -> 2x3 float64 (default.yaml:374 `gmc_method: orb` / `sift`), restated
    with this repository's own restatements of the detectors: oracle/stabilo_ref.py's ORB (FAST 20, 8 levels, Harris ranking,
    rotated BRIEF; 1000 keypoints) with its Hamming 2-NN matcher, or oracle/sift_ref.py's SIFT (plain descriptors) with an exact
    L2 2-NN; Lowe's ratio 0.9, the two spatial filters of apply_features, estimate_affine_partial above. CHOICES (also in the
    product, geotrax_amd/gmc.py FeatureGMC): the matcher's query is the current frame, no detection-box mask, SIFT with OpenCV's
    default thresholds. PARITY UNPINNED against ultralytics / OpenCV."""

    def __init__(self, frame_hw, method="orb", seed=0, max_features=1000):
        self.hw, self.method, self.seed, self.max_features = tuple(frame_hw), method, seed, max_features
        self.prev = None
        self.last = {}

    def _features(self, gray_half):
        if self.method == "orb":
            from . import stabilo_ref as S

            cfg = dict(downsample_ratio=0.5, max_features=self.max_features, ref_multiplier=1.0, filter_ratio=0.9, ransac_threshold=2.0,
                       mask_use=False, mask_margin_ratio=0.15, fast_threshold=20, n_levels=8, scale_factor=1.2, seed=self.seed)
            if not hasattr(self, "_pattern"):
                self._pattern = S.brief_pattern()
            f = S.extract(gray_half, None, cfg, self.max_features, self._pattern)
            return f["xy"], f["desc"]
        from . import sift_ref

        k = sift_ref.detect_and_compute(np.repeat(gray_half[:, :, None], 3, 2), max_features=self.max_features, root=False)
        return k["xy"].astype(np.float32) * np.float32(2.0), k["desc"]

    def apply(self, gray_half: np.ndarray) -> np.ndarray:
        H = np.eye(2, 3)
        xy, desc = self._features(gray_half)
        prev, self.prev = self.prev, (xy, desc)
        if prev is None or len(prev[0]) < 2 or len(xy) == 0:
            return H
        if self.method == "orb":
            from . import stabilo_ref as S

            qi, ti, _ = S.match(desc, prev[1], 0.9)
        else:
            d = ((desc[:, None, :].astype(np.float32) - prev[1][None, :, :].astype(np.float32)) ** 2).sum(-1)
            order = np.argsort(d, 1, kind="stable")[:, :2]
            d1, d2 = np.sqrt(d[np.arange(len(d)), order[:, 0]]), np.sqrt(d[np.arange(len(d)), order[:, 1]])
            good = d1 < np.float32(0.9) * d2
            qi, ti = np.flatnonzero(good), order[good, 0]
        p, q = prev[0][ti], xy[qi]
        keep = filter_matches(p, q, self.hw)
        self.last = dict(prev=p, cur=q, keep=keep)
        if keep.sum() > 4:
            M = estimate_affine_partial(p[keep], q[keep], self.seed)
            if M is not None:
                H = M
        return H
