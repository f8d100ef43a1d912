"""CPU restatement of SIFT / RootSIFT keypoints + descriptors and of the image-to-image registration
built on them -- TEST INFRASTRUCTURE ONLY (imported by tests/, bench.py's cpu_baseline and
__graft_entry__.smoke(); never by the product path).

Path restated: `estimate_homography` (geotrax/utils/registration.py:21-95) -> stabilo.Stabilizer with
detector_name='rsift' -> cv2.SIFT_create(nfeatures, enable_precise_upscale=True).detectAndCompute,
RootSIFT conversion, BFMatcher(NORM_L2).knnMatch(k=2), ratio test, findHomography(USAC_MAGSAC).
stabilo and OpenCV are absent from /root/reference (pyproject pins stabilo>=1.2.3; OpenCV comes with
it), so this follows the published algorithm (Lowe 2004 as implemented by OpenCV's SIFT: sigma 1.6,
3 layers per octave, contrast threshold 0.04, edge threshold 10, image doubled first) and is
**PARITY UNPINNED** against OpenCV itself: no golden vector for this path exists in the reference.
Deliberate differences: atan2/exp from libm instead of OpenCV's fast approximations, Gaussian taps
summed symmetrically in a fixed order (so that the GPU pyramid is bit-identical to this one),
robust fit = MSAC + IRLS (oracle/stabilo_ref.py) instead of MAGSAC++.
"""
from __future__ import annotations

import math

import numpy as np

F = np.float32
N_LAYERS = 3
SIGMA = 1.6
CONTRAST_THR = 0.04
EDGE_THR = 10.0
BORDER = 5
MAX_INTERP = 5
ORI_BINS = 36
ORI_SIG_FCTR = 1.5
ORI_RADIUS = 3 * ORI_SIG_FCTR
ORI_PEAK_RATIO = 0.8
DESCR_WIDTH = 4
DESCR_BINS = 8
DESCR_SCL_FCTR = 3.0
DESCR_MAG_THR = 0.2
INT_DESCR_FCTR = 512.0


def cv_round(x: float) -> int:
    """cvRound: round half to even (lrint)."""
    return int(np.rint(x))


def bgr_to_gray(img: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(BGR2GRAY) on u8: fixed-point weights 1868/9617/4899 >> 14."""
    b, g, r = (img[..., i].astype(np.int32) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def upscale2x(g: np.ndarray) -> np.ndarray:
    """enable_precise_upscale: dst(x, y) = bilinear(src, x/2, y/2), BORDER_REFLECT."""
    h, w = g.shape
    gp = np.concatenate([g, g[-1:]], 0)
    gp = np.concatenate([gp, gp[:, -1:]], 1)              # reflect: index n -> n-1
    out = np.empty((2 * h, 2 * w), F)
    out[0::2, 0::2] = g
    out[0::2, 1::2] = (gp[:h, :w] + gp[:h, 1:w + 1]) * F(0.5)
    out[1::2, 0::2] = (gp[:h, :w] + gp[1:h + 1, :w]) * F(0.5)
    out[1::2, 1::2] = ((gp[:h, :w] + gp[:h, 1:w + 1]) + (gp[1:h + 1, :w] + gp[1:h + 1, 1:w + 1])) * F(0.25)
    return out


def gaussian_taps(sigma: float) -> np.ndarray:
    """cv2.getGaussianKernel(ksize, sigma, CV_32F) with ksize = round(8 sigma + 1) | 1."""
    ksize = cv_round(sigma * 8 + 1) | 1
    r = ksize // 2
    x = np.arange(-r, r + 1, dtype=np.float64)
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).astype(F)


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.abs(i) % p
    return np.where(i >= n, p - i, i)


def blur(img: np.ndarray, sigma: float) -> np.ndarray:
    """Separable Gaussian, BORDER_REFLECT_101. Fixed float32 order: centre tap, then for k = 1..r
    acc = acc + w_k * (x[-k] + x[+k]) with every product and sum rounded to float32 (no fma)."""
    w = gaussian_taps(sigma)
    r = len(w) // 2

    def one_pass(a, axis):
        n = a.shape[axis]
        idx = np.arange(n)
        acc = np.take(a, idx, axis) * w[r]
        for k in range(1, r + 1):
            lo, hi = _reflect101(idx - k, n), _reflect101(idx + k, n)
            acc = acc + w[r + k] * (np.take(a, lo, axis) + np.take(a, hi, axis))
        return acc.astype(F)

    return one_pass(one_pass(img.astype(F), 1), 0)


def downsample2(a: np.ndarray) -> np.ndarray:
    """cv2.resize(a, (w // 2, h // 2), INTER_NEAREST): src index = min(floor(dst * n / (n // 2)), n - 1)."""
    h, w = a.shape
    ys = np.minimum(np.floor(np.arange(h // 2) * (h / (h // 2))).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(w // 2) * (w / (w // 2))).astype(np.int64), w - 1)
    return np.ascontiguousarray(a[np.ix_(ys, xs)])


def layer_sigmas() -> list[float]:
    k = 2.0 ** (1.0 / N_LAYERS)
    sig = [SIGMA]
    for i in range(1, N_LAYERS + 3):
        prev = SIGMA * k ** (i - 1)
        sig.append(math.sqrt((prev * k) ** 2 - prev ** 2))
    return sig


def n_octaves(base_h: int, base_w: int) -> int:
    return cv_round(math.log(min(base_h, base_w)) / math.log(2.0) - 2) + 1     # - firstOctave, firstOctave = -1


def build_pyramids(gray_u8: np.ndarray):
    base = blur(upscale2x(gray_u8.astype(F)), math.sqrt(max(SIGMA * SIGMA - 4 * 0.5 * 0.5, 0.01)))
    sig = layer_sigmas()
    gauss = []
    for o in range(n_octaves(*base.shape)):
        layers = []
        for i in range(N_LAYERS + 3):
            if o == 0 and i == 0:
                layers.append(base)
            elif i == 0:
                layers.append(downsample2(gauss[o - 1][N_LAYERS]))
            else:
                layers.append(blur(layers[i - 1], sig[i]))
        gauss.append(layers)
        if min(layers[0].shape) // 2 < 1:
            break
    dog = [[(g[i + 1] - g[i]).astype(F) for i in range(N_LAYERS + 2)] for g in gauss]
    return gauss, dog


def find_candidates(dog) -> np.ndarray:
    """(octave, layer, r, c) of the scale-space extrema (non-strict comparison, |v| > threshold),
    sorted lexicographically."""
    thr = math.floor(0.5 * CONTRAST_THR / N_LAYERS * 255)
    out = []
    for o, d in enumerate(dog):
        h, w = d[0].shape
        if h <= 2 * BORDER or w <= 2 * BORDER:
            continue
        for i in range(1, N_LAYERS + 1):
            cur = d[i][BORDER:h - BORDER, BORDER:w - BORDER]
            mx = np.full(cur.shape, -np.inf, F)
            mn = np.full(cur.shape, np.inf, F)
            for di in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        if di == 0 and dy == 0 and dx == 0:
                            continue
                        nb = d[i + di][BORDER + dy:h - BORDER + dy, BORDER + dx:w - BORDER + dx]
                        mx = np.maximum(mx, nb)
                        mn = np.minimum(mn, nb)
            ok = (np.abs(cur) > thr) & (((cur > 0) & (cur >= mx)) | ((cur < 0) & (cur <= mn)))
            rr, cc = np.nonzero(ok)
            for r, c in zip(rr, cc):
                out.append((o, i, r + BORDER, c + BORDER))
    return np.asarray(sorted(out), np.int32).reshape(-1, 4)


def _solve3(Hm, b):
    """3x3 solve by LU with partial pivoting in float32 (Matx33f::solve(DECOMP_LU))."""
    A = np.array(Hm, F).copy()
    x = np.array(b, F).copy()
    for k in range(3):
        p = k + int(np.argmax(np.abs(A[k:, k])))
        if abs(A[p, k]) < np.finfo(F).eps:
            return None
        if p != k:
            A[[k, p]] = A[[p, k]]
            x[[k, p]] = x[[p, k]]
        for i in range(k + 1, 3):
            f = F(A[i, k] / A[k, k])
            A[i, k:] = A[i, k:] - f * A[k, k:]
            x[i] = x[i] - f * x[k]
    for k in (2, 1, 0):
        s = x[k]
        for j in range(k + 1, 3):
            s = F(s - A[k, j] * x[j])
        x[k] = F(s / A[k, k])
    return x


def refine(dog_o, octv: int, layer: int, r: int, c: int):
    """adjustLocalExtrema -> (x, y, octave word, size, response, layer, r, c) or None."""
    img_scale = F(1.0 / 255.0)
    ds, sds, cds = F(img_scale * F(0.5)), img_scale, F(img_scale * F(0.25))
    h, w = dog_o[0].shape
    xi = xr = xc = F(0)
    i = 0
    while i < MAX_INTERP:
        img, prv, nxt = dog_o[layer], dog_o[layer - 1], dog_o[layer + 1]
        dD = np.array([(img[r, c + 1] - img[r, c - 1]) * ds, (img[r + 1, c] - img[r - 1, c]) * ds, (nxt[r, c] - prv[r, c]) * ds], F)
        v2 = F(img[r, c] * F(2))
        dxx = (img[r, c + 1] + img[r, c - 1] - v2) * sds
        dyy = (img[r + 1, c] + img[r - 1, c] - v2) * sds
        dss = (nxt[r, c] + prv[r, c] - v2) * sds
        dxy = (img[r + 1, c + 1] - img[r + 1, c - 1] - img[r - 1, c + 1] + img[r - 1, c - 1]) * cds
        dxs = (nxt[r, c + 1] - nxt[r, c - 1] - prv[r, c + 1] + prv[r, c - 1]) * cds
        dys = (nxt[r + 1, c] - nxt[r - 1, c] - prv[r + 1, c] + prv[r - 1, c]) * cds
        X = _solve3([[dxx, dxy, dxs], [dxy, dyy, dys], [dxs, dys, dss]], dD)
        if X is None:
            return None
        xi, xr, xc = F(-X[2]), F(-X[1]), F(-X[0])
        if abs(xi) < 0.5 and abs(xr) < 0.5 and abs(xc) < 0.5:
            break
        if abs(xi) > 2 ** 30 or abs(xr) > 2 ** 30 or abs(xc) > 2 ** 30:
            return None
        c += cv_round(xc); r += cv_round(xr); layer += cv_round(xi)
        if layer < 1 or layer > N_LAYERS or c < BORDER or c >= w - BORDER or r < BORDER or r >= h - BORDER:
            return None
        i += 1
    if i >= MAX_INTERP:
        return None
    img, prv, nxt = dog_o[layer], dog_o[layer - 1], dog_o[layer + 1]
    dD = np.array([(img[r, c + 1] - img[r, c - 1]) * ds, (img[r + 1, c] - img[r - 1, c]) * ds, (nxt[r, c] - prv[r, c]) * ds], F)
    t = F(F(dD[0] * xc) + F(dD[1] * xr)) + F(dD[2] * xi)
    contr = F(img[r, c] * img_scale + F(t * F(0.5)))
    if abs(contr) * N_LAYERS < CONTRAST_THR:
        return None
    v2 = F(img[r, c] * F(2))
    dxx = (img[r, c + 1] + img[r, c - 1] - v2) * sds
    dyy = (img[r + 1, c] + img[r - 1, c] - v2) * sds
    dxy = (img[r + 1, c + 1] - img[r + 1, c - 1] - img[r - 1, c + 1] + img[r - 1, c - 1]) * cds
    tr = F(dxx + dyy)
    det = F(F(dxx * dyy) - F(dxy * dxy))
    if det <= 0 or F(tr * tr) * F(EDGE_THR) >= F((EDGE_THR + 1) ** 2) * det:
        return None
    scale = float(1 << octv)
    x = F((c + xc) * F(scale)); y = F((r + xr) * F(scale))
    octave_word = octv + (layer << 8) + (cv_round((float(xi) + 0.5) * 255) << 16)
    size = F(SIGMA * 2.0 ** ((layer + float(xi)) / N_LAYERS) * scale * 2)
    return float(x), float(y), octave_word, float(size), float(abs(contr)), layer, r, c


def orientation_hist(img: np.ndarray, r: int, c: int, radius: int, sigma: float) -> np.ndarray:
    h, w = img.shape
    ii, jj = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    y, x = r + ii, c + jj
    ok = (y > 0) & (y < h - 1) & (x > 0) & (x < w - 1)
    y, x, ii, jj = y[ok], x[ok], ii[ok], jj[ok]
    dx = (img[y, x + 1] - img[y, x - 1]).astype(F)
    dy = (img[y - 1, x] - img[y + 1, x]).astype(F)
    wgt = np.exp(((ii * ii + jj * jj).astype(F)) * F(-1.0 / (2.0 * sigma * sigma))).astype(F)
    ori = np.degrees(np.arctan2(dy, dx).astype(F)).astype(F)
    ori = np.where(ori < 0, ori + F(360), ori)
    mag = np.sqrt(dx * dx + dy * dy).astype(F)
    b = np.rint(ori * F(ORI_BINS / 360.0)).astype(np.int64) % ORI_BINS
    tmp = np.zeros(ORI_BINS, np.float64)
    np.add.at(tmp, b, (wgt * mag).astype(np.float64))
    tmp = tmp.astype(F)
    t = np.concatenate([tmp[-2:], tmp, tmp[:2]])
    return ((t[:-4] + t[4:]) * F(1 / 16) + (t[1:-3] + t[3:-1]) * F(4 / 16) + t[2:-2] * F(6 / 16)).astype(F)


def keypoint_angles(hist: np.ndarray) -> list[float]:
    n = ORI_BINS
    thr = hist.max() * F(ORI_PEAK_RATIO)
    out = []
    for j in range(n):
        l, r2 = (j - 1) % n, (j + 1) % n
        if hist[j] > hist[l] and hist[j] > hist[r2] and hist[j] >= thr:
            b = j + 0.5 * float(hist[l] - hist[r2]) / float(hist[l] - 2 * hist[j] + hist[r2])
            b = b + n if b < 0 else (b - n if b >= n else b)
            a = float(F(360.0 - (360.0 / n) * b))              # KeyPoint.angle is a float32
            out.append(0.0 if abs(a - 360.0) < 1.19e-7 else a)
    return out


def descriptor(img: np.ndarray, x: float, y: float, ori: float, scl: float) -> np.ndarray:
    """calcSIFTDescriptor: 4x4x8 histogram with trilinear interpolation, 0.2 clamp, x512 -> u8."""
    d, n = DESCR_WIDTH, DESCR_BINS
    h, w = img.shape
    px, py = cv_round(x), cv_round(y)
    cos_t, sin_t = math.cos(math.radians(ori)), math.sin(math.radians(ori))
    hist_width = DESCR_SCL_FCTR * scl
    radius = cv_round(hist_width * 1.4142135623730951 * (d + 1) * 0.5)
    radius = min(radius, int(math.sqrt(h * h + w * w)))
    cos_t, sin_t = F(cos_t / hist_width), F(sin_t / hist_width)
    ii, jj = np.mgrid[-radius:radius + 1, -radius:radius + 1]
    ii, jj = ii.ravel(), jj.ravel()
    c_rot = (jj * cos_t - ii * sin_t).astype(F)
    r_rot = (jj * sin_t + ii * cos_t).astype(F)
    rbin = r_rot + F(d / 2 - 0.5)
    cbin = c_rot + F(d / 2 - 0.5)
    r, c = py + ii, px + jj
    ok = (rbin > -1) & (rbin < d) & (cbin > -1) & (cbin < d) & (r > 0) & (r < h - 1) & (c > 0) & (c < w - 1)
    r, c, rbin, cbin, c_rot, r_rot = r[ok], c[ok], rbin[ok], cbin[ok], c_rot[ok], r_rot[ok]
    dx = (img[r, c + 1] - img[r, c - 1]).astype(F)
    dy = (img[r - 1, c] - img[r + 1, c]).astype(F)
    wgt = np.exp((c_rot * c_rot + r_rot * r_rot) * F(-1.0 / (d * d * 0.5))).astype(F)
    o = np.degrees(np.arctan2(dy, dx).astype(F)).astype(F)
    o = np.where(o < 0, o + F(360), o)
    mag = (np.sqrt(dx * dx + dy * dy) * wgt).astype(F)
    obin = ((o - F(ori)) * F(n / 360.0)).astype(F)
    r0, c0, o0 = np.floor(rbin).astype(np.int64), np.floor(cbin).astype(np.int64), np.floor(obin).astype(np.int64)
    rbin, cbin, obin = rbin - r0.astype(F), cbin - c0.astype(F), obin - o0.astype(F)
    o0 = np.where(o0 < 0, o0 + n, o0)
    o0 = np.where(o0 >= n, o0 - n, o0)
    hist = np.zeros(((d + 2), (d + 2), (n + 2)), np.float64)
    v_r1 = mag * rbin; v_r0 = mag - v_r1
    v_rc11 = v_r1 * cbin; v_rc10 = v_r1 - v_rc11
    v_rc01 = v_r0 * cbin; v_rc00 = v_r0 - v_rc01
    for dr, dc, v in ((0, 0, v_rc00), (0, 1, v_rc01), (1, 0, v_rc10), (1, 1, v_rc11)):
        v1 = v * obin; v0 = v - v1
        np.add.at(hist, (r0 + 1 + dr, c0 + 1 + dc, o0), v0.astype(np.float64))
        np.add.at(hist, (r0 + 1 + dr, c0 + 1 + dc, o0 + 1), v1.astype(np.float64))
    hist = hist.astype(F)
    hist[:, :, 0] += hist[:, :, n]
    hist[:, :, 1] += hist[:, :, n + 1]
    dst = hist[1:d + 1, 1:d + 1, :n].reshape(-1).astype(F)
    thr = F(F(np.sqrt((dst.astype(np.float64) ** 2).sum())) * F(DESCR_MAG_THR))
    dst = np.minimum(dst, thr)
    nrm = F(INT_DESCR_FCTR) / max(F(np.sqrt((dst.astype(np.float64) ** 2).sum())), F(1.19e-7))
    return np.clip(np.rint(dst * nrm), 0, 255).astype(F)


def detect_and_compute(img_bgr: np.ndarray, max_features: int = 250000, root: bool = True, eps: float = 1e-8):
    """-> dict(xy [n,2] f32 full-res pixels, size, angle, response, octave [n], desc [n,128] f32)."""
    gray = bgr_to_gray(img_bgr) if img_bgr.ndim == 3 else img_bgr
    gauss, dog = build_pyramids(gray)
    cand = find_candidates(dog)
    kps = []
    for o, layer, r, c in cand:
        res = refine(dog[o], int(o), int(layer), int(r), int(c))
        if res is None:
            continue
        x, y, word, size, resp, layer2, r2, c2 = res
        scl_octv = size * 0.5 / (1 << int(o))
        hist = orientation_hist(gauss[o][layer2], r2, c2, cv_round(ORI_RADIUS * scl_octv), ORI_SIG_FCTR * scl_octv)
        for a in keypoint_angles(hist):
            kps.append((x, y, size, a, resp, word))
    if len(kps) > max_features:                                   # retainBest: by response (ties by order)
        order = np.argsort(-np.asarray([k[4] for k in kps]), kind="stable")[:max_features]
        kps = [kps[i] for i in sorted(order)]
    xy, size, ang, resp, octs, desc = [], [], [], [], [], []
    for x, y, sz, a, rp, word in kps:
        octave = word & 255
        octave = octave - 256 if octave >= 128 else octave            # stored octave is relative to the doubled base
        layer = (word >> 8) & 255
        g = gauss[octave][layer]
        scale = 1.0 / (1 << octave)
        ori = 360.0 - a
        ori = 0.0 if abs(ori - 360.0) < 1.19e-7 else ori
        dvec = descriptor(g, x * scale, y * scale, ori, sz * scale * 0.5)
        if root:
            dvec = np.sqrt(dvec / F(F(dvec.astype(np.float64).sum()) + F(eps))).astype(F)
        xy.append((x * 0.5, y * 0.5)); size.append(sz * 0.5); ang.append(a); resp.append(rp)
        octs.append((word & ~255) | ((octave - 1) & 255)); desc.append(dvec)
    n = len(xy)
    return dict(xy=np.asarray(xy, F).reshape(n, 2), size=np.asarray(size, F), angle=np.asarray(ang, F), response=np.asarray(resp, F),
                octave=np.asarray(octs, np.int32), desc=np.asarray(desc, F).reshape(n, 128))


def match_ratio(q: np.ndarray, t: np.ndarray, ratio: float):
    """BFMatcher(NORM_L2).knnMatch(k=2) + Lowe ratio: (query idx, train idx, distance)."""
    if len(q) == 0 or len(t) < 2:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, F)
    d = np.sqrt(np.maximum((q * q).sum(1)[:, None] + (t * t).sum(1)[None] - 2.0 * (q.astype(np.float64) @ t.astype(np.float64).T), 0.0))
    o = np.argsort(d, axis=1, kind="stable")[:, :2]
    d1, d2 = d[np.arange(len(q)), o[:, 0]], d[np.arange(len(q)), o[:, 1]]
    keep = d1 < ratio * d2
    return np.nonzero(keep)[0], o[keep, 0], d1[keep].astype(F)
