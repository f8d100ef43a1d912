"""ORACLE -- test infrastructure, not product code.

numpy restatement of the stabilization stage of the reference's hot path: what
`stabilo.Stabilizer.set_ref_frame / stabilize / get_cur_trans_matrix` compute for the reference
(call sites geotrax/extract.py:139,177-187 and geotrax/utils/registration.py:59-85; parameters
geotrax/cfg/default.yaml:100-145: ORB, brute-force Hamming + Lowe ratio 0.9, projective model,
RANSAC threshold 2 px, downsample 0.5, foreground mask grown by 15 %).

stabilo>=1.2.3 (pyproject.toml:59) and the OpenCV calls it makes (ORB_create, BFMatcher.knnMatch,
findHomography(USAC_MAGSAC)) are third-party code that is neither vendored in /root/reference nor
installed here, and OpenCV's ORB sampling table and MAGSAC++ internals cannot be restated from
memory bit for bit. This file therefore restates the *published algorithms* (Rublee et al. ORB:
scale pyramid, FAST-9/16 with 3x3 non-maximum suppression, FAST-score pre-selection then Harris ranking, intensity-centroid
orientation, steered BRIEF on a Gaussian-smoothed patch; Lowe's ratio test; RANSAC with an
MSAC score and a robust iteratively re-weighted Gauss-Newton refit) in the integer-exact form the HIP kernels implement
(geo-trax_amd/csrc/stabilizer.hip), so that every stage can be compared bit for bit. The BRIEF
sampling table is generated HERE (`brief_pattern`: 256 pairs from an isotropic Gaussian, sigma = patch/5,
BRIEF "G II" of Calonder et al., seeded splitmix64, rotated to 256 orientation bins) and the tests compare
the library's table (gtx_stabilizer_pattern) with it, not the other way round.

PARITY UNPINNED against stabilo/OpenCV themselves. What pins the stage end to end: synthetic
sequences with a known camera homography (tests/test_stabilizer_gpu.py) and the envelope of the
reference's golden homographies (tests/golden/U_video_cut_vid_transf.txt).
Held against an independent third-party implementation (scikit-image 0.18.3, NOT a dependency of the reference;
tests/golden/make_independent.py -> tests/test_independent.py): `fast_score > 0` is exactly the pixel set
skimage.feature.corner_fast(n=9) fires on; the homography of a frame pair is within 1 px (9 x 16 grid) of
skimage's ORB + match_descriptors + ransac(ProjectiveTransform) estimate, both within 1 px of the known camera.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

BORDER = 31
CIRCLE = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3),
          (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3)]
UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
GAUSS = np.array([18, 34, 49, 54, 49, 34, 18], dtype=np.int64)
N_BINS = 256
TAN = [int(np.floor(np.tan((j + 0.5) * 2 * np.pi / N_BINS) * 16777216.0 + 0.5)) for j in range(32)]


_M64 = (1 << 64) - 1


def brief_pattern(n_bins: int = N_BINS) -> np.ndarray:
    """Steered-BRIEF sampling table [n_bins][256][ax, ay, bx, by] int8. Base pairs: coordinates drawn from
    N(0, 6.2^2) (sigma = 31/5, BRIEF 'G II'; the normal is Irwin-Hall(12) - 6 over 24-bit splitmix64 uniforms so
    that it is exactly reproducible), rounded half away from zero, redrawn outside [-12, 12]; a pair that would
    compare a pixel with itself is nudged by one. Bin b rotates every point by b * 2 pi / n_bins (ORB steers the
    pattern by the keypoint orientation; OpenCV uses 30 bins of 12 degrees and a learned pair set)."""
    state = [0x9E3779B97F4A7C15]

    def nxt():
        state[0] = (state[0] + 0x9E3779B97F4A7C15) & _M64
        z = state[0]
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def lround(v):
        return int(np.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)

    base = np.zeros((256, 4), np.int64)
    for i in range(256):
        for k in range(4):
            while True:
                acc = sum(nxt() >> 40 for _ in range(12))
                r = lround((acc / 16777216.0 - 6.0) * 6.2)
                if -12 <= r <= 12:
                    base[i, k] = r
                    break
    for i in range(256):
        if base[i, 0] == base[i, 2] and base[i, 1] == base[i, 3]:
            base[i, 2] += -1 if base[i, 2] >= 12 else 1
    out = np.zeros((n_bins, 256, 4), np.int8)
    for b in range(n_bins):
        th = b * (2.0 * np.pi / n_bins)
        cs, sn = float(np.cos(th)), float(np.sin(th))
        for k in range(2):
            x, y = base[:, 2 * k].astype(np.float64), base[:, 2 * k + 1].astype(np.float64)
            for col, v in ((2 * k, x * cs - y * sn), (2 * k + 1, x * sn + y * cs)):
                out[b, :, col] = (np.floor(np.abs(v) + 0.5) * np.sign(v)).astype(np.int8)
    return out


def bgr2gray(frame_bgr: np.ndarray, half: bool) -> np.ndarray:
    f = frame_bgr.astype(np.int32)
    g = (f[..., 0] * 1868 + f[..., 1] * 9617 + f[..., 2] * 4899 + 8192) >> 14
    if half:
        g = (g[0::2, 0::2] + g[0::2, 1::2] + g[1::2, 0::2] + g[1::2, 1::2] + 2) >> 2
    return g.astype(np.uint8)


def resize_int(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """Integer bilinear resize: 16.16 source coordinates, 11-bit weights."""
    sh, sw = src.shape

    def axis(n_dst, n_src):
        d = np.arange(n_dst, dtype=np.int64)
        fp = ((2 * d + 1) * n_src * 32768) // n_dst - 32768
        fp = np.maximum(fp, 0)
        i0 = fp >> 16
        f = (fp >> 5) & 2047
        return np.minimum(i0, n_src - 1), np.minimum(i0 + 1, n_src - 1), f

    x0, x1, fx = axis(dw, sw)
    y0, y1, fy = axis(dh, sh)
    s = src.astype(np.int64)
    top = s[y0][:, x0] * (2048 - fx) + s[y0][:, x1] * fx
    bot = s[y1][:, x0] * (2048 - fx) + s[y1][:, x1] * fx
    return ((top * (2048 - fy)[:, None] + bot * fy[:, None] + (1 << 21)) >> 22).astype(np.uint8)


def level_plan(gw: int, gh: int, n_levels: int, scale_factor: float, max_features: int):
    """(w, h, scale, n_want) per level; feature budget per level as OpenCV's ORB distributes it."""
    factor = 1.0 / np.float32(scale_factor).astype(np.float64)
    want = max_features * (1.0 - factor) / (1.0 - factor ** n_levels)
    out, total = [], 0
    for i in range(n_levels):
        sc = float(np.float32(scale_factor)) ** i
        w, h = int(np.floor(gw / sc + 0.5)), int(np.floor(gh / sc + 0.5))
        if i < n_levels - 1:
            n = int(np.floor(want + 0.5))
            total += n
            want *= factor
        else:
            n = max(max_features - total, 0)
        out.append((w, h, np.float32(sc), n))
    return out


def fast_score(img: np.ndarray, thr: int) -> np.ndarray:
    h, w = img.shape
    score = np.zeros((h, w), dtype=np.uint8)
    if h <= 2 * BORDER or w <= 2 * BORDER:
        return score
    p = img[BORDER:h - BORDER, BORDER:w - BORDER].astype(np.int16)
    d = [img[BORDER + dy:h - BORDER + dy, BORDER + dx:w - BORDER + dx].astype(np.int16) - p for dx, dy in CIRCLE]
    best = np.zeros_like(p)
    for k in range(16):
        mb, md = d[k].copy(), -d[k]
        for j in range(1, 9):
            v = d[(k + j) & 15]
            np.minimum(mb, v, out=mb)
            md = np.minimum(md, -v)
        best = np.maximum(best, np.maximum(mb, md))
    score[BORDER:h - BORDER, BORDER:w - BORDER] = np.where(best > thr, np.minimum(best, 255), 0).astype(np.uint8)
    return score


def harris_keys(img: np.ndarray, ys: np.ndarray, xs: np.ndarray) -> np.ndarray:
    """25(ab - c^2) - (a+b)^2 over the 7x7 block of 3x3 Sobel derivatives, exact int64."""
    I = img.astype(np.int64)
    a = np.zeros(len(xs), dtype=np.int64)
    b = np.zeros_like(a)
    c = np.zeros_like(a)
    for dy in range(-3, 4):
        for dx in range(-3, 4):
            y, x = ys + dy, xs + dx
            ix = (I[y - 1, x + 1] + 2 * I[y, x + 1] + I[y + 1, x + 1]) - (I[y - 1, x - 1] + 2 * I[y, x - 1] + I[y + 1, x - 1])
            iy = (I[y + 1, x - 1] + 2 * I[y + 1, x] + I[y + 1, x + 1]) - (I[y - 1, x - 1] + 2 * I[y - 1, x] + I[y - 1, x + 1])
            a += ix * ix
            b += iy * iy
            c += ix * iy
    return 25 * (a * b - c * c) - (a + b) * (a + b)


def angle_bin(m10: int, m01: int) -> int:
    ax, ay = abs(int(m10)), abs(int(m01))
    swap = ay > ax
    hi, lo = (ay, ax) if swap else (ax, ay)
    o = sum(1 for j in range(32) if (lo << 24) >= hi * TAN[j]) if hi > 0 else 0
    if swap:
        o = 64 - o
    if m10 < 0:
        o = 128 - o
    if m01 < 0:
        o = -o
    return o & (N_BINS - 1)


def mask_rects(boxes_xywh, ratio: float, margin: float, gw: int, gh: int):
    rects = []
    r, m = np.float32(ratio), np.float32(margin)
    for cx, cy, w, h in np.asarray(boxes_xywh, dtype=np.float32).reshape(-1, 4):
        w, h = w * (np.float32(1) + m), h * (np.float32(1) + m)
        x1, y1 = int(np.floor((cx - w / np.float32(2)) * r)), int(np.floor((cy - h / np.float32(2)) * r))
        x2, y2 = int(np.ceil((cx + w / np.float32(2)) * r)), int(np.ceil((cy + h / np.float32(2)) * r))
        x1, y1, x2, y2 = max(x1, 0), max(y1, 0), min(x2, gw - 1), min(y2, gh - 1)
        if x2 >= x1 and y2 >= y1:
            rects.append((x1, y1, x2, y2))
    return rects


def extract(gray: np.ndarray, boxes_xywh, cfg: dict, max_features: int, pattern: np.ndarray):
    """gray: level-0 image (already downsampled). Returns dict(xy [K,2] f32 full-res, level, bin,
    desc [K,32] u8, px [K,2] level pixels), ordered level by level, rank order inside a level."""
    gh, gw = gray.shape
    plan = level_plan(gw, gh, cfg["n_levels"], cfg["scale_factor"], max_features)
    mask = None
    if cfg.get("mask_use", True) and boxes_xywh is not None and len(boxes_xywh):
        mask = np.full((gh, gw), 255, dtype=np.uint8)
        for x1, y1, x2, y2 in mask_rects(boxes_xywh, cfg["downsample_ratio"], cfg["mask_margin_ratio"], gw, gh):
            mask[y1:y2 + 1, x1:x2 + 1] = 0
    inv_ratio = np.float32(1.0) / np.float32(cfg["downsample_ratio"])
    out = dict(xy=[], level=[], bin=[], desc=[], px=[])
    img = gray
    for li, (w, h, sc, n_want) in enumerate(plan):
        if li > 0:
            img = resize_int(img, w, h)
        score = fast_score(img, cfg["fast_threshold"]).astype(np.int16)
        s = score[1:-1, 1:-1]
        keep = s > 0
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx or dy:
                    keep &= s > score[1 + dy:h - 1 + dy, 1 + dx:w - 1 + dx]
        ys, xs = np.nonzero(keep)
        ys, xs = ys + 1, xs + 1
        if mask is not None and len(xs):
            x0 = np.minimum((xs.astype(np.int64) * gw + w // 2) // w, gw - 1)
            y0 = np.minimum((ys.astype(np.int64) * gh + h // 2) // h, gh - 1)
            ok = mask[y0, x0] != 0
            ys, xs = ys[ok], xs[ok]
        if len(xs) == 0 or n_want == 0:
            continue
        # stage 1 (as OpenCV's ORB): the 2*n_want best FAST scores, everything tied with the last kept
        fs = score[ys, xs]
        if len(fs) > 2 * n_want:
            cut = np.sort(fs)[::-1][2 * n_want - 1]
            sel = fs >= cut
            ys, xs = ys[sel], xs[sel]
        # stage 2: the n_want best Harris responses, ties to the smaller pixel index
        keys = harris_keys(img, ys, xs)
        pix = ys.astype(np.int64) * w + xs
        order = np.lexsort((pix, -keys))[:n_want]
        ys, xs = ys[order], xs[order]
        I = img.astype(np.int64)
        for y, x in zip(ys, xs):
            patch = I[y - 20:y + 21, x - 20:x + 21]
            m10 = m01 = 0
            for v in range(-15, 16):
                u = UMAX[abs(v)]
                row = patch[v + 20, 20 - u:20 + u + 1]
                m10 += int((np.arange(-u, u + 1) * row).sum())
                m01 += v * int(row.sum())
            b = angle_bin(m10, m01)
            hpass = sum(GAUSS[k] * patch[:, k:k + 35] for k in range(7))               # [41,35]
            blur = (sum(GAUSS[k] * hpass[k:k + 35, :] for k in range(7)) + 32768) >> 16  # [35,35]
            pt = pattern[b].astype(np.int64)
            bits = blur[pt[:, 1] + 17, pt[:, 0] + 17] < blur[pt[:, 3] + 17, pt[:, 2] + 17]
            words = np.packbits(bits.reshape(4, 64)[:, ::-1], axis=1)[:, ::-1]  # bit t of word g = test 64g+t, little endian
            out["desc"].append(words.reshape(32))
            out["bin"].append(b)
            out["level"].append(li)
            out["px"].append((x, y))
            out["xy"].append((np.float32(x) * sc * inv_ratio, np.float32(y) * sc * inv_ratio))
    k = len(out["bin"])
    return dict(xy=np.asarray(out["xy"], dtype=np.float32).reshape(k, 2), level=np.asarray(out["level"], dtype=np.int32),
                bin=np.asarray(out["bin"], dtype=np.int32), desc=np.asarray(out["desc"], dtype=np.uint8).reshape(k, 32),
                px=np.asarray(out["px"], dtype=np.int32).reshape(k, 2))


_POP = np.array([bin(i).count("1") for i in range(256)], dtype=np.int32)


def match(desc_q: np.ndarray, desc_t: np.ndarray, ratio: float, keep_all: bool = False):
    """Hamming 2-NN of every query against the train set (ties -> lowest index), Lowe ratio test in
    fp32. Returns (q_idx, t_idx, dist) of the good matches in query order. keep_all (stabilo filter_type: none,
    default.yaml:117): every query's nearest neighbour, no test."""
    nq, nt = len(desc_q), len(desc_t)
    if nq == 0 or nt < (1 if keep_all else 2):
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32)
    qi, ti, di = [], [], []
    for s in range(0, nq, 256):
        d = _POP[desc_q[s:s + 256, None, :] ^ desc_t[None, :, :]].sum(-1)
        b = d.argmin(1)
        d1 = d[np.arange(len(b)), b]
        d[np.arange(len(b)), b] = 1 << 30
        d2 = d.min(1)
        good = np.ones(len(b), bool) if keep_all else d1.astype(np.float32) < np.float32(ratio) * d2.astype(np.float32)
        qi.append(np.nonzero(good)[0] + s)
        ti.append(b[good])
        di.append(d1[good])
    return np.concatenate(qi).astype(np.int32), np.concatenate(ti).astype(np.int32), np.concatenate(di).astype(np.int32)


def _hash(x: int) -> int:
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def _denorm(Hn, cx, cy, sc):
    T = np.array([[sc, 0, -sc * cx], [0, sc, -sc * cy], [0, 0, 1.0]])
    return np.linalg.inv(T) @ Hn @ T


def refine_homography(H0, p, q, cx, cy, sc, thr, affine=False):
    """Iteratively re-weighted Gauss-Newton on the transfer error (Tukey biweight, scale from the
    median residual, support = matches within 3 thresholds of H0), h33 = 1, normalised
    coordinates. Mirrors refine_homography() in csrc/stabilizer.hip. affine: h31 = h32 = 0 stay, six
    parameters move, three matches suffice (stabilo transformation_type: affine, default.yaml:121)."""
    npar, min_pts = (6, 3) if affine else (8, 4)
    T = np.array([[sc, 0, -sc * cx], [0, sc, -sc * cy], [0, 0, 1.0]])
    h = T @ H0 @ np.linalg.inv(T)
    h = (h / h[2, 2]).reshape(9).copy()
    x, y, u, v = (p[:, 0] - cx) * sc, (p[:, 1] - cy) * sc, (q[:, 0] - cx) * sc, (q[:, 1] - cy) * sc

    def res(hv):
        w = hv[6] * x + hv[7] * y + 1.0
        return (hv[0] * x + hv[1] * y + hv[2]) / w - u, (hv[3] * x + hv[4] * y + hv[5]) / w - v, w

    rx, ry, w = res(h)
    sup = np.nonzero((np.abs(w) > 1e-9) & (rx * rx + ry * ry <= (3.0 * thr * sc) ** 2))[0]
    if len(sup) < min_pts:
        return None, 0
    for _ in range(8):
        rx, ry, w = res(h)
        un = np.sqrt(rx[sup] ** 2 + ry[sup] ** 2) / sc
        sigma = max(1.4826 * np.sort(un)[len(un) // 2], 0.05)
        c = 4.685 * sigma
        use = un < c
        i = sup[use]
        wt = (1.0 - (un[use] / c) ** 2) ** 2
        iw = 1.0 / w[i]
        px, py = rx[i] + u[i], ry[i] + v[i]
        z = np.zeros_like(iw)
        Jx = np.stack([x[i] * iw, y[i] * iw, iw, z, z, z, -px * x[i] * iw, -px * y[i] * iw], 1)
        Jy = np.stack([z, z, z, x[i] * iw, y[i] * iw, iw, -py * x[i] * iw, -py * y[i] * iw], 1)
        Jx, Jy = Jx[:, :npar], Jy[:, :npar]
        A = (Jx * wt[:, None]).T @ Jx + (Jy * wt[:, None]).T @ Jy
        g = (Jx * (wt * rx[i])[:, None]).sum(0) + (Jy * (wt * ry[i])[:, None]).sum(0)
        try:
            d = np.linalg.solve(A, g)
        except np.linalg.LinAlgError:
            break
        h[:npar] -= d
        if np.abs(d).max() < 1e-14:
            break
    rx, ry, w = res(h)
    n_inl = int((rx * rx + ry * ry <= (thr * sc) ** 2).sum())
    H = np.linalg.inv(T) @ h.reshape(3, 3) @ T
    return (H / H[2, 2], n_inl) if n_inl >= min_pts else (None, 0)


def _errors(H, p, q):
    w = H[2, 0] * p[:, 0] + H[2, 1] * p[:, 1] + H[2, 2]
    dx = (H[0, 0] * p[:, 0] + H[0, 1] * p[:, 1] + H[0, 2]) / w - q[:, 0]
    dy = (H[1, 0] * p[:, 0] + H[1, 1] * p[:, 1] + H[1, 2]) / w - q[:, 1]
    return dx * dx + dy * dy


def ransac_homography(pts_q: np.ndarray, pts_t: np.ndarray, frame_wh, thr: float, n_hyp: int, seed: int, affine: bool = False):
    """pts_*: [n,2] float32 full-res pixels (query -> train). Returns (H 3x3 f64 or None, n_inliers).
    affine: 3-point samples and an affine minimal solve instead of 4-point homographies; same scoring."""
    n = len(pts_q)
    ns = 3 if affine else 4
    if n < ns:
        return None, 0
    p, q = pts_q.astype(np.float64), pts_t.astype(np.float64)
    cx, cy, sc = frame_wh[0] / 2.0, frame_wh[1] / 2.0, 2.0 / frame_wh[0]
    thr2 = float(np.float32(thr) * np.float32(thr))
    best_cost, best_H = None, None
    for hyp in range(n_hyp):
        idx, ctr = [], 0
        while len(idx) < ns:
            c = _hash(seed ^ _hash((hyp * 977 + ctr) & 0xFFFFFFFF)) % n
            ctr += 1
            if c not in idx:
                idx.append(c)
        x, y = (p[idx, 0] - cx) * sc, (p[idx, 1] - cy) * sc
        u, v = (q[idx, 0] - cx) * sc, (q[idx, 1] - cy) * sc
        if affine:
            M = np.stack([x, y, np.ones(3)], 1)
            if not abs(np.linalg.det(M)) > 1e-9:
                continue
            hvec = np.concatenate([np.linalg.solve(M, u), np.linalg.solve(M, v), [0.0, 0.0]])
        else:
            A = np.zeros((8, 8))
            rhs = np.zeros(8)
            for i in range(4):
                A[2 * i] = [x[i], y[i], 1, 0, 0, 0, -u[i] * x[i], -u[i] * y[i]]
                A[2 * i + 1] = [0, 0, 0, x[i], y[i], 1, -v[i] * x[i], -v[i] * y[i]]
                rhs[2 * i], rhs[2 * i + 1] = u[i], v[i]
            try:
                hvec = np.linalg.solve(A, rhs)
            except np.linalg.LinAlgError:
                continue
        H = _denorm(np.append(hvec, 1.0).reshape(3, 3), cx, cy, sc)
        if not abs(H[2, 2]) > 1e-12:
            continue
        w = H[2, 0] * p[:, 0] + H[2, 1] * p[:, 1] + H[2, 2]
        ok = np.abs(w) > 1e-12                      # points on the line at infinity cost the full threshold
        with np.errstate(all="ignore"):
            e = np.where(ok, np.minimum(_errors(H, p, q), thr2), thr2)
        cost = int(np.floor(e * 1024.0 + 0.5).sum())
        if best_cost is None or cost < best_cost:
            best_cost, best_H = cost, H
    if best_H is None:
        return None, 0
    return refine_homography(best_H / best_H[2, 2], p, q, cx, cy, sc, float(np.float32(thr)), affine)


class StabilizerRef:
    """Same call sequence as the product's Stabilizer (set_ref_frame / stabilize)."""

    def __init__(self, cfg: dict, frame_hw, pattern: np.ndarray | None = None, n_hyp: int = 2048):
        self.cfg, self.hw, self.n_hyp = cfg, frame_hw, n_hyp
        self.pattern = brief_pattern() if pattern is None else pattern      # the oracle's own table unless one is forced
        self.half = cfg["downsample_ratio"] == 0.5
        self.ref = self.cur = None

    def _gray(self, frame_bgr):
        g = bgr2gray(frame_bgr, self.half)
        if self.cfg.get("clahe"):                       # stabilo: CLAHE on the working-resolution gray, before detection
            from .clahe_ref import clahe
            g = clahe(g)
        return g

    def set_ref_frame(self, frame_bgr, boxes=None):
        n_ref = int(np.floor(self.cfg["max_features"] * self.cfg["ref_multiplier"] + 0.5))
        self.ref = extract(self._gray(frame_bgr), boxes, self.cfg, n_ref, self.pattern)

    def stabilize(self, frame_bgr, boxes=None):
        self.cur = extract(self._gray(frame_bgr), boxes, self.cfg, self.cfg["max_features"], self.pattern)
        self.m = match(self.cur["desc"], self.ref["desc"], self.cfg["filter_ratio"], keep_all=self.cfg.get("filter_type") == "none")
        qi, ti, _ = self.m
        return ransac_homography(self.cur["xy"][qi], self.ref["xy"][ti], (self.hw[1], self.hw[0]),
                                 self.cfg["ransac_threshold"], self.n_hyp, self.cfg["seed"],
                                 affine=self.cfg.get("transformation_type") == "affine")
