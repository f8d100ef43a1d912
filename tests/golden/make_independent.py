"""Independent third-party vectors for the oracles whose upstream packages (OpenCV, stabilo, ultralytics) are absent.

The build container holds a second interpreter, /opt/conda/bin/python3.9, with scikit-image 0.18.3 and scipy 1.7.1.
scikit-image is NOT a dependency of the reference, so nothing here can turn "parity unpinned" into "pinned against the
reference's own stack"; what it gives is an implementation of FAST, ORB + RANSAC, bilinear warping / resizing and BT.601
that nobody in this repository wrote, to hold the co-designed oracles (oracle/stabilo_ref.py, gmc_ref.py, warp_ref.py,
yolov8_ref.resize_linear_u8, yuv_ref.py) and the HIP kernels against:

    fast{0,1}_bits     skimage.feature.corner_fast(n=9): the set of pixels whose FAST-9/16 response is non-zero
                       (before non-maximum suppression / ranking) on two pyramid images of a rendered frame
    orb_H_{40,149}     ORB + match_descriptors + measure.ransac(ProjectiveTransform): frame t -> frame 0 homography of the
                       synthetic 720p clip (seed 3); orb_S_1: the same with SimilarityTransform for frames 0 -> 1 (what
                       BoT-SORT's GMC estimates)
    warp_out           transform.warp(order=1) of a colour crop under a homography
    resize_out         transform.resize(order=1, anti_aliasing=False) of a colour crop by a non-2x ratio (the general
                       letterbox path)
    yuv_bgr            color.ycbcr2rgb of an I420 image
    match_pairs        feature.match_descriptors(metric="hamming", cross_check=False, max_ratio=0.9) on seeded 256-bit
                       descriptor sets (planted near-duplicates + distractors): the Hamming 2-NN + Lowe ratio stage alone
    ransac_H           measure.ransac(ProjectiveTransform) on seeded point matches (known homography, 0.5 px noise, 30 %
                       outliers): the estimation stage alone

Run (build container only):   python tests/golden/make_independent.py
The script renders the inputs with the repository's own seeded generator under the default interpreter, hands them to
the conda interpreter through a temporary file and writes tests/golden/independent_skimage.npz. The inputs are NOT
stored: tests rebuild them with `inputs()` below and check their CRC32 against the fixture's.
"""
from __future__ import annotations

import subprocess
import sys
import tempfile
import zlib
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
OUT = HERE / "independent_skimage.npz"
CONDA_PY = "/opt/conda/bin/python3.9"
HW = (720, 1280)
SEED = 3
FAST_THR = 20
WARP_H = np.array([[1.004, -0.013, 5.3], [0.011, 0.997, -3.7], [1.5e-5, -2.0e-5, 1.0]])
RESIZE_TO = (192, 341)          # 270 x 480 -> 0.711 (a 2.7K source letterboxed to imgsz 1920 has this ratio)


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def gray_u8(frame_bgr: np.ndarray, half: bool) -> np.ndarray:
    """BT.601 luma in 14-bit fixed point (+ the exact 2x2 mean): plain integer arithmetic, written out here so that
    the inputs do not depend on any oracle module."""
    f = frame_bgr.astype(np.int32)
    g = (f[..., 0] * 1868 + f[..., 1] * 9617 + f[..., 2] * 4899 + 8192) >> 14
    if half:
        g = (g[0::2, 0::2] + g[0::2, 1::2] + g[1::2, 0::2] + g[1::2, 1::2] + 2) >> 2
    return g.astype(np.uint8)


def inputs() -> dict:
    """Everything the skimage stage consumes, rebuilt from seeds (used by this script and by the tests)."""
    for p in (ROOT / "geo-trax_amd", ROOT):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=SEED, h=HW[0], w=HW[1])
    fr = {t: sc.render(t) for t in (0, 1, 40, 149)}
    g_half = gray_u8(fr[0], True)                                    # level 0 of the stabilizer's pyramid (downsample 0.5)
    # a second, smaller image with different statistics: every other pixel of the full-resolution gray of frame 40, cropped
    g_sub = np.ascontiguousarray(gray_u8(fr[40], False)[40:680:2, 100:1180:2])
    rng = np.random.default_rng(11)
    crop = np.ascontiguousarray(fr[0][200:380, 300:620])             # 180 x 320 x 3
    rsrc = np.ascontiguousarray(fr[149][100:370, 500:980])           # 270 x 480 x 3
    yh, yw = 64, 96
    Y = np.clip(gray_u8(fr[0], False)[300:300 + yh, 400:400 + yw].astype(np.int32) * 219 // 255 + 16, 16, 235).astype(np.uint8)
    U = rng.integers(60, 200, (yh // 2, yw // 2), dtype=np.uint8)
    V = rng.integers(60, 200, (yh // 2, yw // 2), dtype=np.uint8)
    i420 = np.concatenate([Y.ravel(), U.ravel(), V.ravel()])
    # descriptor sets for the matcher: 1200 random 256-bit train descriptors; 600 queries = a train row with 0-70 random bits
    # flipped (most pass the ratio test, the heavily damaged ones do not) or, for a fifth of them, unrelated random bits
    rd = np.random.default_rng(23)
    d_train = rd.integers(0, 256, (1200, 32), dtype=np.uint8)
    src_rows = rd.integers(0, 1200, 600)
    d_query = d_train[src_rows].copy()
    bits = np.unpackbits(d_query, axis=1)
    for i in range(600):
        if i % 5 == 4:
            bits[i] = rd.integers(0, 2, 256)
        else:
            flip = rd.choice(256, size=int(rd.integers(0, 71)), replace=False)
            bits[i, flip] ^= 1
    d_query = np.packbits(bits, axis=1)
    # point matches for the estimator: a homography of the size the stabilizer sees, 0.5 px noise, 30 % gross outliers
    H_true = np.array([[1.002, -0.004, 7.5], [0.0035, 0.998, -4.25], [2.0e-6, -1.5e-6, 1.0]])
    pts_p = np.stack([rd.uniform(0, HW[1], 500), rd.uniform(0, HW[0], 500)], 1)
    qh = np.c_[pts_p, np.ones(500)] @ H_true.T
    pts_q = qh[:, :2] / qh[:, 2:] + rd.normal(0, 0.5, (500, 2))
    out_idx = rd.choice(500, 150, replace=False)
    pts_q[out_idx] = np.stack([rd.uniform(0, HW[1], 150), rd.uniform(0, HW[0], 150)], 1)
    return dict(f0=fr[0], f1=fr[1], f40=fr[40], f149=fr[149], fast0=g_half, fast1=g_sub, warp_src=crop, resize_src=rsrc,
                i420=i420, yuv_hw=np.array([yh, yw]), d_query=d_query, d_train=d_train, pts_p=pts_p.astype(np.float32),
                pts_q=pts_q.astype(np.float32), H_true=H_true, scene=sc)


def stage_skimage(tmp: str) -> None:
    """Runs under /opt/conda/bin/python3.9."""
    import warnings

    warnings.filterwarnings("ignore")
    import scipy
    import skimage
    from skimage.color import ycbcr2rgb
    from skimage.feature import ORB, corner_fast, match_descriptors
    from skimage.measure import ransac
    from skimage.transform import ProjectiveTransform, SimilarityTransform, resize, warp

    d = np.load(tmp)
    out = dict(skimage_version=np.array(skimage.__version__), scipy_version=np.array(scipy.__version__))
    for k in ("fast0", "fast1"):
        # strict "> p + t" on [0, 1] floats; t = (thr + 0.5) / 255 puts the cut between the integers thr and thr + 1
        r = corner_fast(d[k], n=9, threshold=(FAST_THR + 0.5) / 255.0)
        out[k + "_bits"] = np.packbits(r > 0)
        out[k + "_shape"] = np.array(r.shape)
        out[k + "_crc"] = np.array(crc(d[k]), dtype=np.uint32)

    def orb(frame):
        o = ORB(n_keypoints=1500, fast_n=9, fast_threshold=(FAST_THR + 0.5) / 255.0, downscale=1.2, n_scales=8)
        o.detect_and_extract(gray_u8(frame, False))
        return o.keypoints[:, ::-1].copy(), o.descriptors.copy()       # (x, y)

    k0, d0 = orb(d["f0"])
    for t, model_cls, name in ((40, ProjectiveTransform, "orb_H_40"), (149, ProjectiveTransform, "orb_H_149"), (1, SimilarityTransform, "orb_S_1")):
        kt, dt = orb(d["f%d" % t])
        if model_cls is ProjectiveTransform:                            # the stabilizer maps the current frame onto frame 0
            m = match_descriptors(dt, d0, cross_check=True, max_ratio=0.9)
            src, dst = kt[m[:, 0]], k0[m[:, 1]]
        else:                                                           # the GMC maps the previous frame onto the current one
            m = match_descriptors(d0, dt, cross_check=True, max_ratio=0.9)
            src, dst = k0[m[:, 0]], kt[m[:, 1]]
        model, inl = ransac((src, dst), model_cls, min_samples=4 if model_cls is ProjectiveTransform else 2, residual_threshold=2.0,
                            max_trials=3000, random_state=np.random.RandomState(0))
        out[name] = model.params.astype(np.float64)
        out[name + "_stats"] = np.array([len(kt), len(m), int(inl.sum())])
    for k in ("f0", "f1", "f40", "f149"):
        out[k + "_crc"] = np.array(crc(d[k]), dtype=np.uint32)

    w = warp(d["warp_src"].astype(np.float64), ProjectiveTransform(matrix=np.linalg.inv(WARP_H)), order=1, mode="constant", cval=0.0,
             preserve_range=True)
    out["warp_out"] = np.clip(np.rint(w), 0, 255).astype(np.uint8)
    out["warp_src_crc"] = np.array(crc(d["warp_src"]), dtype=np.uint32)
    out["warp_H"] = WARP_H
    r = resize(d["resize_src"].astype(np.float64), RESIZE_TO, order=1, mode="edge", anti_aliasing=False, preserve_range=True)
    out["resize_out"] = np.clip(np.rint(r), 0, 255).astype(np.uint8)
    out["resize_src_crc"] = np.array(crc(d["resize_src"]), dtype=np.uint32)
    yh, yw = (int(v) for v in d["yuv_hw"])
    i420 = d["i420"]
    Y = i420[:yh * yw].reshape(yh, yw).astype(np.float64)
    U = i420[yh * yw:yh * yw + yh * yw // 4].reshape(yh // 2, yw // 2).astype(np.float64)
    V = i420[yh * yw + yh * yw // 4:].reshape(yh // 2, yw // 2).astype(np.float64)
    ycc = np.stack([Y, np.repeat(np.repeat(U, 2, 0), 2, 1), np.repeat(np.repeat(V, 2, 0), 2, 1)], -1)
    rgb = ycbcr2rgb(ycc)                                                 # ITU-R BT.601, limited range -> RGB in [0, 1]
    out["yuv_bgr"] = np.clip(np.rint(rgb[..., ::-1] * 255.0), 0, 255).astype(np.uint8)
    out["i420_crc"] = np.array(crc(i420), dtype=np.uint32)
    # Hamming 2-NN + Lowe ratio on bit descriptors (rows of booleans: scipy's hamming = fraction of differing bits)
    bq, bt = np.unpackbits(d["d_query"], axis=1).astype(bool), np.unpackbits(d["d_train"], axis=1).astype(bool)
    out["match_pairs"] = match_descriptors(bq, bt, metric="hamming", cross_check=False, max_ratio=0.9).astype(np.int32)
    out["d_query_crc"] = np.array(crc(d["d_query"]), dtype=np.uint32)
    out["d_train_crc"] = np.array(crc(d["d_train"]), dtype=np.uint32)
    model, inl = ransac((d["pts_p"].astype(np.float64), d["pts_q"].astype(np.float64)), ProjectiveTransform, min_samples=4, residual_threshold=2.0,
                        max_trials=3000, random_state=np.random.RandomState(0))
    out["ransac_H"] = model.params.astype(np.float64)
    out["ransac_inliers"] = np.array(int(inl.sum()))
    out["pts_p_crc"] = np.array(crc(d["pts_p"]), dtype=np.uint32)
    out["pts_q_crc"] = np.array(crc(d["pts_q"]), dtype=np.uint32)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.shape if v.ndim else v.item()) for k, v in out.items() if "stats" in k or "version" in k})


def main() -> None:
    if len(sys.argv) > 2 and sys.argv[1] == "--stage-skimage":
        return stage_skimage(sys.argv[2])
    d = inputs()
    d.pop("scene")
    with tempfile.TemporaryDirectory() as td:
        tmp = str(Path(td) / "inputs.npz")
        np.savez(tmp, **d)
        subprocess.run([CONDA_PY, str(Path(__file__).resolve()), "--stage-skimage", tmp], check=True)


if __name__ == "__main__":
    main()
