"""Agreement metrics between two extract runs ("mAP vs reference" made concrete).

Used by tools/score_run.py to score this build's `<stem>.txt` / `<stem>_vid_transf.txt` against files an operator
produced with the real reference (geotrax extract on the same video), and by the tests on the golden clip. The
reference's own tools judge agreement the same way: centre distance of matched boxes
(tools/compute_bb_center_error.py:13-15,169-196) and track count / length statistics
(tools/compare_tracking.py:177-231).
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment


def iou_xywh(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """[n,4] x [m,4] centre/size boxes -> [n,m] IoU."""
    ax1, ay1, ax2, ay2 = a[:, 0] - a[:, 2] / 2, a[:, 1] - a[:, 3] / 2, a[:, 0] + a[:, 2] / 2, a[:, 1] + a[:, 3] / 2
    bx1, by1, bx2, by2 = b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2
    iw = np.clip(np.minimum(ax2[:, None], bx2[None]) - np.maximum(ax1[:, None], bx1[None]), 0, None)
    ih = np.clip(np.minimum(ay2[:, None], by2[None]) - np.maximum(ay1[:, None], by1[None]), 0, None)
    inter = iw * ih
    return inter / (a[:, 2:3] * a[:, 3:4] + (b[:, 2] * b[:, 3])[None] - inter + 1e-12)


def box_agreement(ours: np.ndarray, ref: np.ndarray, cols=(2, 6), iou_thresholds=np.arange(0.5, 0.96, 0.05)) -> dict:
    """Per-frame one-to-one matching (maximum total IoU, Hungarian) of the boxes of two track tables (columns: frame, id,
    xywh at cols[0]:cols[1], ...). The reference table plays ground truth. -> precision / recall / F1 at IoU 0.5, the
    mean over IoU thresholds .5:.95 of F1 ("AP-like" summary without scores), centre error statistics of the pairs
    matched at IoU >= 0.5, and the identity-consistency of the matches (how many reference tracks keep one id here)."""
    frames = np.union1d(np.unique(ours[:, 0]), np.unique(ref[:, 0]))
    tp = np.zeros(len(iou_thresholds))
    n_ours = n_ref = 0
    centre_err, id_pairs = [], []
    for f in frames:
        a, b = ours[ours[:, 0] == f], ref[ref[:, 0] == f]
        n_ours += len(a)
        n_ref += len(b)
        if len(a) == 0 or len(b) == 0:
            continue
        iou = iou_xywh(a[:, cols[0]:cols[1]], b[:, cols[0]:cols[1]])
        r, c = linear_sum_assignment(-iou)
        m = iou[r, c]
        for k, t in enumerate(iou_thresholds):
            tp[k] += int((m >= t).sum())
        ok = m >= 0.5
        centre_err += list(np.hypot(a[r[ok], cols[0]] - b[c[ok], cols[0]], a[r[ok], cols[0] + 1] - b[c[ok], cols[0] + 1]))
        id_pairs += list(zip(a[r[ok], 1].astype(int), b[c[ok], 1].astype(int)))
    prec, rec = tp / max(n_ours, 1), tp / max(n_ref, 1)
    f1 = 2 * prec * rec / np.maximum(prec + rec, 1e-12)
    by_ref = {}
    for o, r_ in id_pairs:
        by_ref.setdefault(r_, set()).add(o)
    ce = np.asarray(centre_err) if centre_err else np.zeros(1)
    return {"boxes_ours": int(n_ours), "boxes_ref": int(n_ref), "precision@0.5": float(prec[0]), "recall@0.5": float(rec[0]), "f1@0.5": float(f1[0]),
            "f1@[.5:.95]": float(f1.mean()), "centre_error_px": {"mean": float(ce.mean()), "median": float(np.median(ce)), "p95": float(np.percentile(ce, 95)),
                                                                    "max": float(ce.max())},
            "ref_tracks_matched": len(by_ref), "ref_tracks_with_one_id_here": int(sum(1 for v in by_ref.values() if len(v) == 1)),
            "id_fragmentations": int(sum(len(v) - 1 for v in by_ref.values()))}


def track_statistics(t: np.ndarray) -> dict:
    """Track count, length distribution and missing-frame counts (tools/compare_tracking.py:177-231)."""
    ids = np.unique(t[:, 1])
    lengths, missing = [], []
    for i in ids:
        fr = np.sort(t[t[:, 1] == i, 0])
        lengths.append(len(fr))
        missing.append(int(fr[-1] - fr[0] + 1 - len(fr)))
    L = np.asarray(lengths)
    return {"tracks": int(len(ids)), "rows": int(len(t)), "length": {"mean": float(L.mean()), "median": float(np.median(L)), "min": int(L.min()), "max": int(L.max())},
            "frames_missing_inside_tracks": int(np.sum(missing))}


def homography_agreement(ours: np.ndarray, ref: np.ndarray, frame_wh=(3840, 2160)) -> dict:
    """Two `_vid_transf` tables (frame, h11..h33): per common frame the largest reprojection difference over a 9 x 16 grid of
    frame points; proposed bar 1.0 px (the RANSAC inlier threshold is 2 px)."""
    w, h = frame_wh
    ys, xs = np.meshgrid(np.linspace(0, h - 1, 9), np.linspace(0, w - 1, 16), indexing="ij")
    P = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    a = {int(r[0]): r[1:].reshape(3, 3) for r in ours}
    b = {int(r[0]): r[1:].reshape(3, 3) for r in ref}
    common = sorted(set(a) & set(b))
    d = []
    for f in common:
        pa, pb = a[f] @ P, b[f] @ P
        d.append(float(np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max()))
    d = np.asarray(d) if d else np.zeros(1)
    return {"frames_ours": len(a), "frames_ref": len(b), "frames_common": len(common), "grid_diff_px": {"mean": float(d.mean()), "p95": float(np.percentile(d, 95)),
                                                                                                      "max": float(d.max())}}


def homography_envelope(T: np.ndarray, frame_wh=(3840, 2160)) -> dict:
    """Shape of a `_vid_transf` table: what the golden file of the reference looks like and what any
    sane stabilization of hovering-drone footage must look like: positive determinant, h33 = 1, tiny perspective terms,
    small rotation, translation that drifts smoothly."""
    M = T[:, 1:].reshape(-1, 3, 3)
    M = M / M[:, 2:3, 2:3]
    step = np.abs(np.diff(M[:, :2, 2], axis=0)).max() if len(M) > 1 else 0.0
    return {"n": int(len(M)), "det_min": float(np.linalg.det(M).min()), "h33_dev_max": float(np.abs(T[:, 9] - 1).max()),
            "perspective_abs_max": float(np.abs(M[:, 2, :2]).max()), "rotation_abs_max": float(np.abs(np.arctan2(M[:, 1, 0], M[:, 0, 0])).max()),
            "scale_dev_max": float(np.abs(np.sqrt(np.abs(np.linalg.det(M[:, :2, :2]))) - 1).max()),
            "translation_abs_max": [float(np.abs(M[:, 0, 2]).max()), float(np.abs(M[:, 1, 2]).max())], "translation_step_max": float(step)}
