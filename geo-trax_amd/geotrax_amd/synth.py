"""Seeded synthetic drone footage: the benchmark clip
``data/U_video_cut.mp4`` is not in the tree and there is no decoder, so workloads are rendered.

A scene is a static textured "world" image plus ``n_vehicles`` bright axis-aligned rectangles that
move a few pixels per frame; the camera follows a smooth homography random walk whose end point
matches the golden stabilization envelope (translation up to about (3, 6) px, rotation ~1e-3,
perspective ~1e-7; measured on data/results-pixel/U_video_cut_vid_transf.txt). Ground truth boxes
and homographies are known by construction, which gives an accuracy check that does not depend on
the (unavailable) reference run.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# box statistics of the golden clip (data/results-pixel/U_video_cut.txt): w 10.9..320 (mean 89),
# h 19.2..154.5 (mean 42), ~132 boxes per 3840x2160 frame.
GOLDEN_BOXES_PER_FRAME = 132


def _smooth_noise(rng, h, w, cell, amp):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.standard_normal((gh, gw)).astype(np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = g[y0][:, x0] * (1 - fx) + g[y0][:, x0 + 1] * fx
    b = g[y0 + 1][:, x0] * (1 - fx) + g[y0 + 1][:, x0 + 1] * fx
    return amp * (a * (1 - fy) + b * fy)


def bilinear_sample(img: np.ndarray, xs: np.ndarray, ys: np.ndarray) -> np.ndarray:
    """img [H,W,C] float32 sampled at float coords (clamped to the border)."""
    h, w = img.shape[:2]
    xs = np.clip(xs, 0, w - 1.001)
    ys = np.clip(ys, 0, h - 1.001)
    x0, y0 = xs.astype(np.int32), ys.astype(np.int32)
    fx, fy = (xs - x0)[..., None], (ys - y0)[..., None]
    a = img[y0, x0] * (1 - fx) + img[y0, x0 + 1] * fx
    b = img[y0 + 1, x0] * (1 - fx) + img[y0 + 1, x0 + 1] * fx
    return a * (1 - fy) + b * fy


@dataclass
class Scene:
    h: int
    w: int
    margin: int
    world: np.ndarray          # [(h+2m),(w+2m),3] float32 background, world coords = frame0 coords + m
    veh_xywh: np.ndarray       # [n,4] frame-0 centre x,y,w,h
    veh_vel: np.ndarray        # [n,2] px / frame
    veh_color: np.ndarray      # [n,3]
    seed: int

    def camera(self, t: int, n_frames: int = 150) -> np.ndarray:
        """G_t: frame-0 pixel coords -> frame-t pixel coords (3x3). Smooth drift that reaches the
        golden envelope at t = n_frames-1."""
        s = t / max(n_frames - 1, 1)
        rng = np.random.default_rng(self.seed + 7919)
        ph = rng.uniform(0, 2 * np.pi, 4)
        tx = -2.96 * s + 0.25 * np.sin(2 * np.pi * 1.5 * s + ph[0]) * s
        ty = -6.00 * s + 0.25 * np.sin(2 * np.pi * 1.1 * s + ph[1]) * s
        th = 1.0e-3 * s * np.sin(2 * np.pi * 0.7 * s + ph[2])
        sc = 1.0 + 2e-4 * s
        p1, p2 = 1.0e-7 * s, -0.8e-7 * s
        cx, cy = self.w / 2, self.h / 2
        c, sn = np.cos(th) * sc, np.sin(th) * sc
        A = np.array([[c, -sn, tx + cx - c * cx + sn * cy], [sn, c, ty + cy - sn * cx - c * cy], [p1, p2, 1.0]])
        return A if t > 0 else np.eye(3)

    def boxes(self, t: int, n_frames: int = 150) -> np.ndarray:
        """Ground-truth xywh boxes in frame-t pixels (axis-aligned hull of the moved rectangle)."""
        G = self.camera(t, n_frames)
        out = []
        for (x, y, w, h), v in zip(self.veh_xywh, self.veh_vel):
            x, y = x + v[0] * t, y + v[1] * t
            cs = np.array([[x - w / 2, y - h / 2, 1], [x + w / 2, y - h / 2, 1], [x + w / 2, y + h / 2, 1], [x - w / 2, y + h / 2, 1]]).T
            p = G @ cs
            p = p[:2] / p[2]
            out.append([(p[0].min() + p[0].max()) / 2, (p[1].min() + p[1].max()) / 2, p[0].max() - p[0].min(), p[1].max() - p[1].min()])
        return np.asarray(out, dtype=np.float32)

    def orthophoto(self, size: int = 4800, scale: float = 1.18, angle: float = 0.21) -> tuple[np.ndarray, np.ndarray]:
        """A synthetic orthophoto of the scene: the static world (no vehicles) seen through a similarity (zoom `scale`,
        rotation `angle`, centred) on a size x size canvas -> (BGR uint8 image, 3x3 ground truth mapping frame-0 pixels to
        orthophoto pixels). What config 4 of SURVEY.md 8d registers the reference frame against."""
        c, s = scale * np.cos(angle), scale * np.sin(angle)
        A = np.array([[c, -s, size / 2 - c * self.w / 2 + s * self.h / 2], [s, c, size / 2 - s * self.w / 2 - c * self.h / 2], [0, 0, 1.0]])
        Ai = np.linalg.inv(A)
        ys, xs = np.mgrid[0:size, 0:size].astype(np.float32)
        fx = Ai[0, 0] * xs + Ai[0, 1] * ys + Ai[0, 2]
        fy = Ai[1, 0] * xs + Ai[1, 1] * ys + Ai[1, 2]
        inside = (fx >= -self.margin + 1) & (fx < self.w + self.margin - 2) & (fy >= -self.margin + 1) & (fy < self.h + self.margin - 2)
        img = bilinear_sample(self.world, fx + self.margin, fy + self.margin)
        img[~inside] = 60.0
        return np.clip(np.rint(img), 0, 255).astype(np.uint8), A

    def orthophoto_large(self, size: int = 15000, scale: float = 1.3, angle: float = 0.2, seed: int = 7) -> tuple[np.ndarray, np.ndarray]:
        """An orthophoto cut-out of the reference's size (`georef.transformation.cutout_width_px: 15000`, default.yaml:154): a
        size x size textured canvas of its own (the same kind of ground as the scene: smooth noise + high-contrast structure, so
        that the whole cut-out yields keypoints, not only the part the video sees) with the scene's static world laid in
        through a similarity (zoom `scale`, rotation `angle`, centred) -> (BGR uint8 image, 3x3 ground truth mapping frame-0
        pixels to orthophoto pixels). Built in row blocks: 675 MB as uint8."""
        rng = np.random.default_rng(seed)
        out = np.empty((size, size, 3), np.uint8)
        cells = [(640, 22.0), (96, 10.0)]
        grids = [rng.standard_normal((size // c + 3, size // c + 3)).astype(np.float32) for c, _ in cells]
        tint = rng.uniform(-6, 6, 3).astype(np.float32)
        xs = np.arange(size, dtype=np.float32)
        step = 1000
        for y0 in range(0, size, step):
            y1 = min(y0 + step, size)
            ys = np.arange(y0, y1, dtype=np.float32)
            base = np.full((y1 - y0, size), 105.0, np.float32)
            for (c, amp), g in zip(cells, grids):
                gy, gx = ys / c, xs / c
                iy, ix = gy.astype(int), gx.astype(int)
                fy, fx = (gy - iy)[:, None], (gx - ix)[None, :]
                a = g[iy][:, ix] * (1 - fx) + g[iy][:, ix + 1] * fx
                b = g[iy + 1][:, ix] * (1 - fx) + g[iy + 1][:, ix + 1] * fx
                base += amp * (a * (1 - fy) + b * fy)
            base += rng.standard_normal(base.shape).astype(np.float32) * 3
            out[y0:y1] = np.clip(np.rint(base[..., None] + tint), 0, 255).astype(np.uint8)
        n_struct = int(2600 * (size / 3840.0) * (size / 2160.0))             # the scene's density of markings and roofs
        bw, bh = rng.integers(3, 40, n_struct), rng.integers(3, 40, n_struct)
        x, y = rng.integers(0, size - 40, n_struct), rng.integers(0, size - 40, n_struct)
        col = np.clip(rng.uniform(30, 230, (n_struct, 1)) + rng.uniform(-8, 8, (n_struct, 3)), 0, 255).astype(np.uint8)
        for k in range(n_struct):
            out[y[k]:y[k] + bh[k], x[k]:x[k] + bw[k]] = col[k]
        # the scene's world, through the similarity, inside the bounding box of its corners
        c, s = scale * np.cos(angle), scale * np.sin(angle)
        A = np.array([[c, -s, size / 2 - c * self.w / 2 + s * self.h / 2], [s, c, size / 2 - s * self.w / 2 - c * self.h / 2], [0, 0, 1.0]])
        Ai = np.linalg.inv(A)
        m = self.margin
        corners = A @ np.array([[-m, self.w + m, self.w + m, -m], [-m, -m, self.h + m, self.h + m], [1, 1, 1, 1.0]])
        bx0, bx1 = max(int(np.floor(corners[0].min())), 0), min(int(np.ceil(corners[0].max())) + 1, size)
        by0, by1 = max(int(np.floor(corners[1].min())), 0), min(int(np.ceil(corners[1].max())) + 1, size)
        for y0 in range(by0, by1, step):
            y1 = min(y0 + step, by1)
            yy, xx = np.mgrid[y0:y1, bx0:bx1].astype(np.float32)
            fx = (Ai[0, 0] * xx + Ai[0, 1] * yy + Ai[0, 2]).astype(np.float32)
            fy = (Ai[1, 0] * xx + Ai[1, 1] * yy + Ai[1, 2]).astype(np.float32)
            inside = (fx >= -m + 1) & (fx < self.w + m - 2) & (fy >= -m + 1) & (fy < self.h + m - 2)
            img = np.clip(np.rint(bilinear_sample(self.world, fx + m, fy + m)), 0, 255).astype(np.uint8)
            blk = out[y0:y1, bx0:bx1]
            blk[inside] = img[inside]
        return out, A

    def render(self, t: int, n_frames: int = 150) -> np.ndarray:
        """Frame t as BGR uint8 [h,w,3]."""
        G = self.camera(t, n_frames)
        Gi = np.linalg.inv(G)
        ys, xs = np.mgrid[0:self.h, 0:self.w].astype(np.float32)
        den = Gi[2, 0] * xs + Gi[2, 1] * ys + Gi[2, 2]
        wx = (Gi[0, 0] * xs + Gi[0, 1] * ys + Gi[0, 2]) / den
        wy = (Gi[1, 0] * xs + Gi[1, 1] * ys + Gi[1, 2]) / den
        if t == 0:
            m = self.margin
            img = self.world[m:m + self.h, m:m + self.w].copy()
        else:
            img = bilinear_sample(self.world, wx + self.margin, wy + self.margin)
        # vehicles: drawn in world (frame-0) coords, anti-aliased by coverage of the pixel centre
        for (x, y, w, h), v, col in zip(self.veh_xywh, self.veh_vel, self.veh_color):
            x, y = x + v[0] * t, y + v[1] * t
            x0, x1 = int(np.floor(x - w / 2 - 8)), int(np.ceil(x + w / 2 + 8))
            y0, y1 = int(np.floor(y - h / 2 - 12)), int(np.ceil(y + h / 2 + 12))
            x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, self.w), min(y1, self.h)
            if x1 <= x0 or y1 <= y0:
                continue
            sx, sy = wx[y0:y1, x0:x1], wy[y0:y1, x0:x1]
            cov = np.clip(w / 2 + 0.5 - np.abs(sx - x), 0, 1) * np.clip(h / 2 + 0.5 - np.abs(sy - y), 0, 1)
            img[y0:y1, x0:x1] = img[y0:y1, x0:x1] * (1 - cov[..., None]) + col * cov[..., None]
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_scene(seed: int = 0, h: int = 2160, w: int = 3840, n_vehicles: int | None = None, margin: int = 32) -> Scene:
    rng = np.random.default_rng(seed)
    H, W = h + 2 * margin, w + 2 * margin
    scale = w / 3840.0
    base = 105 + _smooth_noise(rng, H, W, max(int(160 * scale), 8), 22) + _smooth_noise(rng, H, W, max(int(24 * scale), 4), 10)
    base = base + rng.standard_normal((H, W)).astype(np.float32) * 3
    world = np.repeat(base[..., None], 3, axis=2)
    world += rng.uniform(-6, 6, 3).astype(np.float32)
    # static high-contrast structure (markings, roofs): corners for the keypoint detector
    n_struct = int(2600 * scale * scale) + 40
    for _ in range(n_struct):
        bw, bh = rng.integers(3, max(int(40 * scale), 6)), rng.integers(3, max(int(40 * scale), 6))
        x, y = rng.integers(0, W - bw), rng.integers(0, H - bh)
        world[y:y + bh, x:x + bw] = rng.uniform(30, 230) + rng.uniform(-8, 8, 3)
    n = n_vehicles if n_vehicles is not None else max(int(round(GOLDEN_BOXES_PER_FRAME * scale * scale)), 4)
    vw = np.clip(rng.normal(89, 30, n), 30, 320) * scale
    vh = np.clip(rng.normal(42, 8, n), 19, 150) * scale
    swap = rng.random(n) < 0.3
    vw, vh = np.where(swap, vh, vw), np.where(swap, vw, vh)
    x = rng.uniform(0.03 * w, 0.97 * w, n)
    y = rng.uniform(0.03 * h, 0.97 * h, n)
    speed = rng.uniform(0, 3, n) * scale
    vel = np.where(swap[:, None], np.stack([np.zeros(n), speed], 1), np.stack([speed, np.zeros(n)], 1)) * rng.choice([-1, 1], (n, 1))
    color = rng.uniform(150, 250, (n, 1)) + rng.uniform(-25, 25, (n, 3))
    return Scene(h, w, margin, world.astype(np.float32), np.stack([x, y, vw, vh], 1).astype(np.float32),
                 vel.astype(np.float32), np.clip(color, 0, 255).astype(np.float32), seed)
