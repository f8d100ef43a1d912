"""Frame sources for the extract loop.

The reference reads frames with ``cv2.VideoCapture`` (geotrax/extract.py:146,248). OpenCV is not
part of this build, and the benchmark clip is not in the tree, so the reader is an abstraction
with the same three calls the loop uses (``isOpened`` / ``read`` / ``release``) over:

  * ``*.npy``                    one array [F,H,W,3] uint8 BGR (memory-mapped)
  * a directory                  of per-frame ``*.npy`` / ``*.png`` / ``*.jpg`` files (sorted)
  * ``synthetic://?seed=0&frames=150&h=2160&w=3840``   the seeded scene of geotrax_amd.synth
  * any other file (``.mp4`` ...) through cv2 when it is importable
"""
from __future__ import annotations

from pathlib import Path
from urllib.parse import parse_qs, urlparse

import numpy as np

VIDEO_SUFFIXES = {".mp4", ".avi", ".mov", ".mkv", ".m4v"}


class FrameReader:
    frame_count = 0
    frame_hw = (0, 0)

    def isOpened(self) -> bool:  # noqa: N802 (cv2 naming, the loop calls it)
        return self._open

    def release(self) -> None:
        self._open = False

    def read(self):
        raise NotImplementedError


class ArrayReader(FrameReader):
    def __init__(self, frames):
        self.frames, self.i, self._open = frames, 0, True
        self.frame_count = len(frames)
        self.frame_hw = tuple(frames[0].shape[:2]) if len(frames) else (0, 0)

    def read(self):
        if self.i >= self.frame_count:
            return False, None
        f = np.ascontiguousarray(self.frames[self.i])
        self.i += 1
        return True, f


class DirReader(FrameReader):
    def __init__(self, path: Path):
        self.files = sorted(p for p in path.iterdir() if p.suffix.lower() in (".npy", ".png", ".jpg", ".jpeg", ".bmp"))
        self.i, self._open = 0, bool(self.files)
        self.frame_count = len(self.files)
        self.frame_hw = self._load(self.files[0]).shape[:2] if self.files else (0, 0)

    @staticmethod
    def _load(p: Path) -> np.ndarray:
        if p.suffix.lower() == ".npy":
            return np.load(p)
        from PIL import Image

        return np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[..., ::-1])  # -> BGR

    def read(self):
        if self.i >= self.frame_count:
            return False, None
        f = self._load(self.files[self.i])
        self.i += 1
        return True, np.ascontiguousarray(f, dtype=np.uint8)


class SyntheticReader(FrameReader):
    def __init__(self, seed=0, frames=150, h=2160, w=3840):
        from .synth import make_scene

        self.scene = make_scene(seed=seed, h=h, w=w)
        self.frame_count, self.frame_hw, self.i, self._open = frames, (h, w), 0, True

    def read(self):
        if self.i >= self.frame_count:
            return False, None
        f = self.scene.render(self.i, self.frame_count)
        self.i += 1
        return True, f


class Cv2Reader(FrameReader):
    def __init__(self, path: Path):
        import cv2

        self.cap = cv2.VideoCapture(str(path))
        self._open = self.cap.isOpened()
        self.frame_count = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))
        self.frame_hw = (int(self.cap.get(cv2.CAP_PROP_FRAME_HEIGHT)), int(self.cap.get(cv2.CAP_PROP_FRAME_WIDTH)))

    def read(self):
        return self.cap.read()

    def release(self):
        self.cap.release()
        self._open = False


def open_source(source) -> FrameReader:
    s = str(source)
    if s.startswith("synthetic:"):
        q = {k: int(v[0]) for k, v in parse_qs(urlparse(s).query).items()}
        return SyntheticReader(**q)
    p = Path(s)
    if p.is_dir():
        return DirReader(p)
    if p.suffix == ".npy":
        return ArrayReader(np.load(p, mmap_mode="r"))
    try:
        return Cv2Reader(p)
    except ImportError as e:
        raise RuntimeError(f"'{p}' needs a video decoder (cv2) which is not installed; export the frames to a "
                           ".npy array / a directory of images, or use synthetic://") from e


def source_exists(source) -> bool:
    s = str(source)
    return s.startswith("synthetic:") or Path(s).exists()


def get_video_dimensions(source) -> tuple[int, int]:
    """(width, height) like the reference's file_utils.get_video_dimensions (file_utils.py:183-189)."""
    r = open_source(source)
    h, w = r.frame_hw
    r.release()
    return w, h
