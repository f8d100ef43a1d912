"""Frame sources for the extract loop.

The reference reads frames with ``cv2.VideoCapture`` (geotrax/extract.py:146,248). OpenCV is not
part of this build, and the benchmark clip is not in the tree, so the reader is an abstraction
with the same three calls the loop uses (``isOpened`` / ``read`` / ``release``) over:

  * ``*.npy``                    one array [F,H,W,3] uint8 BGR (memory-mapped)
  * a directory                  of per-frame ``*.npy`` / ``*.png`` / ``*.jpg`` files (sorted)
  * ``synthetic://?seed=0&frames=150&h=2160&w=3840``   the seeded scene of geotrax_amd.synth
  * ``*.y4m``                    YUV4MPEG2, 8-bit 4:2:0 (``ffmpeg -i clip.mp4 -pix_fmt yuv420p clip.y4m``): the uncompressed
                                 video container; frames travel to the GPU as I420 planes (half the bytes of BGR) and are
                                 converted there (csrc/yuv.hip) -- the route for real footage on a box without a decoder
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


class NpyReader(ArrayReader):
    """One [F,H,W,3] uint8 array in a .npy file, memory-mapped; raw_layout() tells the read-ahead feeder where the frames are."""

    def __init__(self, path: Path):
        self.path = Path(path)
        arr = np.load(self.path, mmap_mode="r")
        if arr.ndim != 4 or arr.shape[-1] != 3 or arr.dtype != np.uint8:
            raise ValueError(f"'{path}': expected a uint8 array of shape [frames, height, width, 3], found {arr.dtype} {arr.shape}")
        super().__init__(arr)

    def raw_layout(self):
        a = self.frames
        if not (isinstance(a, np.memmap) and a.flags["C_CONTIGUOUS"]):
            return None
        step = int(a.shape[1]) * int(a.shape[2]) * 3
        return self.path, "bgr", int(a.offset) + step * np.arange(len(a), dtype=np.int64)


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


class Yuv420Frame:
    """One I420 frame as it sits in the file: `data` = Y plane (h*w) + U + V planes (((h+1)//2)*((w+1)//2) each), uint8.
    The engine uploads `data` and converts on the GPU; `bgr()` is the host conversion (same arithmetic) for callers that
    need an ndarray (reference-frame consumers, tests)."""

    def __init__(self, data: np.ndarray, h: int, w: int):
        self.data, self.h, self.w = data, h, w

    @property
    def shape(self):
        return (self.h, self.w, 3)

    @property
    def nbytes(self) -> int:                               # of the BGR frame it stands for (the engine sizes its staging by it)
        return self.h * self.w * 3

    def bgr(self) -> np.ndarray:
        return yuv420_to_bgr_host(self.data, self.h, self.w)


def yuv420_to_bgr_host(data: np.ndarray, h: int, w: int) -> np.ndarray:
    """Host twin of csrc/yuv.hip (BT.601 limited range, OpenCV's 20-bit fixed point, nearest chroma)."""
    ch, cw = (h + 1) // 2, (w + 1) // 2
    d = np.asarray(data, dtype=np.uint8).reshape(-1)
    y = d[:h * w].reshape(h, w).astype(np.int64)
    u = d[h * w:h * w + ch * cw].reshape(ch, cw).astype(np.int64) - 128
    v = d[h * w + ch * cw:h * w + 2 * ch * cw].reshape(ch, cw).astype(np.int64) - 128
    u, v = np.repeat(np.repeat(u, 2, 0), 2, 1)[:h, :w], np.repeat(np.repeat(v, 2, 0), 2, 1)[:h, :w]
    yl = np.maximum(y - 16, 0) * 1220542
    half = 1 << 19
    b = (yl + half + 2116026 * u) >> 20
    g = (yl + half - 852492 * v - 409993 * u) >> 20
    r = (yl + half + 1673527 * v) >> 20
    return np.clip(np.stack([b, g, r], -1), 0, 255).astype(np.uint8)


class Y4mReader(FrameReader):
    """YUV4MPEG2 (.y4m), 8-bit 4:2:0. Header: 'YUV4MPEG2 W<w> H<h> F<num>:<den> [I..] [A..] [C420*]'; every frame is the line
    'FRAME[ params]' followed by the three planes. read() returns Yuv420Frame objects; seek(i) makes frame i the next one
    (frame-sharded ranks jump to their range without reading what precedes it)."""

    def __init__(self, path: Path):
        self.path = Path(path)
        self.f = open(self.path, "rb")
        head = self.f.readline()
        tok = head.split()
        if not tok or tok[0] != b"YUV4MPEG2":
            self.f.close()
            raise ValueError(f"'{path}' is not a YUV4MPEG2 file")
        par = {t[:1]: t[1:].decode() for t in tok[1:]}
        self.w, self.h = int(par[b"W"]), int(par[b"H"])
        cs = par.get(b"C", "420jpeg")
        if not cs.startswith("420") or "p1" in cs or "p12" in cs or "p16" in cs:
            self.f.close()
            raise NotImplementedError(f"'{path}': colour space C{cs} -- only 8-bit 4:2:0 is implemented (ffmpeg -pix_fmt yuv420p)")
        num, den = (int(v) for v in par.get(b"F", "30:1").split(":"))
        self.fps = num / den if den else 0.0
        self.frame_hw = (self.h, self.w)
        self.frame_bytes = self.h * self.w + 2 * ((self.h + 1) // 2) * ((self.w + 1) // 2)
        self.data_start = self.f.tell()
        # Index of the frames: 'FRAME' lines normally carry no parameters (fixed stride), but the format allows them, so the
        # payload offsets come from one scan of the headers (a seek and a short read per frame) instead of from arithmetic:
        # frame-sharded ranks seek by frame number and must all agree on the frame count.
        # A file that ends inside a frame (an interrupted export) or whose headers go wrong after some complete frames is
        # played up to its last complete frame, like the reference's cv2 loop, which reads until the first failed read
        # (extract.py:146-148); only a file without a single readable frame is refused.
        size = self.path.stat().st_size
        self.offsets, pos = [], self.data_start
        self.truncated = None                              # why the index stops before the end of the file, if it does
        while pos < size:
            self.f.seek(pos)
            line = self.f.readline(256)
            if not line.startswith(b"FRAME") or not line.endswith(b"\n"):
                self.truncated = f"byte {pos} should start a FRAME header (frame {len(self.offsets)}); the file is damaged or not 8-bit 4:2:0"
                break
            pos += len(line) + self.frame_bytes
            if pos > size:
                self.truncated = f"frame {len(self.offsets)} is cut short ({pos - size} bytes missing)"
                break
            self.offsets.append(pos - self.frame_bytes)
        self.frame_count = len(self.offsets)
        if self.truncated is not None:
            if self.frame_count == 0:
                self.f.close()
                raise ValueError(f"'{path}': {self.truncated}")
            import logging

            logging.getLogger(__name__).warning(f"'{path}': {self.truncated}; playing the {self.frame_count} complete frames before it")
        self.i, self._open = 0, self.frame_count > 0

    def raw_layout(self):
        """(path, 'i420', payload offset of every frame): what the read-ahead feeder needs to read the frames itself."""
        return self.path, "i420", np.asarray(self.offsets, dtype=np.int64)

    def seek(self, i: int) -> None:
        self.i = int(i)

    def read(self):
        if self.i >= self.frame_count:
            return False, None
        self.f.seek(self.offsets[self.i])
        buf = np.frombuffer(self.f.read(self.frame_bytes), dtype=np.uint8)
        if len(buf) != self.frame_bytes:
            return False, None
        self.i += 1
        return True, Yuv420Frame(buf, self.h, self.w)

    def release(self):
        self._open = False
        self.f.close()


def bgr_to_i420(frame: np.ndarray) -> bytes:
    """One BGR frame as I420 planes (BT.601 limited range, 2x2 chroma averaging): the payload of a .y4m FRAME."""
    h, w = frame.shape[:2]
    b, g, r = (frame[..., k].astype(np.float64) for k in range(3))
    y = 16 + (65.481 * r + 128.553 * g + 24.966 * b) / 255
    cb = 128 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255
    cr = 128 + (112.0 * r - 93.786 * g - 18.214 * b) / 255

    def sub(c):
        c = np.pad(c, ((0, h % 2), (0, w % 2)), mode="edge")
        return (c[0::2, 0::2] + c[0::2, 1::2] + c[1::2, 0::2] + c[1::2, 1::2]) / 4

    return b"".join(np.clip(np.rint(plane), 0, 255).astype(np.uint8).tobytes() for plane in (y, sub(cb), sub(cr)))


def write_y4m(path, frames_bgr, fps=(30000, 1001)) -> None:
    """Writes BGR frames as a .y4m file (BT.601 limited range, 2x2 chroma averaging) -- for tests and for exporting the
    synthetic clips; real footage comes from `ffmpeg -pix_fmt yuv420p`. An item may also be the bytes bgr_to_i420() made
    (a clip that repeats frames converts each of them once)."""
    frames_bgr = list(frames_bgr)
    first = next(f for f in frames_bgr if not isinstance(f, (bytes, bytearray)))
    h, w = first.shape[:2]
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F{fps[0]}:{fps[1]} Ip A1:1 C420jpeg\n".encode())
        for fr in frames_bgr:
            f.write(b"FRAME\n")
            f.write(fr if isinstance(fr, (bytes, bytearray)) else bgr_to_i420(fr))


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
        return NpyReader(p)
    if p.suffix.lower() == ".y4m":
        return Y4mReader(p)
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
