"""Read-ahead frame source over the C ABI (gtx_feeder_*): `cap.read()` of the reference's loop
(geotrax/extract.py:146) taken off the thread that drives the detector.

A source that exposes ``raw_layout()`` (frames.Y4mReader, the memory-mapped .npy reader) is read by the
library's own threads (pread into pinned slots, async upload on a copy stream, I420 -> BGR on the GPU);
any other reader is drained by one host thread here that pushes its frames into the same pinned ring.
Either way the engine receives `DeviceBatch` objects: frames already on their way into HBM, ordered
against the consuming detector's stream by an event -- the detector stage thread never touches the file.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import check


class DeviceBatch:
    """`n` consecutive BGR frames at device address `ptr`; `wait_on(ctx)` orders ctx's stream behind their upload."""

    __slots__ = ("ptr", "n", "index", "_feeder")

    def __init__(self, ptr: int, n: int, index: int, feeder: "FrameFeeder | None" = None):
        self.ptr, self.n, self.index, self._feeder = int(ptr), int(n), int(index), feeder

    def wait_on(self, ctx: _lib.Context) -> None:
        if self._feeder is not None:
            self._feeder.wait(self.index, ctx)


class FrameFeeder:
    def __init__(self, frame_hw: tuple[int, int], *, kind: str = "bgr", batch: int = 2, ring: int = 6, device: int | None = None,
                 ctx: _lib.Context | None = None):
        """ctx: a context whose stream carries the transfers (the engine creates one at a chosen place of its stream order:
        which hardware queue the copy stream shares decides whose launches wait behind the transfers); None: a stream of the
        feeder's own on `device`."""
        self.lib = _lib.load()
        self.ctx = ctx
        self.device = ctx.device if ctx is not None else (_lib.default_device() if device is None else device)
        self.h, self.w = int(frame_hw[0]), int(frame_hw[1])
        self.kind = {"bgr": 0, "i420": 1}[kind]
        self.batch, self.ring = int(batch), int(ring)
        self.src_bytes = self.h * self.w * 3 if self.kind == 0 else self.h * self.w + 2 * ((self.h + 1) // 2) * ((self.w + 1) // 2)
        h = C.c_void_p()
        if ctx is not None:
            check(self.lib.gtx_feeder_create_on(ctx.handle, self.h, self.w, self.kind, self.batch, self.ring, C.byref(h)))
        else:
            check(self.lib.gtx_feeder_create(self.device, self.h, self.w, self.kind, self.batch, self.ring, C.byref(h)))
        self.handle = h
        self._pusher = None
        self._push_error = None

    # ---- sources
    def open_file(self, path, offsets, n_threads: int = 3) -> None:
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        check(self.lib.gtx_feeder_open_file(self.handle, str(path).encode(), _lib.ptr(off), len(off), int(n_threads)))

    def open_memory(self, frames, n_threads: int = 3) -> None:
        """`frames`: a sequence of host arrays (BGR ndarrays, or the I420 `data` of Yuv420Frame objects) delivered in order; the
        same array may appear any number of times. The library's own threads copy them into the pinned ring (no Python, no GIL
        on that path); the arrays are kept alive by this object."""
        self._held = [np.ascontiguousarray(f.data if hasattr(f, "data") and hasattr(f, "bgr") else f, dtype=np.uint8) for f in frames]
        for a in self._held:
            if a.nbytes != self.src_bytes:
                raise ValueError(f"a frame has {a.nbytes} bytes, the feeder was built for {self.src_bytes}")
        ptrs = (C.c_void_p * len(self._held))(*[a.ctypes.data for a in self._held])
        check(self.lib.gtx_feeder_open_memory(self.handle, ptrs, len(self._held), int(n_threads)))

    def open_reader(self, frames) -> None:
        """`frames`: iterable of host frames (ndarray BGR, or frames.Yuv420Frame for an I420 feeder), drained on a thread of
        its own. An exception of the iterable ends the source and is re-raised by batches() after the frames before it."""
        check(self.lib.gtx_feeder_open_push(self.handle))

        def run():
            try:
                for f in frames:
                    a = f.data if hasattr(f, "data") and hasattr(f, "bgr") else np.ascontiguousarray(f, dtype=np.uint8)
                    a = np.ascontiguousarray(a, dtype=np.uint8)
                    if self.lib.gtx_feeder_push(self.handle, _lib.ptr(a), a.nbytes) != 0:
                        raise _lib.GtxError(-4, self.lib.gtx_last_error().decode("utf-8", "replace"))
            except BaseException as e:                      # noqa: BLE001 (handed to the consumer)
                self._push_error = e
            finally:
                self.lib.gtx_feeder_finish(self.handle)

        self._pusher = threading.Thread(target=run, name="gtx-feeder-push", daemon=True)
        self._pusher.start()

    def open_indexed(self, frame_at, n_frames: int, threads: int = 3) -> None:
        """Frames 0..n_frames-1 of a random-access host source (`frame_at(i)` -> ndarray / Yuv420Frame), copied into the pinned
        ring by `threads` host threads (gtx_feeder_push_at)."""
        check(self.lib.gtx_feeder_open_push(self.handle))

        def work(t):
            try:
                for i in range(t, n_frames, threads):
                    f = frame_at(i)
                    a = f.data if hasattr(f, "data") and hasattr(f, "bgr") else f
                    a = np.ascontiguousarray(a, dtype=np.uint8)
                    if self.lib.gtx_feeder_push_at(self.handle, i, _lib.ptr(a), a.nbytes) != 0:
                        raise _lib.GtxError(-4, self.lib.gtx_last_error().decode("utf-8", "replace"))
            except BaseException as e:                      # noqa: BLE001 (handed to the consumer)
                if self._push_error is None:
                    self._push_error = e
                # a frame is missing in the middle: nothing behind it can be delivered, and the other producers and the consumer
                # may be waiting for it (or for ring space behind it) -- end the source now
                self.lib.gtx_feeder_stop(self.handle)

        def run():
            ts = [threading.Thread(target=work, args=(t,), name=f"gtx-feeder-push-{t}", daemon=True) for t in range(max(threads, 1))]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            self.lib.gtx_feeder_finish(self.handle)

        self._pusher = threading.Thread(target=run, name="gtx-feeder-push", daemon=True)
        self._pusher.start()

    # ---- consumer
    def next(self) -> DeviceBatch | None:
        p, n, j = C.c_void_p(), C.c_int(), C.c_int64()
        rc = self.lib.gtx_feeder_next(self.handle, C.byref(p), C.byref(n), C.byref(j))
        if rc != 0 and self._push_error is not None:            # the producers' own failure is the one to report
            e, self._push_error = self._push_error, None
            raise e
        check(rc)
        if n.value == 0:
            if self._push_error is not None:
                e, self._push_error = self._push_error, None
                raise e
            return None
        return DeviceBatch(p.value, n.value, j.value, self)

    def wait(self, batch_index: int, ctx: _lib.Context | None) -> None:
        check(self.lib.gtx_feeder_wait(self.handle, int(batch_index), ctx.handle if ctx is not None else None))

    def release(self, n_batches: int) -> None:
        check(self.lib.gtx_feeder_release(self.handle, int(n_batches)))

    def batches(self, in_flight: int):
        """Batches in clip order for ExtractEngine.run(). `in_flight`: how many batches the consumer keeps in flight -- when it
        asks for batch q, the passes over batches <= q - in_flight are complete (the engine submits one batch per detector
        stream and collects the oldest before it pulls the next), so their slots go back to the readers."""
        assert self.ring > in_flight, "the feeder's ring must be deeper than the consumer's pipeline"
        q = 0
        while True:
            if q >= in_flight:
                self.release(q - in_flight + 1)
            b = self.next()
            if b is None:
                return
            yield b
            q += 1

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.gtx_feeder_stop(self.handle)             # a push blocked on a full ring returns with an error
            if self._pusher is not None:
                self._pusher.join()
                self._pusher = None
            self.lib.gtx_feeder_destroy(self.handle)          # joins the library's threads
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
