"""The pipelined extract engine: detect -> (GMC) -> track -> stabilize over a stream of frame batches.

This is the hot loop of the reference (geotrax/extract.py:145-197) re-scheduled for one MI355X without
changing any per-frame result: the HIP detector works on batch k+D while the host runs the tracker over
batch k in clip order and the stabilizer objects (own HIP stream each) register frames t, t+1, .. against
the reference frame; results come back strictly in frame order. `geotrax_amd.extract.track_with_model`
(the product path) and `bench.py` both drive this class, through the asynchronous C-ABI pairs
`gtx_detector_submit_dev/_collect`, `gtx_stabilizer_submit_gray_dev/_collect`, `gtx_gmc_submit_gray_dev/
_collect` (include/gtx.h).

Ordering rules kept from the reference loop:
  * the first frame fed is the reference frame: its boxes pass through unstabilized, no transform row
    (extract.py:176-179);
  * a frame without detections still reaches the tracker (and, with BoT-SORT, the GMC): the pinned
    ultralytics (>=8.4.80, trackers/track.py on_predict_postprocess_end) calls tracker.update(det, img) on every
    frame, so the frame counter advances, unmatched tracks go lost / age out and the GMC's previous frame
    moves on; the frame writes no rows and the stabilizer registers it with no mask (extract.py:159,176-187);
  * when the tracker returns nothing the raw detections are kept with ids None (written as -1 and dropped
    later, extract.py:161-165, 287);
  * the stabilizer's foreground mask is the box set that is written out (tracker boxes, else raw boxes).
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import queue
import threading
import time
import weakref
from dataclasses import dataclass

import numpy as np

from . import _lib
from .detector import Detector
from .geometry import warp_boxes
from .stabilizer import Stabilizer
from .tracker import Tracker


@dataclass
class FrameResult:
    index: int                       # position in feeding order (0 = reference frame)
    xyxy: np.ndarray                 # [n,4] boxes written out (tracker posterior, or raw detections when ids is None)
    conf: np.ndarray
    cls: np.ndarray
    ids: np.ndarray | None           # [n] track ids, None when the tracker returned nothing
    xywh: np.ndarray | None          # the same boxes as centre/size float32, None when the frame has none
    xywh_stab: np.ndarray | None     # stabilized boxes (None when there are no boxes or stabilization is off)
    H: np.ndarray | None             # 3x3 f64 current -> reference, None for the reference frame / when no model was found
    n_det: int = 0                   # raw detections of the frame
    det_ms: float = 0.0              # detector GPU time of the frame's batch divided by its size
    gmc: np.ndarray | None = None    # 2x3 camera-motion warp the GMC found for the frame (engines built with gmc=True)
    stab_ms: float = 0.0             # GPU time of the frame's stabilizer pass (0 for the reference frame / no stabilizer)
    H_fallback: bool = False         # True when registration failed and H is the last known transform


def xyxy_to_xywh(b: np.ndarray) -> np.ndarray | None:
    if len(b) == 0:
        return None
    return np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float32)


def parse_prio(text: str | None) -> tuple[int, int]:
    """GTX_ENGINE_PRIO = "<detectors>,<stabilizers + GMC>" (1 highest, 0 default, -1 lowest); one value sets both."""
    vals = [v.strip() for v in (text or "0,0").split(",") if v.strip()]
    try:
        nums = [int(v) for v in vals]
    except ValueError:
        nums = []
    if len(nums) == 1:
        nums = nums * 2
    if len(nums) != 2 or any(v not in (-1, 0, 1) for v in nums):
        raise ValueError(f"GTX_ENGINE_PRIO must be one or two of -1, 0, 1 separated by a comma (detectors, stabilizers), got {text!r}")
    return nums[0], nums[1]


def queue_of_streams(tokens: list[str], n_queues: int = 4) -> list[int]:
    """Hardware queue (0-based) every stream of a creation sequence lands on, by the HIP runtime's rule as measured on MI355X
    (tools/stream_map_probe.py, profiles/r04_stream_map.txt): a stream gets its place when it is created; the first
    `n_queues` (GPU_MAX_HW_QUEUES, 4) streams open a queue each, a later one joins the queue that carries the fewest
    streams, the highest-numbered one among equals; the null stream ('n') counts like any other from its first use."""
    load, out = [], []
    for _ in tokens:
        if len(load) < n_queues:
            load.append(0)
            q = len(load) - 1
        else:
            q = max(range(n_queues), key=lambda i: (-load[i], i))
        load[q] += 1
        out.append(q)
    return out


def plan_stream_order(n_dets: int, n_stab: int, n_queues: int = 4) -> list[str]:
    """The creation order of an engine's streams: d = detector, s = stabilizer, f = the read-ahead feeder's copy stream,
    g = the GMC, n = the null stream, x = a stream that is never used.

    Streams that share a hardware queue run in order, so a detector that shares one waits behind the other stream's
    kernels and barrier packets and the two detector streams stop overlapping: with both detectors on ONE queue the
    extract loop fed by the feeder reads 780 frames/s on MI355X, with a queue per detector 940 (profiles/r04_stream_plan.txt).
    The order built here gives every detector a queue that it shares only with idle streams ('n', 'x') and spreads
    stabilizers, feeder and GMC over the remaining queues; it ends so that the next two streams created (the GMC's second
    stream, made by the library; the sharded run's priming stream) join those too. With more than two detector streams there are not enough
    queues for that and the order is simply detectors-first."""
    if n_dets >= n_queues - 1:
        return ["d", "n"] + ["d"] * (n_dets - 1) + ["s"] * n_stab + ["f", "g"]
    busy = ["s"] * n_stab + ["f", "g"]
    order, dets_left, null_left = [], n_dets, True

    def landing():
        return queue_of_streams(order + ["?"], n_queues)[-1]

    while busy or dets_left:
        q = landing()
        if q < n_dets:                                       # a detector's queue: the detector itself, then only idle streams
            holds = [t for t, qq in zip(order, queue_of_streams(order, n_queues)) if qq == q]
            if "d" not in holds and dets_left:
                order.append("d")
                dets_left -= 1
            elif null_left:
                order.append("n")
                null_left = False
            else:
                order.append("x")
        elif busy:
            order.append(busy.pop(0))
        else:
            order.append("x")
    if null_left:
        order.append("n")
    while any(q < n_dets for q in queue_of_streams(order + ["?", "?"], n_queues)[-2:]):   # level the queues: the next two streams
        order.append("x")                                      # anyone creates (the GMC's second, a priming stream) join the busy ones
    return order


class PacedSource:
    """Wraps an iterable of batches that arrive at their own pace (a live stream: frames exist when the camera delivers them).
    ExtractEngine.run() then works for latency: a batch's results leave before stage 1 blocks on the next batch, and the
    stabilizer stage hands out what it holds while nothing arrives. A source has to SAY so: a slow synchronous reader (an image
    folder, a decoder, GTX_FEEDER=0) also keeps `next()` waiting, and for it the throughput schedule -- the read of batch k + 1
    under the GPU pass of batch k -- is the right one (an earlier version guessed from one 20 ms pull and flipped on page-cache stalls)."""
    paced = True

    def __init__(self, batches):
        self._it = batches

    def __iter__(self):
        return iter(self._it)


class StreamPlan:
    """The HIP streams of the extract engine on one device: created ONCE per process, all together, in the order
    plan_stream_order() gives, and handed out by role to every engine built afterwards.

    Once, because the queue a stream lands on depends on every stream the process has created and destroyed before
    (queue_of_streams): an engine that created its own streams got a different, usually worse, mapping for the second video
    of a process than for the first. All together, because the library's set-up copies bring the null stream into being
    at a moment of their own; here it is opened at its place in the order (gtx_device_open_null_stream)."""
    _plans: dict = {}
    _lock = threading.Lock()

    def __init__(self, device: int, order: list[str], p_det: int = 0, p_stab: int = 0):
        self.device, self.order, self.p_det, self.p_stab = device, list(order), p_det, p_stab
        self.ctxs = []                                           # [role, Context, weakref of the holder | None]
        for tok in self.order:
            if tok == "n":
                _lib.check(_lib.load().gtx_device_open_null_stream(device))
            else:
                self.ctxs.append([tok, _lib.Context(device, p_det if tok == "d" else p_stab if tok in ("s", "g") else 0), None])

    @classmethod
    def get(cls, device: int, n_dets: int, n_stab: int) -> "StreamPlan":
        """The device's plan; the first call lays it out for (n_dets, n_stab) -- GTX_ENGINE_ORDER (comma-separated tokens)
        overrides the order -- later calls reuse it and take() adds what it lacks."""
        with cls._lock:
            if device not in cls._plans:
                env = os.environ.get("GTX_ENGINE_ORDER")
                try:
                    n_queues = max(int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), 1)      # the runtime's own variable; 4 is its default
                except ValueError:
                    n_queues = 4
                order = [t.strip() for t in env.split(",") if t.strip()] if env else plan_stream_order(n_dets, n_stab, n_queues)
                if any(t not in ("d", "s", "f", "g", "n", "x") for t in order):
                    raise ValueError(f"GTX_ENGINE_ORDER: tokens are d, s, f, g, n, x; got {env!r}")
                p_det, p_stab = parse_prio(os.environ.get("GTX_ENGINE_PRIO"))
                cls._plans[device] = StreamPlan(device, order, p_det, p_stab)
                cls._plans[device].verify()
            return cls._plans[device]

    def overlap(self, a: _lib.Context, b: _lib.Context, spin_us: float = 1000.0) -> tuple[float, float]:
        """(ms one idle wave spins on a's stream, ms the same on both streams at once): gtx_streams_overlap."""
        one, two = C.c_float(), C.c_float()
        _lib.check(_lib.load().gtx_streams_overlap(a.handle, b.handle, float(spin_us), C.byref(one), C.byref(two)))
        return float(one.value), float(two.value)

    def verify(self) -> None:
        """The layout above follows a MEASURED rule of the HIP runtime (queue_of_streams), not a documented one. Once per process
        and device: do the detector streams really run at the same time? If two of them share a hardware queue (a runtime that
        places streams differently), the later one is replaced by the first of up to eight fresh streams that does overlap with
        the first detector stream, and a warning says so -- 15-20 % of the frame rate depend on it (profiles/r04_stream_plan.txt).
        GTX_ENGINE_VERIFY=0 skips the check (1 ms spins, best of three, ~10 ms per pair: well above a loaded host's launch and sync jitter)."""
        self.verified = None
        if os.environ.get("GTX_ENGINE_VERIFY", "1") == "0":
            return
        dets = [e for e in self.ctxs if e[0] == "d"]
        if len(dets) < 2:
            return
        import logging

        log = logging.getLogger(__name__)
        self.verified = True
        for ent in dets[1:]:
            one, two = self.overlap(dets[0][1], ent[1])
            log.debug(f"stream plan: device {self.device}: one spin {one:.2f} ms, two at once {two:.2f} ms")
            if two < 1.5 * one:
                continue
            one, two = self.overlap(dets[0][1], ent[1])          # a loaded host can fake one verdict (launch / sync jitter): it has to repeat
            if two < 1.5 * one:
                continue
            self.verified = False
            spares = []
            for _ in range(8):
                cand = _lib.Context(self.device, self.p_det)
                o1, o2 = self.overlap(dets[0][1], cand)
                if o2 < 1.5 * o1 and all(self.overlap(other[1], cand)[1] < 1.5 * o1 for other in dets[1:] if other is not ent):
                    log.warning(f"stream plan: two detector streams of device {self.device} ran one after the other ({two:.2f} ms for two "
                                f"{one:.2f} ms kernels): this HIP runtime does not place streams the way the plan assumes; "
                                "a replacement stream that does overlap was found and is used instead")
                    spares.append(ent[1])
                    ent[1] = cand
                    break
                spares.append(cand)
            else:
                log.warning(f"stream plan: two detector streams of device {self.device} share a hardware queue ({two:.2f} ms for two {one:.2f} ms "
                            "kernels) and no replacement stream overlaps: expect 15-20 % lower frame rates (GPU_MAX_HW_QUEUES, GTX_ENGINE_ORDER)")
            self.ctxs.extend(["x", c, True] for c in spares)     # kept alive and never handed out: destroying them would move later streams

    def take(self, role: str, holder=None) -> _lib.Context:
        """A context of `role` nobody holds (`holder`: the object it is for; when that object is gone the context is free
        again without give_back). A role the plan has run out of gets a new stream, wherever the runtime puts it."""
        with self._lock:
            for ent in self.ctxs:
                if ent[0] == role and (ent[2] is None or (ent[2] is not True and ent[2]() is None)):
                    ent[2] = weakref.ref(holder) if holder is not None else True
                    return ent[1]
            ctx = _lib.Context(self.device, self.p_det if role == "d" else self.p_stab if role in ("s", "g") else 0)
            self.ctxs.append([role, ctx, weakref.ref(holder) if holder is not None else True])
            return ctx

    def give_back(self, ctx: _lib.Context) -> None:
        with self._lock:
            for ent in self.ctxs:
                if ent[1] is ctx:
                    ent[2] = None


class ExtractEngine:
    def __init__(self, weights: dict, frame_hw: tuple[int, int], det_kw: dict, tracker: Tracker | None, stab_kw: dict | None, *,
                 device: int | None = None, batch: int = 2, det_streams: int = 2, stab_streams: int = 4, gmc: bool | str = False,
                 detectors: list[Detector] | None = None, feeder_stream: bool = False):
        """det_kw: Detector keywords (imgsz, conf, iou, max_det, classes, agnostic_nms, half, rect). tracker None: raw
        detections pass through (ids None; the frame-sharded bench tracks later on rank 0). stab_kw None: no
        stabilization. `detectors`: already-built Detector objects to adopt (same weights, own contexts). `feeder_stream`: also
        take `self.feeder_ctx`, the context a read-ahead feeder's transfers run on (geotrax_amd.feeder.FrameFeeder(ctx=...)),
        from the stream plan."""
        self.device = _lib.default_device() if device is None else device
        self.frame_hw = (int(frame_hw[0]), int(frame_hw[1]))
        self.B = max(int(batch), 1)
        self.dets = list(detectors or [])
        n_dets = max(int(det_streams), len(self.dets), 1)
        n_stab = max(1, min(int(stab_streams), 4 * n_dets * self.B - 2)) if stab_kw is not None else 0   # frames in flight < gray ring lifetime
        self.tracker = tracker
        if tracker is not None and getattr(tracker, "with_reid", False):   # BoT-SORT `with_reid: true, model: auto`: a vector per box
            det_kw = dict(det_kw, obj_feats=True)
            if any(not getattr(d, "obj_feats", False) for d in self.dets):
                raise ValueError("the tracker associates on appearance vectors (with_reid): adopted detectors must be built with obj_feats=True")
        self.gmc = None
        self.stabs = []
        # Every stream comes from the device's StreamPlan: which streams share a hardware queue is fixed once per process
        # (each detector a queue of its own, stabilizers + feeder + GMC spread over the others), not by the order this
        # constructor happens to build things in or by what the process created before.
        self.plan = StreamPlan.get(self.device, n_dets, n_stab)
        self._taken = []                 # the plan's contexts this engine holds (handed back by close())

        def take(role):
            ctx = self.plan.take(role, holder=self)
            self._taken.append(ctx)
            return ctx

        while len(self.dets) < n_dets:
            self.dets.append(Detector(weights, self.frame_hw, max_batch=self.B, ctx=take("d"), **det_kw))   # det_kw carries obj_feats when the tracker asks
        while len(self.stabs) < n_stab:
            self.stabs.append(Stabilizer(self.frame_hw, ctx=take("s"), **stab_kw))
        self.feeder_ctx = take("f") if feeder_stream else None   # the context a read-ahead feeder's transfers run on
        if gmc:                          # True / "sparseOptFlow": the GPU Lucas-Kanade GMC; "orb" / "sift": the feature-based ones (gmc.FeatureGMC); "ecc": gmc.EccGMC
            from .gmc import make_gmc

            self.gmc = make_gmc(self.frame_hw, method=gmc if isinstance(gmc, str) else "sparseOptFlow", ctx=take("g"))
        self.use_dev_gray = bool(self.stabs) and float(stab_kw.get("downsample_ratio", 0.5)) == 0.5
        self._stage = {}                 # per detector: device staging buffer for host frames
        self._host_frames = {}           # frames kept for the host-gray fallback of the stabilizer
        self._index, self._have_ref = 0, False
        self._last_H = None              # last valid current->reference transform, in frame order
        self._prof = self.prof = None    # GTX_ENGINE_PROF=1: see run()
        self._gmc_sub = self._gmc_col = 0   # frames queued on the GMC stream / warps taken (one writer thread each)

    _IDLE = object()                 # drain() -> _stabilized(): no frame arrived for a few milliseconds

    #: host threads the engine adds to the caller's: the detector stage and the GMC + tracker stage (the stabilizer stage runs on
    #: the thread that iterates run()). All three wait on HIP events with hipEventBlockingSync, i.e. asleep while the GPU works:
    #: 8 ranks x 3 threads + rank 0's replay thread fit the 16 host cores a GPU box grants without spinning against each other.
    host_threads = 2

    # ---- lifecycle
    def set_reference(self, frame: np.ndarray) -> None:
        """Registers every stabilizer against `frame` (host BGR) ahead of run(): the frames fed afterwards are all
        registered against it. The foreground mask is the frame's raw detections."""
        if hasattr(frame, "bgr"):                               # a Yuv420Frame of a .y4m source
            frame = frame.bgr()
        d = self.dets[0].detect(np.ascontiguousarray(frame, np.uint8))
        g = self.dets[0].gray_dptr(0)
        boxes = d.xywh if len(d) else None
        for st in self.stabs:                                   # same reference image -> identical reference keypoints
            if self.use_dev_gray:
                st.set_ref_gray_dev(g[0], g[1], g[2], boxes)
            else:
                st.set_ref_frame(frame, boxes)
        self._have_ref = True

    def reset(self, keep_reference: bool = False) -> None:
        """Forget the tracks and the camera-motion state (and, unless told otherwise, the reference frame)."""
        self._index = 0
        self._last_H = None
        if not keep_reference:
            self._have_ref = False
        if self.tracker is not None:
            self.tracker.reset()
        if self.gmc is not None:
            while self._gmc_sub > self._gmc_col:                # a run that was abandoned half way
                self._gmc_collect()
            self.gmc.reset_params()

    def close(self) -> None:
        for d in self.dets:
            for p in [self._stage.pop(id(d), None), self._stage.pop(("yuv", id(d)), None)]:
                if p:
                    d.ctx.dev_free(p)
            d.close()
        for s in self.stabs:
            s.close()
        if self.gmc is not None:
            self.gmc.close()
        for c in self._taken:                                   # the streams stay alive for the next engine of the process
            self.plan.give_back(c)
        self.feeder_ctx = None
        self.dets, self.stabs, self.gmc, self._taken = [], [], None, []

    # ---- feeding
    def _submit(self, det: Detector, batch) -> int:
        if isinstance(batch, (int, np.integer)):                # device pointer to B contiguous frames
            self._gmc_frames(det, int(batch), self.B)
            det.submit_dev(int(batch), self.B)
            return self.B
        from .feeder import DeviceBatch
        from .frames import Yuv420Frame

        if isinstance(batch, DeviceBatch):                      # frames a read-ahead feeder is bringing into HBM: the detector's
            if not 1 <= batch.n <= self.B:                      # stream waits for their upload, this thread does not
                raise ValueError(f"a batch holds 1..{self.B} frames, got {batch.n}")
            if self.stabs and not self.use_dev_gray:
                raise ValueError("device batches need the stabilizer to work on the detector's gray image (downsample_ratio 0.5)")
            batch.wait_on(det.ctx)
            self._gmc_frames(det, batch.ptr, batch.n)
            det.submit_dev(batch.ptr, batch.n)
            return batch.n

        frames = [f if isinstance(f, Yuv420Frame) else np.ascontiguousarray(f, dtype=np.uint8) for f in batch]
        if not 1 <= len(frames) <= self.B:
            raise ValueError(f"a batch holds 1..{self.B} frames, got {len(frames)}")
        nbytes = self.frame_hw[0] * self.frame_hw[1] * 3
        key = id(det)
        if key not in self._stage:
            self._stage[key] = det.ctx.dev_alloc(nbytes * self.B)
        for i, f in enumerate(frames):
            if tuple(f.shape[:2]) != self.frame_hw:
                raise ValueError(f"frame is {f.shape[1]}x{f.shape[0]}, engine was built for {self.frame_hw[1]}x{self.frame_hw[0]}")
            if isinstance(f, Yuv420Frame):                      # I420 planes cross PCIe (half the bytes), BGR is made on the GPU
                ykey = ("yuv", key)
                if ykey not in self._stage:
                    self._stage[ykey] = det.ctx.dev_alloc(len(f.data))
                det.ctx.dev_upload(self._stage[ykey], f.data)
                _lib.check(det.ctx.lib.gtx_yuv420_to_bgr_dev(det.ctx.handle, C.c_void_p(self._stage[ykey]), f.h, f.w,
                                                             C.c_void_p(self._stage[key] + i * nbytes)))
            else:
                det.ctx.dev_upload(self._stage[key] + i * nbytes, f)
        self._gmc_frames(det, self._stage[key], len(frames))
        det.submit_dev(self._stage[key], len(frames))
        if self.stabs and not self.use_dev_gray:
            self._host_frames[key] = [f.bgr() if isinstance(f, Yuv420Frame) else f for f in frames]
        return len(frames)

    def _gmc_frames(self, det: Detector, ptr: int, n: int) -> None:
        """A GMC that works on the BGR frames (gmc_method ecc) gets every frame of a batch here, when the batch goes to its
        detector: the frame's image is prepared on the detector's stream, behind whatever brought the frame into HBM and AHEAD of
        the detector pass -- collect() of that pass therefore also covers it, and the next batch may overwrite the frame buffer.
        Submission order = clip order (batches go to the detectors in clip order)."""
        if self.gmc is None or not getattr(self.gmc, "wants_frames", False):
            return
        nbytes = self.frame_hw[0] * self.frame_hw[1] * 3
        for i in range(n):
            self.gmc.submit_frame_dev(ptr + i * nbytes, self.frame_hw[0], self.frame_hw[1], producer=det.ctx)
            self._gmc_sub += 1

    def run(self, batches, paced: bool | None = None):
        """paced: the source is a live stream (see PacedSource); None = what the source declares (`batches.paced`, default False).
        batches: iterable of device pointers (B contiguous BGR u8 frames in HBM), of feeder.DeviceBatch objects (<= B frames
        on their way into HBM) or of lists of <= B host frames.
        Yields one FrameResult per frame, in feeding order. An item may also be a pair (batch, prev): the batch does not
        continue the previous one (a shard rank of the frame-sharded run) and the GMC is primed with `prev`, the
        device pointer of the frame that precedes the batch in the clip (None: the batch opens the clip).

        With a tracker or stabilizers the loop runs as three host stages on their own threads, joined by bounded
        queues (the C-ABI calls release the GIL): detector submit/collect -> GMC + tracker (clip order) ->
        stabilizer submit/collect + box warp (this thread, which yields). Per-frame results are the same as
        frame at a time; only the host work overlaps. GTX_ENGINE_THREADS=0 keeps everything on the calling thread."""
        self._paced = bool(getattr(batches, "paced", False)) if paced is None else bool(paced)
        # GTX_ENGINE_PROF=1: seconds the host stages spend in their blocking calls, summed over the run (self.prof afterwards)
        self._prof = collections.defaultdict(float) if os.environ.get("GTX_ENGINE_PROF") == "1" else None
        self.prof = self._prof
        self.marks = [] if self._prof is not None else None     # (what, frame or pass, seconds since run()) of every stage boundary
        self._t_run = time.perf_counter()
        threaded = (self.tracker is not None or bool(self.stabs)) and os.environ.get("GTX_ENGINE_THREADS", "1") != "0"
        frames = self._tracked_frames_threaded(batches) if threaded else self._tracked_frames(batches)
        return self._stabilized(frames)

    # ---- stage 1: detector batches (submit keeps every detector stream busy; collect blocks on the oldest pass)
    def _detected_batches(self, batches):
        it = iter(batches)
        inflight = collections.deque()                          # (detector, frames in the batch)
        k = 0

        paced = self._paced                                      # a live stream (PacedSource): work for latency, not for throughput
        exhausted = False

        def submit_next():
            nonlocal k, exhausted
            b = next(it, None)
            if b is None:
                exhausted = True
                return
            prev = None
            if isinstance(b, tuple):                            # (batch, device pointer of the frame before it | None | False)
                b, prev = b
                prev = False if prev is None else prev          # None: the batch opens the clip -> restart without a frame
            det = self.dets[k % len(self.dets)]
            k += 1
            inflight.append((det, self._submit(det, b), prev))

        host_gray = bool(self.stabs) and not self.use_dev_gray
        try:
            def fill():                                             # a pass in flight on every detector stream -- unless the source is a stream
                while not exhausted and not paced and len(inflight) < len(self.dets):
                    submit_next()

            submit_next()
            fill()
            while inflight:
                det, nb, prev = inflight.popleft()
                t0 = time.perf_counter() if self._prof is not None else 0.0
                dets = det.collect()
                if self._prof is not None:
                    self._prof["det_collect"] += time.perf_counter() - t0
                    self.marks.append(("det", k - len(inflight) - 1, time.perf_counter() - self._t_run))
                grays = [det.gray_dptr(b) for b in range(nb)]
                hosts = self._host_frames.pop(id(det), None)
                det_ms = float(sum(dets[0].speed.values())) / nb if dets else 0.0
                late = host_gray or paced                           # paced source: the batch's results leave before the stage blocks on the next one
                if not late:
                    fill()                                      # keep the detectors busy while the host works on the batch
                n_skip = 0
                if self.gmc is not None and getattr(self.gmc, "wants_frames", False):
                    if prev is not None:
                        raise NotImplementedError("gmc_method ecc registers every frame against the first frame of the clip: not available to a frame-sharded run")
                elif self.gmc is not None:                      # the batch queues on the GMC stream now, results in order
                    restart = prev is not None                  # a shard rank's batch: it does not continue the previous one
                    if restart and prev is not False:           # the frame that precedes the batch in the clip, from HBM
                        self.gmc.submit_frame_dev(int(prev), self.frame_hw[0], self.frame_hw[1], restart=True)
                        self._gmc_sub += 1
                        n_skip, restart = 1, False
                    for g in grays:                             # every frame, detections or not (BOTSORT.update -> gmc.apply)
                        if restart:                             # no frame before it: the batch's first frame opens the sequence
                            self.gmc.reset_sequence()
                            restart = False
                        self.gmc.submit_gray_dev(*g)
                        self._gmc_sub += 1
                yield det, dets, grays, hosts, det_ms, n_skip
                if host_gray:
                    submit_next()
                elif late and not inflight and not exhausted:       # a stream: everything in flight has been handed on; now wait for the next batch
                    submit_next()
        finally:                                                # consumer stopped early or a stage failed: leave no pass in flight
            for det, *_ in inflight:
                try:
                    det.collect()
                except Exception:
                    pass
            self._host_frames.clear()

    # ---- stage 2: camera-motion compensation + tracker, strictly in clip order
    def _track_batch(self, det, dets, grays, hosts, det_ms, n_skip=0):
        # a generator: a frame goes on to the stabilizer stage while the tracker works on the next one of its batch
        for _ in range(n_skip):                                 # warp of the frame that only primed the GMC (identity)
            self._gmc_collect()
        for b, (d, g) in enumerate(zip(dets, grays)):
            ids = None
            xyxy, conf, cls = d.xyxy, d.conf, d.cls
            warp = self._gmc_collect() if self.gmc is not None else None
            if self.tracker is not None:                        # also on frames without detections (frame counter, lost/removed ageing)
                t0 = time.perf_counter() if self._prof is not None else 0.0
                t_xyxy, t_ids, t_score, t_cls, _ = self.tracker.update(d.xyxy, d.conf, d.cls, gmc=warp, feats=d.feats)
                if self._prof is not None:
                    self._prof["tracker"] += time.perf_counter() - t0
                    self.marks.append(("trk", self._index, time.perf_counter() - self._t_run))
                if len(t_ids):
                    xyxy, conf, cls, ids = t_xyxy, t_score, t_cls, t_ids
            r = FrameResult(self._index, xyxy, conf, cls, ids, xyxy_to_xywh(xyxy), None, None, len(d), det_ms, warp)
            self._index += 1
            yield r, det, g, hosts[b] if hosts is not None else None

    def _gmc_collect(self):
        warp = self.gmc.collect()
        self._gmc_col += 1
        return warp

    def _tracked_frames(self, batches):
        for item in self._detected_batches(batches):
            yield from self._track_batch(*item)

    def _tracked_frames_threaded(self, batches):
        # Queue depths. A frame's gray image lives in its detector's 16-deep ring (detector.hpp kGrayRing) until that
        # detector has started 15 more passes; by then stage 1 has handed on at least 14 * n_dets * B later frames, so
        # the frames between stage 1 and the stabilizer's collect must stay below that.
        n_batches = 2
        n_frames = 2 * self.B * len(self.dets)
        in_flight = (n_batches + 1) * self.B + n_frames + len(self.stabs) + 1
        assert in_flight <= 14 * len(self.dets) * self.B, "engine queues outlive the detector's gray ring"
        assert self.gmc is None or (n_batches + 2) * self.B <= 63, "engine queues outrun the GMC's 64-deep result ring"
        assert self.gmc is None or not getattr(self.gmc, "wants_frames", False) or (len(self.dets) + n_batches + 1) * self.B <= 31, \
            "engine queues outrun the ECC GMC's 32-deep frame ring"
        q_det = queue.Queue(maxsize=n_batches)                  # detected batches
        q_trk = queue.Queue(maxsize=n_frames)                   # tracked frames
        stop = threading.Event()
        END = object()
        IDLE = self._IDLE

        def put(q, item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def stage(src, q, work):
            try:
                for item in src:
                    for out in work(item):
                        if not put(q, out):
                            return
                put(q, END)
            except BaseException as e:                          # handed to the consumer, which re-raises it
                put(q, e)
            finally:
                src.close()                                     # runs the source's cleanup on this thread

        def drain(q, tell_idle=False):
            while True:
                try:
                    item = q.get(timeout=0.004 if tell_idle else 0.05)
                except queue.Empty:
                    if stop.is_set():                           # the consumer is gone and so may be the producer
                        return
                    if tell_idle:                               # nothing for 4 ms: a paced source; the consumer may take what is pending
                        yield IDLE
                    continue
                if item is END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item

        t1 = threading.Thread(target=stage, args=(self._detected_batches(batches), q_det, lambda item: (item,)), daemon=True)
        t2 = threading.Thread(target=stage, args=(drain(q_det), q_trk, lambda item: self._track_batch(*item)), daemon=True)
        t1.start()
        t2.start()
        try:
            yield from drain(q_trk, tell_idle=True)
        finally:
            stop.set()
            for q in (q_det, q_trk):                            # unblock producers stuck in put()
                try:
                    while True:
                        q.get_nowait()
                except queue.Empty:
                    pass
            t1.join()
            t2.join()

    # ---- stage 3: stabilizer submit/collect + box warp; results leave in frame order
    def _stabilized(self, frames):
        pending = collections.deque()                           # (stabilizer, partial FrameResult) awaiting collect

        def last_known(r):
            # stabilo keeps `trans_matrix_last_known`: when a frame cannot be registered (too few matches, no model)
            # the previous valid transform is used for its boxes and reported as its matrix. Results leave here in
            # frame order, so "last" is the previous frame's, whichever stabilizer object produced it.
            if r.H is None:
                if self._last_H is not None:
                    r.H, r.H_fallback = self._last_H.copy(), True
            else:
                self._last_H = r.H.copy()

        prof = self._prof

        def finish():
            st, r = pending.popleft()
            t0 = time.perf_counter() if prof is not None else 0.0
            st.collect()
            if prof is not None:
                prof["stab_collect"] += time.perf_counter() - t0
                self.marks.append(("stab", r.index, time.perf_counter() - self._t_run))
            r.H = st.get_cur_trans_matrix(raw=True)
            r.stab_ms = st.last_ms()
            last_known(r)
            if r.xywh is not None:
                r.xywh_stab = warp_boxes(r.H, r.xywh) if r.H is not None else r.xywh.copy()
            return r

        try:
            for item in frames:
                if item is self._IDLE:                              # the stages in front have nothing ready (a live stream between frames):
                    while pending:                                  # hand out what the stabilizers hold instead of waiting for frame t + 3
                        yield finish()
                    continue
                r, det, g, host = item
                if not self.stabs:
                    yield r
                    continue
                if not self._have_ref:                              # reference frame: boxes pass through, no transform row
                    det.ctx.synchronize()
                    for st in self.stabs:
                        if self.use_dev_gray:
                            st.set_ref_gray_dev(g[0], g[1], g[2], r.xywh)
                        else:
                            st.set_ref_frame(host, r.xywh)
                    self._have_ref = True
                    r.xywh_stab = None if r.xywh is None else r.xywh.copy()
                    yield r
                    continue
                if len(pending) == len(self.stabs):                 # results are taken in frame order
                    yield finish()
                st = self.stabs[r.index % len(self.stabs)]
                if self.use_dev_gray:
                    t0 = time.perf_counter() if prof is not None else 0.0
                    st.submit_gray_dev(g[0], g[1], g[2], r.xywh)
                    if prof is not None:
                        prof["stab_submit"] += time.perf_counter() - t0
                        self.marks.append(("sub", r.index, time.perf_counter() - self._t_run))
                        prof["frames"] += 1
                    pending.append((st, r))
                else:                                               # other downsample ratios: the stabilizer makes its own gray
                    st.stabilize(host, r.xywh)
                    r.H = st.get_cur_trans_matrix(raw=True)
                    last_known(r)
                    if r.xywh is not None:
                        r.xywh_stab = warp_boxes(r.H, r.xywh) if r.H is not None else r.xywh.copy()
                    yield r
            while pending:
                yield finish()
        finally:                                                # abandoned run: take what the stabilizers still owe
            while pending:
                try:
                    pending.popleft()[0].collect()
                except Exception:
                    pass
            frames.close()
