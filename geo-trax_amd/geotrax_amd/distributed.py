"""Frame-sharded extraction across the GPUs of one node.

Detection and stabilization of a frame depend only on that frame (and on the reference frame's
keypoints), the tracker is a sequential scan. So: one process per GPU; the clip is dealt to the ranks
in runs of consecutive frames, round-robin (shard_runs); every rank registers against the same
reference frame (it reads that one frame itself -- cheaper than a broadcast and bit-identical on
every rank); after every round the per-frame fixed-stride records of the round are gathered to rank 0
over ``torch.distributed`` (RCCL on GPUs, gloo in the CPU tests), where a second host thread replays
the tracker over them in clip order and warps the tracker boxes with each frame's H while the GPUs
work on the next round: no serial tail remains after the last frame but the last round itself.

The exchanges: once per video rank 0 reads the weight file and broadcasts it as one flat fp32 buffer
(broadcast_weights; BASELINE north star: "RCCL broadcast of weights"), the ranks agree on the source
(agree_on_source: it opens everywhere and has the same frame count), and once per round the records go
to rank 0 packed to their real length (wire_pack: ~3.3 KB per frame at 132 boxes instead of the 48 KB
fixed-stride record), behind one all-gather of the packed sizes that doubles as the failure flag.
Difference from the single-GPU ("exact") order, by construction: the
stabilizer mask of a frame is built from the raw detections instead of the tracker-output boxes
(extract.py:181 uses the latter), because the tracker has not run yet when a shard rank stabilizes.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_frames: int, rank: int, world: int, first: int = 0) -> tuple[int, int]:
    """Contiguous [start, stop) of rank `rank` over frames [first, n_frames)."""
    n = max(n_frames - first, 0)
    per, rem = divmod(n, world)
    start = first + rank * per + min(rank, rem)
    return start, start + per + (1 if rank < rem else 0)


def pack_frame_record(max_det: int, xyxy, conf, cls, H, gmc=None, with_gmc: bool = False) -> np.ndarray:
    """Fixed-stride float64 record: n, max_det x (x1,y1,x2,y2,conf,cls), [gmc valid, 2x3 camera-motion warp,]
    valid, h11..h33. The GMC block is present when with_gmc (BoT-SORT runs: the shard rank computes the warp, rank 0's
    tracker applies it)."""
    rec = np.zeros(1 + max_det * 6 + (7 if with_gmc else 0) + 10, dtype=np.float64)
    n = min(len(conf), max_det)
    rec[0] = n
    body = rec[1:1 + max_det * 6].reshape(max_det, 6)
    body[:n, :4], body[:n, 4], body[:n, 5] = xyxy[:n], conf[:n], cls[:n]
    if with_gmc and gmc is not None:
        rec[-17] = 1.0
        rec[-16:-10] = np.asarray(gmc, dtype=np.float64).reshape(6)
    if H is not None:
        rec[-10] = 1.0
        rec[-9:] = np.asarray(H, dtype=np.float64).reshape(9)
    return rec


def unpack_frame_gmc(rec: np.ndarray):
    """The 2x3 warp of a record packed with_gmc, or None when the rank had none for the frame."""
    return rec[-16:-10].reshape(2, 3).copy() if rec[-17] > 0 else None


def unpack_frame_record(rec: np.ndarray, max_det: int):
    n = int(rec[0])
    body = rec[1:1 + max_det * 6].reshape(max_det, 6)[:n]
    H = rec[-9:].reshape(3, 3).copy() if rec[-10] > 0 else None
    return body[:, :4].astype(np.float32), body[:, 4].astype(np.float32), body[:, 5].astype(np.int32), H


# ---- wire format of a round's records. In memory a frame's record is the fixed-stride float64 row pack_frame_record() makes
# (what gtx_tracker_replay consumes); on the wire it shrinks to what it holds:
#   int32 n | uint8 flags (1: H valid, 2: GMC valid) | 3 pad bytes | n x 6 float32 (x1 y1 x2 y2 conf cls) | [9 f64 H] | [6 f64 warp]
# Boxes, scores and classes are float32 / small integers at their source (detector.py), so float32 bodies lose nothing;
# the homography and the camera-motion warp stay float64 (they are written with %.16g).
def wire_pack(block: np.ndarray, max_det: int, with_gmc: bool = False) -> np.ndarray:
    """[frames, stride] float64 records -> one uint8 array."""
    parts = []
    for rec in np.asarray(block, dtype=np.float64):
        n = int(rec[0])
        h_ok, g_ok = rec[-10] > 0, bool(with_gmc) and rec[-17] > 0
        head = np.zeros(8, np.uint8)
        head[:4] = np.frombuffer(np.int32(n).tobytes(), np.uint8)
        head[4] = (1 if h_ok else 0) | (2 if g_ok else 0)
        parts.append(head)
        if n:
            parts.append(rec[1:1 + n * 6].astype(np.float32).view(np.uint8))
        if h_ok:
            parts.append(np.ascontiguousarray(rec[-9:]).view(np.uint8))
        if g_ok:
            parts.append(np.ascontiguousarray(rec[-16:-10]).view(np.uint8))
    return np.concatenate(parts) if parts else np.zeros(0, np.uint8)


def wire_unpack(buf: np.ndarray, frames: int, max_det: int, with_gmc: bool = False) -> np.ndarray:
    """Inverse of wire_pack: -> [frames, stride] float64 records."""
    stride = 1 + max_det * 6 + (7 if with_gmc else 0) + 10
    out = np.zeros((frames, stride), dtype=np.float64)
    raw = np.ascontiguousarray(buf, dtype=np.uint8).tobytes()
    o = 0
    for f in range(frames):
        n = int(np.frombuffer(raw, np.int32, 1, o)[0])
        flags = raw[o + 4]
        o += 8
        if not 0 <= n <= max_det:
            raise ValueError(f"wire record {f}: {n} boxes outside [0, {max_det}]")
        out[f, 0] = n
        if n:
            out[f, 1:1 + n * 6] = np.frombuffer(raw, np.float32, n * 6, o)
            o += n * 24
        if flags & 1:
            out[f, -10] = 1.0
            out[f, -9:] = np.frombuffer(raw, np.float64, 9, o)
            o += 72
        if flags & 2:
            out[f, -17] = 1.0
            out[f, -16:-10] = np.frombuffer(raw, np.float64, 6, o)
            o += 48
    if o != len(raw):
        raise ValueError(f"wire block holds {len(raw)} bytes, its {frames} records account for {o}")
    return out


def broadcast_weights(tensors: dict | None, names: dict | None, dist, device=None, error: str | None = None):
    """Rank 0 hands in the tensors it read from the weight file (and the class names); every rank gets them back: a small
    header through broadcast_object_list, then ONE flat fp32 buffer through dist.broadcast (44 MB for YOLOv8s; RCCL when the
    ranks own a GPU each). `error` (rank 0): the file could not be loaded -- every rank raises RuntimeError with that text."""
    import torch

    rank = dist.get_rank()
    head = [None]
    order = None
    if rank == 0:
        if error is None:
            order = sorted(tensors)
            head[0] = {"keys": order, "shapes": [tuple(int(v) for v in np.asarray(tensors[k]).shape) for k in order], "names": dict(names or {})}
        else:
            head[0] = {"error": str(error)}
    dist.broadcast_object_list(head, src=0)
    meta = head[0]
    if "error" in meta:
        raise RuntimeError(meta["error"])
    sizes = [int(np.prod(sh)) if len(sh) else 1 for sh in meta["shapes"]]
    if rank == 0:
        flat = torch.from_numpy(np.concatenate([np.asarray(tensors[k], np.float32).ravel() for k in order]))
    else:
        flat = torch.empty(sum(sizes), dtype=torch.float32)
    if device is not None:
        flat = flat.to(device)
    dist.broadcast(flat, src=0)
    if rank == 0:
        return tensors, dict(names or {})
    host, out, o = flat.cpu().numpy(), {}, 0
    for k, sh, n in zip(meta["keys"], meta["shapes"], sizes):
        out[k] = host[o:o + n].reshape(sh).copy()
        o += n
    return out, {int(k): str(v) for k, v in meta["names"].items()}


def agree_on_source(ok: bool, n_frames: int, dist, device=None) -> tuple[bool, bool, int]:
    """Before the first round: does the source open on EVERY rank, and with the same frame count? One all-gather of
    (ok, frames) per rank -> (all ok, counts agree, rank 0's count). A rank that cannot open its copy (a file system that is
    not shared, a flaky mount) must not leave the others waiting in the round's collectives."""
    import torch

    world = dist.get_world_size()
    mine = torch.tensor([1 if ok else 0, int(n_frames)], dtype=torch.int64)
    if device is not None:
        mine = mine.to(device)
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    rows = [[int(v) for v in g.cpu()] for g in got]
    all_ok = all(r[0] == 1 for r in rows)
    return all_ok, all_ok and len({r[1] for r in rows}) == 1, rows[0][1]


_replay_reservation: tuple[frozenset, int] | None = None   # (the mask the process started with, the core set aside): once per process


def reserve_replay_core(world: int) -> int | None:
    """At four ranks and more rank 0's tracker replay thread is busy most of the time next to 8 x 3 stage
    threads: keep one core for it. Called by every rank's main thread before it starts its engine: the calling thread (and
    every thread it creates afterwards) leaves the LAST core of the process's affinity mask alone; rank 0's replay thread
    then pins itself to that core (pin_to_core). Returns the core, or None when pinning is off (fewer than four ranks, fewer
    than four cores, GTX_PIN_REPLAY=0) -- GTX_PIN_REPLAY=1 forces it for tests.

    The reservation is made ONCE per process: `geotrax_amd.batch` runs every video of a folder in one process and calls this
    per video; a second call returns the core chosen the first time and never narrows the mask again (it used to drop one
    more core per video). release_replay_core() gives the original mask back."""
    import os

    global _replay_reservation
    mode = os.environ.get("GTX_PIN_REPLAY", "auto")
    if mode == "0" or (mode != "1" and world < 4) or not hasattr(os, "sched_getaffinity"):
        return None
    if _replay_reservation is not None:
        original, core = _replay_reservation
        try:                                                 # the caller may be another thread than the first time: same mask
            os.sched_setaffinity(0, set(original) - {core})
        except OSError:
            return None
        return core
    cores = sorted(os.sched_getaffinity(0))
    if len(cores) < 4:
        return None
    try:
        os.sched_setaffinity(0, set(cores[:-1]))
    except OSError:
        return None
    _replay_reservation = (frozenset(cores), cores[-1])
    return cores[-1]


def release_replay_core() -> None:
    """Undo reserve_replay_core() for the calling thread: the affinity mask the process started with."""
    import os

    global _replay_reservation
    if _replay_reservation is not None:
        try:
            os.sched_setaffinity(0, set(_replay_reservation[0]))
        except OSError:
            pass
        _replay_reservation = None


def pin_to_core(core: int | None) -> bool:
    import os

    if core is None:
        return False
    try:
        os.sched_setaffinity(0, {core})
        return True
    except OSError:
        return False


def init_process_group(local_device_count: int | None = None):
    """torch.distributed for a launcher-started run (RANK / WORLD_SIZE / MASTER_* in the environment). One process per
    GPU -> backend "nccl" (RCCL over xGMI) with collectives on device tensors; ranks that must share a GPU (tests on a
    one-GPU box), or GTX_DIST_BACKEND=gloo -> gloo with host tensors. Returns (dist, device for the collectives' tensors,
    local GPU index)."""
    import os

    import torch
    import torch.distributed as dist

    world, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count() if local_device_count is None else local_device_count
    backend = os.environ.get("GTX_DIST_BACKEND") or ("nccl" if n_dev >= min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))) and n_dev > 0 else "gloo")
    global _created_group
    if not dist.is_initialized():
        _created_group = True
        if backend == "nccl":
            # the engine's streams take their places on the hardware queues BEFORE torch / RCCL create streams of their own: the plan's
            # layout (engine.StreamPlan: a queue per detector) assumes it starts from an empty device; laid out for the default stream
            # counts of cfg/default.yaml, an engine configured otherwise adds what it lacks
            from .engine import StreamPlan

            StreamPlan.get(local, 2, 4)
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    return dist, dev, (local % max(n_dev, 1))


_created_group = False


def shutdown_process_group() -> None:
    """Destroys the process group if init_process_group() above created it (a launcher-started `geotrax_amd.extract` run ends
    with it; a caller that brought its own group keeps it)."""
    global _created_group
    if _created_group:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()
        _created_group = False


def gather_records(local: np.ndarray, failed: bool, dist=None, device=None, max_det: int | None = None, with_gmc: bool = False):
    """The data-path exchange of a round: every rank contributes its [per, stride] float64 block and a failure flag; rank 0
    gets (list of blocks by rank, any_failed), the others (None, any_failed). On the wire a block is packed to its real
    length (wire_pack); one all-gather of the packed sizes (-1 = this rank failed) tells every rank whether the video is
    void and how long the padded gather has to be. A rank that failed on its shard still takes part, so nobody waits for a
    timeout; every rank then voids the whole video like the reference does for any exception (extract.py:198-200)."""
    if dist is None:
        return [local], bool(failed)
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()
    if max_det is None:                                          # from the record layout: stride = 1 + 6 max_det [+ 7] + 10
        max_det = (local.shape[1] - 11 - (7 if with_gmc else 0)) // 6
    wire = wire_pack(local, max_det, with_gmc) if not failed else np.zeros(0, np.uint8)
    size = torch.tensor([-1 if failed else len(wire)], dtype=torch.int64)
    if device is not None:
        size = size.to(device)
    sizes = [torch.empty_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)                                 # every rank learns whether the video is void
    sizes = [int(t.item()) for t in sizes]
    any_failed = any(v < 0 for v in sizes)
    pad = max(max(sizes), 8)
    buf = np.zeros(pad, np.uint8)
    buf[:len(wire)] = wire
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    if rank != 0:
        return None, any_failed
    if any_failed:
        return [np.zeros_like(local) for _ in range(world)], True
    return [wire_unpack(b.cpu().numpy()[:sizes[r]], local.shape[0], max_det, with_gmc) for r, b in enumerate(bufs)], False


def shard_runs(n_frames: int, world: int, first: int = 0, run_frames: int | None = None) -> list[list[tuple[int, int]]]:
    """How frames [first, n_frames) are dealt to the ranks: rounds[k][r] = [start, stop) of rank r in round k.
    run_frames None: one round of contiguous ranges (shard_range). Otherwise the clip is cut into runs of run_frames
    consecutive frames dealt round-robin, `world` runs per round: after a round the records of world * run_frames consecutive
    frames can go to rank 0, whose tracker replays them while the GPUs work on the next round (no serial tail after the
    last frame has been detected). Runs keep frames consecutive so that BoT-SORT's GMC needs one priming frame per run."""
    if run_frames is None or world == 1:
        return [[shard_range(n_frames, r, world, first) for r in range(world)]]
    rounds, pos = [], first
    while pos < n_frames:
        row = []
        for _ in range(world):
            e = min(pos + run_frames, n_frames)
            row.append((pos, e))
            pos = e
        rounds.append(row)
    return rounds or [[(first, first)] * world]


class Replayer:
    """Rank 0: the sequential half of the loop over gathered records, fed in clip order, any number of frames at a time
    (extract.py:153-187 with the detector and stabilizer results taken from the records): tracker.update on every frame (with
    the rank's camera-motion warp for BoT-SORT) in one C call per feed (gtx_tracker_replay), ids -1 when the tracker returns
    nothing, boxes warped by the frame's H, the first frame passes through as the reference frame."""

    def __init__(self, first: int, tracker, warp_boxes, max_det: int, with_gmc: bool = False):
        self.first, self.tracker, self.warp_boxes, self.max_det, self.with_gmc = first, tracker, warp_boxes, max_det, with_gmc
        self.lists = ([], [], [], [], [], [], [])            # frame_arr, track_id, bbox, bbox_stab, class_id, conf, transforms
        self.last_H = None

    @staticmethod
    def _xywh(b):
        return np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float32)

    def feed(self, frames, recs) -> None:
        """frames: frame numbers (clip order, continuing the previous feed); recs: their packed records."""
        if len(frames) == 0:
            return
        frame_arr, track_id, bbox, bbox_stab, class_id, confs, transforms = self.lists
        per, t_xyxy, t_ids, t_score, t_cls, _ = self.tracker.replay(np.stack(recs), self.max_det, with_gmc=self.with_gmc)
        starts = np.concatenate([[0], np.cumsum(per)])
        for i, (f, rec) in enumerate(zip(frames, recs)):
            xyxy, conf, cls, H = unpack_frame_record(rec, self.max_det)
            a, b = int(starts[i]), int(starts[i + 1])
            bx, ids, sc, cl = t_xyxy[a:b], t_ids[a:b], t_score[a:b], t_cls[a:b]
            if f != self.first:
                if H is None:
                    H = self.last_H                              # stabilo's last known transform (see engine._stabilized)
                else:
                    self.last_H = H
                if H is not None:
                    transforms.append(np.hstack((np.array([[f]]), H.reshape(1, -1))))
            if len(conf) == 0:
                continue
            if len(ids) == 0:
                bx, ids, sc, cl = xyxy, np.full(len(conf), -1), conf, cls
            n = len(ids)
            xywh = self._xywh(bx)
            frame_arr.append(np.full((n, 1), f, dtype=np.uint32))
            track_id.append(np.asarray(ids).reshape(-1, 1).astype(np.uint16) if (np.asarray(ids) >= 0).all() else np.full((n, 1), -1))
            bbox.append(xywh)
            class_id.append(np.asarray(cl).astype(np.uint8).reshape(-1, 1))
            confs.append(np.asarray(sc, dtype=np.float32).reshape(-1, 1))
            bbox_stab.append(xywh if f == self.first or H is None else self.warp_boxes(H, xywh))


def replay_records(blocks, n_frames: int, first: int, world: int, tracker, warp_boxes, max_det: int, with_gmc: bool = False):
    """One-shot form (contiguous ranges, everything gathered): blocks[r][k] = record of rank r's k-th frame.
    -> (frame_arr, track_id, bbox, bbox_stab, class_id, conf, transforms)."""
    rp = Replayer(first, tracker, warp_boxes, max_det, with_gmc)
    frames, recs = [], []
    for r in range(world):
        s, e = shard_range(n_frames, r, world, first)
        frames += list(range(s, e))
        recs += [blocks[r][k] for k in range(e - s)]
    rp.feed(frames, recs)
    return rp.lists


def extract_sharded(n_frames: int, first: int, produce, tracker, warp_boxes, max_det: int, dist=None, device=None, with_gmc: bool = False,
                    run_frames: int | None = None, replay_core: int | None = None):
    """One video, frames [first, n_frames) sharded over the ranks of `dist` (None: one process) as shard_runs() deals them.
    `produce(runs)` yields this rank's packed records, frame by frame, for its list of [start, stop) runs in order
    (geotrax_amd.extract drives the HIP engine there, one pipeline across the runs; the CPU tests a deterministic stand-in).
    After every round each rank contributes its run's records to one gather (preceded by an all-reduce of a failure flag);
    rank 0 hands the round to a replay thread (tracker + box warps, in clip order) and goes on producing, so the tracker works
    while the GPUs do and nothing is left to replay serially after the last round but that round itself.
    Rank 0 returns the per-frame lists aggregate_results() expects, the other ranks None. If any rank fails, every rank raises
    RuntimeError after the collective of the round in which it failed (a failing rank still takes part, with zeros: nobody
    waits for a timeout)."""
    import queue
    import threading

    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    rounds = shard_runs(n_frames, world, first, run_frames)
    mine = [row[rank] for row in rounds]
    per = max(max(e - s for s, e in row) for row in rounds)      # records per rank and round, padded
    stride = 1 + max_det * 6 + (7 if with_gmc else 0) + 10
    failed, err = False, None
    gen = None
    replayer = Replayer(first, tracker, warp_boxes, max_det, with_gmc) if rank == 0 else None
    work: queue.Queue = queue.Queue()
    replay_err: list = []

    def replay_loop():
        pin_to_core(replay_core)                                 # reserve_replay_core(): the other threads of the job stay off it
        try:
            while True:
                item = work.get()
                if item is None:
                    return
                replayer.feed(*item)
        except BaseException as e:                               # surfaced on the main thread after the join
            replay_err.append(e)

    th = None
    if rank == 0:
        th = threading.Thread(target=replay_loop, name="gtx-replay", daemon=True)
        th.start()
    try:
        for k, row in enumerate(rounds):
            local = np.zeros((per, stride), dtype=np.float64)
            s, e = mine[k]
            if not failed and e > s:
                try:
                    if gen is None:
                        gen = iter(produce([r for r in mine if r[1] > r[0]]))
                    for i in range(e - s):
                        local[i] = next(gen)
                except (Exception, SystemExit) as ex:            # this rank's shard is lost (SystemExit: initialize_streams on an unreadable
                    failed, err = True, ex                       # source): tell the others through the collective
            blocks, any_failed = gather_records(local, failed, dist, device, max_det, with_gmc)
            if any_failed:
                raise RuntimeError(f"frame-sharded extraction failed on {'this rank: ' + repr(err) if failed else 'another rank'}")
            if rank == 0:
                frames, recs = [], []
                for r, (rs, re_) in enumerate(row):
                    frames += list(range(rs, re_))
                    recs += [blocks[r][i] for i in range(re_ - rs)]
                work.put((frames, recs))
    finally:
        if gen is not None and hasattr(gen, "close"):
            gen.close()
        if th is not None:
            work.put(None)
            th.join()
    if replay_err:
        raise replay_err[0]
    if rank != 0:
        return None
    return replayer.lists
