"""Frame-sharded extraction across the GPUs of one node.

Detection and stabilization of a frame depend only on that frame (and on the reference frame's
keypoints), the tracker is a sequential scan. So: one process per GPU, contiguous frame ranges
per rank, every rank registers against the same reference frame (it reads that one frame itself
-- cheaper than a broadcast and bit-identical on every rank), per-frame fixed-stride records are
gathered to rank 0 over ``torch.distributed`` (RCCL on GPUs, gloo in the CPU tests), and rank 0
replays the tracker over the frames in order and warps the tracker boxes with each frame's H.

The only exchange on the data path is that one gather of KB-sized records; weights are loaded by
every rank from the same file. Difference from the single-GPU ("exact") order, by construction: the
stabilizer mask of a frame is built from the raw detections instead of the tracker-output boxes
(extract.py:181 uses the latter), because the tracker has not run yet when a shard rank stabilizes.
"""
from __future__ import annotations

from typing import Callable

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


def extract_sharded(n_frames: int, first: int, read_frame: Callable[[int], np.ndarray],
                    detect: Callable[[np.ndarray], tuple], set_ref: Callable[[np.ndarray, np.ndarray | None], None],
                    stabilize: Callable[[np.ndarray, np.ndarray | None], np.ndarray | None], tracker, warp_boxes,
                    max_det: int, dist=None, device=None):
    """Runs this rank's shard and, on rank 0, returns the per-frame lists the aggregation step
    expects: (frame_arr, track_id, bbox, bbox_stab, class_id, conf, transforms). Other ranks return None.

    detect(frame) -> (xyxy [n,4], conf [n], cls [n]);  set_ref(frame, xywh|None);
    stabilize(frame, xywh|None) -> 3x3 or None;  tracker.update(xyxy, conf, cls) -> (xyxy, id, score, cls, idx);
    warp_boxes(H, xywh) -> xywh.  `dist` is torch.distributed (initialised) or None for one process."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1

    def xywh_of(b):
        return np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float32) \
            if len(b) else None

    ref = read_frame(first)
    rx, _, _ = detect(ref)
    set_ref(ref, xywh_of(rx))
    start, stop = shard_range(n_frames, rank, world, first)
    per = -(-max(n_frames - first, 0) // world)                  # records per rank, padded
    stride = 1 + max_det * 6 + 10
    local = np.zeros((per, stride), dtype=np.float64)
    for k, f in enumerate(range(start, stop)):
        frame = ref if f == first else read_frame(f)
        xyxy, conf, cls = detect(frame)
        H = None if f == first else stabilize(frame, xywh_of(xyxy))
        local[k] = pack_frame_record(max_det, xyxy, conf, cls, H)
    if dist is not None and world > 1:
        import torch

        t = torch.from_numpy(local)
        if device is not None:
            t = t.to(device)
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, bufs, dst=0)
        if rank != 0:
            return None
        allrec = [b.cpu().numpy() for b in bufs]
    else:
        allrec = [local]
    # ---- rank 0: sequential tracker over the frames in order
    frame_arr, track_id, bbox, bbox_stab, class_id, confs, transforms = [], [], [], [], [], [], []
    for r in range(world):
        s, e = shard_range(n_frames, r, world, first)
        for k, f in enumerate(range(s, e)):
            xyxy, conf, cls, H = unpack_frame_record(allrec[r][k], max_det)
            bx, ids, sc, cl, _ = tracker.update(xyxy, conf, cls)     # every frame, also without detections (ultralytics track.py)
            if len(conf) == 0:
                if H is not None:
                    transforms.append(np.hstack((np.array([[f]]), H.reshape(1, -1))))
                continue
            if len(ids) == 0:
                bx, ids, sc, cl = xyxy, np.full(len(conf), -1), conf, cls
            n = len(ids)
            xywh = xywh_of(bx)
            frame_arr.append(np.full((n, 1), f, dtype=np.uint32))
            track_id.append(np.asarray(ids).reshape(-1, 1).astype(np.uint16) if (np.asarray(ids) >= 0).all() else np.full((n, 1), -1))
            bbox.append(xywh)
            class_id.append(np.asarray(cl).astype(np.uint8).reshape(-1, 1))
            confs.append(np.asarray(sc, dtype=np.float32).reshape(-1, 1))
            if f == first:
                bbox_stab.append(xywh)
            else:
                bbox_stab.append(warp_boxes(H, xywh) if H is not None else xywh.copy())
                if H is not None:
                    transforms.append(np.hstack((np.array([[f]]), H.reshape(1, -1))))
    return frame_arr, track_id, bbox, bbox_stab, class_id, confs, transforms
