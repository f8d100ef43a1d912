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
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    return dist, dev, (local % max(n_dev, 1))


def gather_records(local: np.ndarray, failed: bool, dist=None, device=None):
    """The one data-path collective of a frame-sharded video: every rank contributes its [per, stride] float64 block
    (padded to the same `per`) and a failure flag; rank 0 gets (list of blocks by rank, any_failed), the others
    (None, any_failed). A rank that failed on its shard still takes part (with zeros), so nobody waits for a
    timeout; rank 0 then voids the whole video like the reference does for any exception (extract.py:198-200)."""
    if dist is None or dist.get_world_size() == 1:
        return [local], bool(failed)
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()
    flag = torch.tensor([1.0 if failed else 0.0], dtype=torch.float64)
    t = torch.from_numpy(np.ascontiguousarray(local))
    if device is not None:
        flag, t = flag.to(device), t.to(device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)                  # every rank learns whether the video is void
    any_failed = bool(flag.item() > 0)
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    if rank != 0:
        return None, any_failed
    return [b.cpu().numpy() for b in bufs], any_failed


def replay_records(blocks, n_frames: int, first: int, world: int, tracker, warp_boxes, max_det: int, with_gmc: bool = False):
    """Rank 0: the sequential half of the loop over the gathered records, in clip order (extract.py:153-187 with the
    detector and stabilizer results taken from the records): tracker.update on every frame (with the rank's
    camera-motion warp for BoT-SORT), ids -1 when the tracker returns nothing, boxes warped by the frame's H, the first
    frame passes through as the reference frame. -> (frame_arr, track_id, bbox, bbox_stab, class_id, conf, transforms)."""
    def xywh_of(b):
        return np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float32)

    frame_arr, track_id, bbox, bbox_stab, class_id, confs, transforms = [], [], [], [], [], [], []
    last_H = None
    # the tracker sees every frame in clip order in one C call (gtx_tracker_replay); the bookkeeping below is per frame
    order = [blocks[r][k] for r in range(world) for k in range(shard_range(n_frames, r, world, first)[1] - shard_range(n_frames, r, world, first)[0])]
    per, t_xyxy, t_ids, t_score, t_cls, _ = tracker.replay(np.stack(order), max_det, with_gmc=with_gmc) if order else (np.zeros(0, np.int32),) + (np.zeros((0, 4), np.float32),) * 5
    starts = np.concatenate([[0], np.cumsum(per)]) if len(per) else np.zeros(1, np.int64)
    i = 0
    for r in range(world):
        s, e = shard_range(n_frames, r, world, first)
        for k, f in enumerate(range(s, e)):
            rec = blocks[r][k]
            xyxy, conf, cls, H = unpack_frame_record(rec, max_det)
            a, b = int(starts[i]), int(starts[i + 1])
            i += 1
            bx, ids, sc, cl = t_xyxy[a:b], t_ids[a:b], t_score[a:b], t_cls[a:b]
            if f != first:
                if H is None:
                    H = last_H                                   # stabilo's last known transform (see engine._stabilized)
                else:
                    last_H = H
                if H is not None:
                    transforms.append(np.hstack((np.array([[f]]), H.reshape(1, -1))))
            if len(conf) == 0:
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
            bbox_stab.append(xywh if f == first or H is None else warp_boxes(H, xywh))
    return frame_arr, track_id, bbox, bbox_stab, class_id, confs, transforms


def extract_sharded(n_frames: int, first: int, produce, tracker, warp_boxes, max_det: int, dist=None, device=None, with_gmc: bool = False):
    """One video, frames [first, n_frames) sharded in contiguous ranges over the ranks of `dist` (None: one process).
    `produce(start, stop)` yields this rank's packed records in frame order (geotrax_amd.extract drives the HIP engine
    there; the CPU tests a deterministic stand-in). Rank 0 returns the per-frame lists aggregate_results() expects, the
    other ranks None. If any rank fails, every rank raises RuntimeError after the (still completed) collective."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    start, stop = shard_range(n_frames, rank, world, first)
    per = -(-max(n_frames - first, 0) // world)                  # records per rank, padded
    stride = 1 + max_det * 6 + (7 if with_gmc else 0) + 10
    local = np.zeros((per, stride), dtype=np.float64)
    failed, err = False, None
    try:
        for k, rec in enumerate(produce(start, stop)):
            local[k] = rec
    except Exception as e:                                       # this rank's shard is lost: tell the others through the collective
        failed, err = True, e
    blocks, any_failed = gather_records(local, failed, dist, device)
    if any_failed:
        raise RuntimeError(f"frame-sharded extraction failed on {'this rank: ' + repr(err) if failed else 'another rank'}")
    if rank != 0:
        return None
    return replay_records(blocks, n_frames, first, world, tracker, warp_boxes, max_det, with_gmc)
