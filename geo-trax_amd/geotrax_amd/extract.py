#!/usr/bin/env python3
"""extract -- the per-frame extraction stage on MI355X.

Host-side restatement of the reference stage driver geotrax/extract.py (same function names,
argument meaning, log lines, error behaviour and output files):

    detect_track_stabilize   extract.py:114-131
    track_with_model         extract.py:134-214   the hot loop
    load_detector            extract.py:217-236
    initialize_streams       extract.py:239-257
    save_results             extract.py:487-523
    add_processing_args / parse_cli_args / main   extract.py:571-612

The loop body is the reference's; what sits underneath is this build's: ``YOLO.track`` runs the
HIP detector and the C++ tracker, ``Stabilizer`` runs the HIP keypoint/matching/RANSAC kernels on
the half-resolution gray image the detector's preprocess pass already produced on the GPU (so a
frame crosses PCIe once and is read from HBM once).

Usage:  python -m geotrax_amd.extract <source> [options]      (same flags as `geotrax extract`)
"""
from __future__ import annotations

import argparse
import datetime
import logging
import os
import sys
import time
from pathlib import Path

import numpy as np
import yaml

from . import __version__, tables
from .config_utils import backfill_args_from_config, load_config_all
from .frames import get_video_dimensions, open_source, source_exists
from .model import YOLO
from .postprocess import aggregate_results, postprocess_tracks

_INFERENCE_KEYS = {'conf', 'iou', 'imgsz', 'max_det', 'classes', 'augment', 'agnostic_nms', 'half', 'device',
                   'vid_stride', 'mode', 'task', 'stream_buffer', 'rect'}


def detect_track_stabilize(args: argparse.Namespace, logger: logging.Logger) -> None:
    """Process one video (extract.py:114-131)."""
    model = load_detector_sharded(args, logger) if frame_sharding_active() else load_detector(args, logger)
    config = load_config_all(args, logger, model_names=model.names if model else None)
    proc = config['main']['processing']
    out_cfg_raw = config['main'].get('output', {})
    backfill_args_from_config(args, {
        'cut_frame_left': proc['cut_frame_left'],
        'cut_frame_right': proc['cut_frame_right'],
        'interpolate': config['main']['extraction']['interpolate'],
        'output_folder': out_cfg_raw.get('folder', 'results'),
    })
    out_cfg = {**out_cfg_raw, 'folder': args.output_folder}
    if frame_sharding_active():
        res = track_with_model_sharded(model, config, logger)
        if res is None:                                     # ranks other than 0 have handed their records over
            return
        tracks, transforms = res
    elif pipelined(config):
        tracks, transforms = track_with_model(model, config, logger)
    else:
        tracks, transforms = track_with_model_blocking(model, config, logger)
    w_h = get_video_dimensions(config['main']['args'].source)
    tracks = postprocess_tracks(tracks, config, logger, w_h)
    save_results(tracks, transforms, config, logger, out_cfg)


class _Collector:
    """Per-frame result lists in the layout aggregate_results() expects (extract.py:142,160-187)."""

    def __init__(self):
        self.frame, self.ids, self.raw, self.stab, self.cls, self.conf, self.transforms = [], [], [], [], [], [], []

    def add_boxes(self, frame_num: int, boxes) -> np.ndarray | None:
        n = len(boxes)
        if n == 0:
            return None
        # the narrowings are part of the output contract: uint16 ids, uint8 classes, float32 boxes
        ids = np.full((n, 1), -1) if boxes.id is None else boxes.id.detach().numpy(force=True).astype(np.uint16).reshape(-1, 1)
        xywh = boxes.xywh.detach().numpy(force=True).astype(np.float32)
        self.frame.append(np.full((n, 1), frame_num, dtype=np.uint32))
        self.ids.append(ids)
        self.raw.append(xywh)
        self.cls.append(boxes.cls.detach().numpy(force=True).astype(np.uint8).reshape(-1, 1))
        self.conf.append(boxes.conf.detach().numpy(force=True).astype(np.float32).reshape(-1, 1))
        return xywh

    def add_transform(self, frame_num: int, H: np.ndarray | None) -> None:
        if H is not None:
            self.transforms.append(np.hstack((np.array([[frame_num]]), H.flatten().reshape(1, -1))))


def _frame_batches(reader, first: int, last, batch: int, frame_nums: list):
    """Frames first..last of the source in groups of `batch` (the skipped prefix still advances the counter,
    extract.py:147-151); appends the frame number of every yielded frame to `frame_nums`."""
    frame_num, group = 0, []
    while reader.isOpened():
        ok, frame = reader.read()
        if frame_num < first:
            frame_num += 1
            continue
        if not ok:
            break
        group.append(frame)
        frame_nums.append(frame_num)
        if len(group) == batch:
            yield group
            group = []
        if last is not None and frame_num >= last:
            break
        frame_num += 1
    if group:
        yield group


def _engine_kwargs(config: dict) -> tuple[dict, dict | None, dict]:
    ul = config['ultralytics']
    imgsz = ul.get('imgsz', 640)
    det_kw = dict(imgsz=int(max(imgsz) if isinstance(imgsz, (list, tuple)) else imgsz), conf=float(ul.get('conf') or 0.1),
                  iou=float(ul.get('iou', 0.7)), max_det=int(ul.get('max_det', 300)), classes=ul.get('classes'),
                  agnostic_nms=bool(ul.get('agnostic_nms', False)), half=bool(ul.get('half', False)), rect=bool(ul.get('rect', False)))   # absent -> the reference config's value (default.yaml:300)
    eng_cfg = config['main'].get('engine') or {}
    if eng_cfg.get('fp32_split') is not None:             # `engine: {fp32_split: false}` in the config: the exact-fp32 MFMA convolutions
        det_kw['fp32_split'] = bool(eng_cfg['fp32_split'])   # instead of split-f16x3 (half: false only; default: GTX_FP32_SPLIT or on)
    do_stab = config['main']['extraction']['stabilize']
    stab_kw = {k: v for k, v in config['stabilo'].items() if k not in ('gpu', 'viz', 'benchmark')} if do_stab else None
    return det_kw, stab_kw, eng_cfg


def frame_sharding_active() -> bool:
    """Under a launcher (WORLD_SIZE > 1) `geotrax_amd.extract` shards the frames of its one video over the ranks
    (SURVEY.md 8e / BASELINE north star); `geotrax_amd.batch` deals whole videos to the ranks instead and switches
    this off (GTX_FRAME_SHARDING=0)."""
    import os

    mode = os.environ.get("GTX_FRAME_SHARDING", "1")
    if mode == "force":                                   # the sharded path with whatever world size the launcher gave, even 1 (tests:
        return "RANK" in os.environ                       # RCCL's code path on a one-GPU box)
    return int(os.environ.get("WORLD_SIZE", "1")) > 1 and mode != "0"


def load_detector_sharded(args: argparse.Namespace, logger: logging.Logger) -> YOLO:
    """load_detector for a launcher-started, frame-sharded run: rank 0 reads the weight file, every other rank receives the
    tensors through one broadcast (geotrax_amd.distributed.broadcast_weights; RCCL when the ranks own a GPU each) -- the
    weight exchange BASELINE's north star names. A model that cannot be loaded ends every rank the way load_detector ends one
    process (error line, exit status 1)."""
    from . import distributed as D

    dist, dev, _ = D.init_process_group()
    model, err = None, None
    if dist.get_rank() == 0:
        try:
            model = load_detector(args, logger)
        except SystemExit:                                # load_detector has logged why
            err = "rank 0 could not load the detection model"
    try:
        tensors, names = D.broadcast_weights(model.tensors if model else None, model.names if model else None, dist, dev, error=err)
    except RuntimeError as e:
        if dist.get_rank() != 0:
            logger.error(f"Error loading the YOLOv8 model: {e}")
        sys.exit(1)
    if model is None:
        from .config_utils import load_config

        model = YOLO(tensors, task=load_config(getattr(args, 'cfg', None), logger).get('ultralytics', {}).get('task', 'detect'))
        model.names = names or model.names
    return model


def track_with_model_sharded(model: YOLO, config: dict, logger: logging.Logger) -> tuple[np.ndarray, np.ndarray] | None:
    """The hot loop with the frames of the video sharded over the ranks of the launcher (one process per GPU):
    runs of consecutive frames dealt round-robin (distributed.shard_runs), every rank registers against the same reference
    frame and detects + stabilizes its runs through one engine pipeline (foreground mask from the raw detections; BoT-SORT:
    the rank's GMC is primed with the frame before each run), fixed-stride records go to rank 0 in one gather per round
    (RCCL when every rank has its own GPU), and rank 0 replays the tracker in clip order on a second thread while the
    GPUs work on the next round. Rank 0 returns (tracks, transforms); the other ranks None. A failure on any rank voids the
    video on all of them (geotrax_amd.distributed.extract_sharded)."""
    from . import distributed as D
    from .engine import ExtractEngine
    from .geometry import warp_boxes

    args = config['main']['args']
    det_kw, stab_kw, eng_cfg = _engine_kwargs(config)
    do_stab = stab_kw is not None
    dist, dev, local = D.init_process_group()
    rank = dist.get_rank()
    first, last = args.cut_frame_left or 0, args.cut_frame_right
    tracker = model._make_tracker(config['ultralytics'].get('tracker', {'tracker_type': 'botsort'}))   # also decides whether a GMC runs
    if getattr(tracker, 'with_reid', False):
        raise NotImplementedError("the frame-sharded run carries boxes and warps between ranks, not appearance vectors: "
                                  "`with_reid: true` needs the single-process run (one GPU per video)")
    with_gmc = model._gmc_method is not None
    if model._gmc_method in ('orb', 'sift', 'ecc'):
        raise NotImplementedError(f"gmc_method '{model._gmc_method}': the frame-sharded run primes every rank's GMC with the frame before its batch, which only 'sparseOptFlow' takes "
                                  "('ecc' registers every frame against the first frame of the clip); run unsharded")
    max_det = det_kw['max_det']
    state = {}

    def produce(runs):
        """This rank's records, frame by frame, for its [start, stop) runs in order: ONE engine pipeline across the runs; a run
        opens with the clip's frame before it as the GMC's priming frame (BoT-SORT), batches never span two runs."""
        reader = initialize_streams(config['main'], config['ultralytics']['imgsz'], logger)
        state['reader'] = reader
        n_total = reader.frame_count if last is None else min(reader.frame_count, last + 1)
        assert n_total == state['n_frames']
        engine = ExtractEngine(model.tensors, reader.frame_hw, det_kw, None, stab_kw, device=local, batch=int(eng_cfg.get('batch', 2)),
                               det_streams=int(eng_cfg.get('det_streams', 2)), stab_streams=int(eng_cfg.get('stab_streams', 4)), gmc=with_gmc,
                               feeder_stream=os.environ.get("GTX_FEEDER", "1") != "0" and eng_cfg.get('read_ahead', True) is not False)
        state['engine'] = engine
        seekable = hasattr(reader, 'seek')
        cursor = {'pos': 0, 'last': None}                    # sequential sources: next frame read() returns, and the one before it
        fail_at = os.environ.get("GTX_TEST_FAIL_AT_FRAME")   # fault injection for the failure-path test

        def frame_at(f):
            """Frame f of the clip (f never decreases between calls except for the reference / priming frames of a seekable source)."""
            if seekable:
                if cursor['pos'] != f:
                    reader.seek(f)
                ok, frame = reader.read()
                cursor['pos'] = f + 1
            else:
                if f == cursor['pos'] - 1 and cursor['last'] is not None:
                    return cursor['last']
                ok, frame = True, None
                while ok and cursor['pos'] <= f:
                    ok, frame = reader.read()
                    cursor['pos'] += 1
                cursor['last'] = frame
            if not ok or frame is None or (fail_at is not None and f == int(fail_at)):
                raise RuntimeError(f"frame {f} could not be read")
            return frame

        # Device copies of the priming frames: a ring, because the GMC stream reads a copy when the run's first batch has been
        # collected, not when it is handed over. Between those two moments stage 1 can hand over at most one batch per detector
        # stream (in flight) + the two batches of the stage-1 queue + the one stage 2 holds, and every run is at least one
        # batch: a slot is written again n_prime runs later, so n_prime = detectors + 4 covers it with one to spare. The
        # uploads go through a context of their own (dev_upload waits for ITS stream only, not for a detector's pass).
        prime = []
        n_prime = len(engine.dets) + 4

        def prime_ptr(frame):
            if 'prime' not in state:
                state['prime'] = (engine.plan.take("p", holder=engine), prime)   # a stream beside the plan's: it joins a stabilizer queue, not a detector's
            pctx = state['prime'][0]
            if len(prime) < n_prime:
                prime.append(pctx.dev_alloc(frame.nbytes))   # (.nbytes of a Yuv420Frame is that of its BGR frame)
            p = prime[state.get('prime_i', 0) % n_prime]
            state['prime_i'] = state.get('prime_i', 0) + 1
            pctx.dev_upload(p, np.ascontiguousarray(frame.bgr() if hasattr(frame, "bgr") else frame, np.uint8))
            return p

        def batches():
            for start, stop in runs:
                prev_ptr = None
                if do_stab and not state.get('have_ref'):
                    engine.set_reference(frame_at(first))    # every rank registers against the clip's reference frame
                    state['have_ref'] = True
                if with_gmc and start > first:
                    prev_ptr = prime_ptr(frame_at(start - 1))
                group, opened = [], False
                for f in range(start, stop):
                    group.append(frame_at(f))
                    if len(group) == engine.B or f == stop - 1:
                        yield group if opened else (group, prev_ptr)    # a run does not continue the previous batch: (re)start the GMC
                        opened, group = True, []

        def fed_batches():
            """The same batches through read-ahead feeders (geotrax_amd.feeder): the rank's frames, run after run, are read and
            uploaded beside the pipeline instead of by frame_at() on the detector stage thread; the priming frames (BoT-SORT: the
            frame before each run) come through a second, one-frame feeder whose ring replaces the explicit one above."""
            from .feeder import FrameFeeder

            path, kind, offsets = reader.raw_layout()
            B, n_dets = engine.B, len(engine.dets)
            mine = [f for start, stop in runs for f in range(start, stop)]
            primes = [start - 1 for start, stop in runs if with_gmc and start > first]
            fd = FrameFeeder(reader.frame_hw, kind=kind, batch=B, ring=max(int(eng_cfg.get('read_ahead_batches', 3)), 1) + n_dets + 1, device=local,
                             ctx=engine.feeder_ctx)
            state['feeders'] = [fd]
            fd.open_file(path, offsets[mine], n_threads=int(eng_cfg.get('reader_threads', 3)))
            main_it = fd.batches(n_dets)
            prime_it = None
            if primes:
                pf = FrameFeeder(reader.frame_hw, kind=kind, batch=1, ring=n_prime + 1, device=local)
                state['feeders'].append(pf)
                pf.open_file(path, offsets[primes], n_threads=1)
                prime_it = pf.batches(n_prime)              # a slot is written again n_prime runs later, as with the ring above
            for start, stop in runs:
                prev_ptr = None
                if do_stab and not state.get('have_ref'):
                    engine.set_reference(frame_at(first))
                    state['have_ref'] = True
                if with_gmc and start > first:
                    pb = next(prime_it)
                    pb.wait_on(None)                        # long since resident: it was requested when the feeder opened
                    prev_ptr = pb.ptr
                for k in range(-(-(stop - start) // B)):
                    b = next(main_it)
                    yield b if k else (b, prev_ptr)

        layout_ok = hasattr(reader, 'raw_layout') and reader.raw_layout() is not None
        lengths_ok = all((stop - start) % engine.B == 0 for start, stop in runs[:-1])       # batches must not span two runs
        use_feeder = (layout_ok and lengths_ok and fail_at is None and os.environ.get("GTX_FEEDER", "1") != "0" and eng_cfg.get('read_ahead', True) is not False
                      and (not engine.stabs or engine.use_dev_gray))
        for r in engine.run(fed_batches() if use_feeder else batches()):
            yield D.pack_frame_record(max_det, r.xyxy, r.conf, r.cls, None if r.H_fallback else r.H, r.gmc, with_gmc=with_gmc)

    # The source must open on EVERY rank, with the same frame count, before anybody enters a round's collectives: a rank that
    # cannot open its copy would otherwise leave early while the others wait in the gather. The reference ends the process on an
    # unopenable source (extract.py:250-252); here every rank does, together.
    try:
        probe = initialize_streams(config['main'], config['ultralytics']['imgsz'], logger)
        n_here, ok = probe.frame_count, True
        probe.release()
    except SystemExit:
        n_here, ok = 0, False
    all_ok, same, _ = D.agree_on_source(ok, n_here, dist, dev)
    if not all_ok or not same:
        logger.error(f"Failed to open: '{args.source}' on every rank" + ("." if not all_ok else " with the same frame count."))
        sys.exit(1)
    replay_core = D.reserve_replay_core(dist.get_world_size())
    try:
        state['n_frames'] = n_here if last is None else min(n_here, last + 1)
        # frames per rank and round: whole batches, at most the config's `engine.shard_run_frames` (16), and no more than an even
        # share of the clip so that short clips still reach every rank (0 / null: one contiguous range per rank, one gather)
        run_frames = eng_cfg.get('shard_run_frames', 16)
        if run_frames:
            bsz = max(int(eng_cfg.get('batch', 2)), 1)
            share = -(-max(state['n_frames'] - first, 1) // dist.get_world_size())
            run_frames = max(bsz, min(-(-int(run_frames) // bsz) * bsz, -(-share // bsz) * bsz))
        lists = D.extract_sharded(state['n_frames'], first, produce, tracker, warp_boxes, max_det, dist=dist, device=dev, with_gmc=with_gmc,
                                  run_frames=run_frames or None, replay_core=replay_core)
    except (Exception, SystemExit) as e:                     # SystemExit: a source that stops opening between the probe and the run
        logger.error(f"Error processing: '{args.source}' due to: {e}")
        return (np.empty((0, 12), dtype=np.float32), np.empty((0, 10))) if rank == 0 else None
    finally:
        for fd_ in state.get('feeders', []):
            fd_.close()
        if 'prime' in state:
            for p_ in state['prime'][1]:
                state['prime'][0].dev_free(p_)
            state['engine'].plan.give_back(state['prime'][0])    # the stream stays for the next video of the process
        if 'reader' in state:
            state['reader'].release()
        if 'engine' in state:
            state['engine'].close()
    if rank != 0:
        return None
    frame_arr, track_id, bbox, bbox_stab, class_id, conf, transforms = lists
    if not do_stab:
        bbox_stab, transforms = [], []
    return aggregate_results(frame_arr, track_id, bbox, bbox_stab, class_id, conf, transforms, logger)


def _frames_in_range(reader, first: int, last):
    """Frames first..last of the source one by one (the skipped prefix is read and dropped, extract.py:147-151)."""
    for group in _frame_batches(reader, first, last, 1, []):
        yield group[0]


def _read_ahead_batches(reader, engine, eng_cfg: dict, first: int, last, frame_nums: list, logger: logging.Logger):
    """The frames of `reader` as feeder.DeviceBatch objects: read, uploaded (and colour-converted) beside the pipeline by
    geotrax_amd.feeder instead of on the detector stage thread. Returns (batches iterable, feeder) or (None, None) when the
    run needs host frames (a stabilizer that makes its own gray image) or `engine: {read_ahead: false}` / GTX_FEEDER=0."""
    from .feeder import FrameFeeder

    off = os.environ.get("GTX_FEEDER", "1") == "0" or eng_cfg.get('read_ahead', True) is False
    if off or (engine.stabs and not engine.use_dev_gray):
        return None, None
    layout = reader.raw_layout() if hasattr(reader, 'raw_layout') else None
    n_dets = len(engine.dets)
    ring = max(int(eng_cfg.get('read_ahead_batches', 3)), 1) + n_dets + 1
    stop = reader.frame_count if last is None else min(reader.frame_count, last + 1)
    if layout is not None:
        path, kind, offsets = layout
        feeder = FrameFeeder(reader.frame_hw, kind=kind, batch=engine.B, ring=ring, device=engine.device, ctx=engine.feeder_ctx)
        feeder.open_file(path, offsets[first:stop], n_threads=int(eng_cfg.get('reader_threads', 3)))
        frame_nums.extend(range(first, max(stop, first)))
    else:
        from .frames import Y4mReader

        feeder = FrameFeeder(reader.frame_hw, kind="i420" if isinstance(reader, Y4mReader) else "bgr", batch=engine.B, ring=ring,
                             device=engine.device, ctx=engine.feeder_ctx)

        def numbered():
            for k, f in enumerate(_frames_in_range(reader, first, last)):
                frame_nums.append(first + k)
                yield f

        feeder.open_reader(numbered())
    return feeder.batches(n_dets), feeder


def track_with_model(model: YOLO, config: dict, logger: logging.Logger) -> tuple[np.ndarray, np.ndarray]:
    """The hot loop (extract.py:134-214): read -> detect+track -> stabilize, through the pipelined engine
    (geotrax_amd.engine: batches on the detector streams, tracker in clip order, stabilizers on their own
    streams; per-frame results identical to the frame-at-a-time order); the frames come through the read-ahead
    feeder (geotrax_amd.feeder), so the stage threads never wait for the file. Any exception voids the whole
    video (empty tables), exactly like the reference (:198-200)."""
    from .engine import ExtractEngine

    args = config['main']['args']
    do_stab = config['main']['extraction']['stabilize']
    ul = config['ultralytics']
    det_kw, stab_kw, eng_cfg = _engine_kwargs(config)
    first, last = args.cut_frame_left or 0, args.cut_frame_right
    out, frame_nums, det_ms, stab_ms, n_frames = _Collector(), [], [], [], 0
    engine = feeder = None
    reader = initialize_streams(config['main'], config['ultralytics']['imgsz'], logger)   # exits on a missing / unopenable source, like the reference
    t_wall = time.time()
    try:
        tracker = model._make_tracker(ul.get('tracker', {'tracker_type': 'botsort'}))
        engine = ExtractEngine(model.tensors, reader.frame_hw, det_kw, tracker, stab_kw, batch=int(eng_cfg.get('batch', 2)),
                               det_streams=int(eng_cfg.get('det_streams', 2)), stab_streams=int(eng_cfg.get('stab_streams', 4)),
                               gmc=model._gmc_method or False,          # the method's name: sparseOptFlow (GPU LK), orb / sift (gmc.FeatureGMC)
                               feeder_stream=os.environ.get("GTX_FEEDER", "1") != "0" and eng_cfg.get('read_ahead', True) is not False)
        model._det = engine.dets[0]                        # introspection (names, gray) keeps working on the model object
        t_engine = time.time()
        batches, feeder = _read_ahead_batches(reader, engine, eng_cfg, first, last, frame_nums, logger)
        if batches is None:
            batches = _frame_batches(reader, first, last, engine.B, frame_nums)
        t_loop = time.time()
        for r in engine.run(batches):
            frame_num = frame_nums[r.index]
            n_frames += 1
            det_ms.append(r.det_ms)
            if r.xywh is not None:
                n = len(r.xywh)
                # the narrowings are part of the output contract: uint16 ids, uint8 classes, float32 boxes
                out.frame.append(np.full((n, 1), frame_num, dtype=np.uint32))
                out.ids.append(np.full((n, 1), -1) if r.ids is None else np.asarray(r.ids).astype(np.uint16).reshape(-1, 1))
                out.raw.append(r.xywh.astype(np.float32))
                out.cls.append(np.asarray(r.cls).astype(np.uint8).reshape(-1, 1))
                out.conf.append(np.asarray(r.conf).astype(np.float32).reshape(-1, 1))
                if do_stab:
                    out.stab.append(r.xywh_stab)
            if do_stab and r.index > 0:
                out.add_transform(frame_num, r.H)
                stab_ms.append(r.stab_ms)
    except Exception as e:
        logger.error(f"Error processing: '{args.source}' due to: {e}")
        return np.empty((0, 12), dtype=np.float32), np.empty((0, 10))
    else:
        if n_frames:
            # the reference's three averages (extract.py:205-207), same wording. Its loop is blocking, so its pipeline
            # figure is 1000 n / (sum yolo + sum stab); here the stages overlap on the GPU: the first two lines are the
            # per-frame GPU times of the detector pass and of the stabilizer pass, the third line is that same formula
            # (what the reference would print for these stage times) followed by what the run really delivered, file to result
            t_end = time.time()
            wall, loop = t_end - t_wall, t_end - t_loop
            logger.info(f"Average YOLOv8 (preprocess + inference + postprocess) time: {sum(det_ms) / len(det_ms):5.1f}ms.")
            if stab_ms:
                logger.info(f"Average stabilization time: {sum(stab_ms) / len(stab_ms):5.1f}ms")
            logger.info(f"Average pipeline time: {1000 * len(det_ms) / (sum(det_ms) + sum(stab_ms)):4.1f}fps. "
                        f"(wall clock, stages overlapped: {n_frames / loop:4.1f}fps over {n_frames} frames; "
                        f"{n_frames / wall:4.1f}fps with the {wall - loop:.2f}s of engine set-up)")
            model.last_run = dict(frames=n_frames, wall_s=wall, wall_fps=n_frames / wall, loop_s=loop, loop_fps=n_frames / loop,
                                  engine_setup_s=t_engine - t_wall, reader_setup_s=t_loop - t_engine,
                                  reference_convention_fps=1000 * len(det_ms) / (sum(det_ms) + sum(stab_ms)),
                                  det_ms=sum(det_ms) / len(det_ms), stab_ms=(sum(stab_ms) / len(stab_ms)) if stab_ms else None,
                                  read_ahead=feeder is not None)
    finally:
        if feeder is not None:
            feeder.close()
        reader.release()
        if engine is not None:
            model._det = None
            engine.close()
    return aggregate_results(out.frame, out.ids, out.raw, out.stab, out.cls, out.conf, out.transforms, logger)


def track_with_model_blocking(model: YOLO, config: dict, logger: logging.Logger) -> tuple[np.ndarray, np.ndarray]:
    """The reference's loop as it stands (extract.py:134-214), one frame at a time on the calling thread through the two
    drop-in objects of INTEGRATION.md section 1: ``model.track(frame, **config['ultralytics'], persist=True)`` and
    ``Stabilizer(**config['stabilo'])``. This is what a maintainer gets who only swaps the two imports in the reference's
    own extract.py; `engine: {pipelined: false}` in the config (or GTX_ENGINE=blocking) selects it here. Its tables are the
    pipelined engine's byte for byte (tests/test_dropin_gpu.py); it is several times slower, every call waits for the GPU."""
    from .stabilizer import Stabilizer

    args = config['main']['args']
    do_stab = config['main']['extraction']['stabilize']
    first, last = args.cut_frame_left or 0, args.cut_frame_right
    reader = initialize_streams(config['main'], config['ultralytics']['imgsz'], logger)
    stabilizer = Stabilizer(**config['stabilo'])
    eng_cfg = config['main'].get('engine') or {}
    if eng_cfg.get('fp32_split') is not None:
        model.fp32_split = bool(eng_cfg['fp32_split'])
    out, frame_num, yolo_time, stab_time = _Collector(), 0, [], []
    try:
        while reader.isOpened():
            success, frame = reader.read()
            if frame_num < first:
                frame_num += 1
                continue
            if not success:
                break
            results = model.track(frame, **config['ultralytics'], persist=True)
            boxes = results[0].boxes
            yolo_time.append(sum(results[0].speed.values()))
            xywh = out.add_boxes(frame_num, boxes)                # None for a frame without boxes
            if do_stab:
                start_time = time.time()
                if frame_num == first:
                    stabilizer.set_ref_frame(frame, xywh)
                    if xywh is not None:
                        out.stab.append(xywh)
                else:
                    stabilizer.stabilize(frame, xywh)
                    if xywh is not None:
                        out.stab.append(stabilizer.transform_cur_boxes())
                    out.add_transform(frame_num, stabilizer.get_cur_trans_matrix())
                stab_time.append(1000 * (time.time() - start_time))
            if last is not None and frame_num >= last:
                break
            frame_num += 1
    except Exception as e:
        logger.error(f"Error processing: '{args.source}' due to: {e}")
        return np.empty((0, 12), dtype=np.float32), np.empty((0, 10))
    else:
        if yolo_time:
            logger.info(f"Average YOLOv8 (preprocess + inference + postprocess) time: {sum(yolo_time) / len(yolo_time):5.1f}ms.")
            if stab_time:
                logger.info(f"Average stabilization time: {sum(stab_time) / len(stab_time):5.1f}ms")
            logger.info(f"Average pipeline time: {1000 * len(yolo_time) / (sum(yolo_time) + sum(stab_time)):4.1f}fps.")
    finally:
        reader.release()
        stabilizer.close()
    return aggregate_results(out.frame, out.ids, out.raw, out.stab, out.cls, out.conf, out.transforms, logger)


def pipelined(config: dict) -> bool:
    """`engine: {pipelined: false}` / GTX_ENGINE=blocking: the frame-at-a-time loop instead of the engine."""
    eng_cfg = config['main'].get('engine') or {}
    if str((config.get('stabilo') or {}).get('detector_name', 'orb')) in ('sift', 'rsift') and config['main']['extraction'].get('stabilize', True):
        return False                        # those detectors register host frames one at a time (stabilizer.py): the blocking loop
    return eng_cfg.get('pipelined', True) is not False and os.environ.get("GTX_ENGINE", "") != "blocking"


def load_detector(args: argparse.Namespace, logger: logging.Logger) -> YOLO:
    """extract.py:217-236. The model reference comes from --model or cfg -> extraction -> model."""
    from .config_utils import load_config

    raw = getattr(args, 'model', None)
    if isinstance(raw, list):
        raw = ' '.join(raw)
    cfg = load_config(getattr(args, 'cfg', None), logger)
    ref = raw or cfg.get('extraction', {}).get('model') or cfg.get('ultralytics', {}).get('model')
    if not ref:
        logger.critical("No detection model configured: set cfg -> extraction -> model or pass --model <file.safetensors>.")
        sys.exit(1)
    try:
        if str(ref).startswith('synthetic:'):
            from .weights import synthetic_yolov8

            seed = int(str(ref).split(':', 1)[1] or 0)
            model = YOLO(synthetic_yolov8(seed=seed, nc=4), task=cfg.get('ultralytics', {}).get('task', 'detect'))
        else:
            model = YOLO(model=ref, task=cfg.get('ultralytics', {}).get('task', 'detect'))
    except Exception as e:
        logger.error(f"Error loading the YOLOv8 model: {e}")
        sys.exit(1)
    logger.info(f"Detection model '{ref}' loaded successfully.")
    return model


def initialize_streams(config: dict, imgsz: int, logger: logging.Logger):
    """Open the frame source (extract.py:239-257; no progress bar here)."""
    source = config['args'].source
    if not source_exists(source):
        logger.critical(f"Video file '{source}' not found.")
        sys.exit(1)
    try:
        reader = open_source(source)
    except Exception as e:                                  # a damaged header, a container without a decoder: the source cannot be
        logger.error(f"Failed to open: '{source}' ({e}).")  # opened -- same exit as cv2's isOpened() == False (extract.py:250-252)
        sys.exit(1)
    if not reader.isOpened():
        logger.error(f"Failed to open: '{source}'.")
        sys.exit(1)
    return reader


def get_output_dir(source: Path, out_cfg: dict) -> Path:
    """file_utils.get_output_dir (file_utils.py:31-40): absolute folder as-is, else next to the video."""
    folder = Path(out_cfg.get('folder', 'results'))
    return folder if folder.is_absolute() else Path(source).parent / folder


def _serializable(o):
    if isinstance(o, dict):
        return {(_serializable(k) if not isinstance(k, (str, int, float, bool)) else k): _serializable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_serializable(v) for v in o]
    if isinstance(o, Path):
        return str(o)
    if isinstance(o, np.generic):
        return o.item()
    if isinstance(o, argparse.Namespace):
        return _serializable(vars(o))
    return o


def save_results(tracks: np.ndarray, transforms: np.ndarray, config: dict, logger: logging.Logger, out_cfg: dict) -> None:
    """<stem>.txt (%g), <stem>_vid_transf.txt (%.16g) and the run-metadata YAML (extract.py:487-523)."""
    source = Path(str(config['main']['args'].source))
    if str(source).startswith('synthetic:'):
        source = Path('synthetic.mp4')
    save_dir = get_output_dir(source, out_cfg)
    save_dir.mkdir(parents=True, exist_ok=True)
    tracks_txt_file = save_dir / f"{source.stem}{out_cfg.get('tracks_postfix', '')}.txt"
    transf_txt_file = save_dir / f"{source.stem}{out_cfg.get('stab_transform_postfix', '_vid_transf')}.txt"
    info_yaml_file = source.with_suffix('.yaml')

    try:
        if tracks.size != 0:
            tables.savetxt(tracks_txt_file, tracks, 6)             # np.savetxt(..., fmt='%g', delimiter=','), same bytes (tables.py)
            logger.info(f"Tracking results saved to: '{tracks_txt_file.resolve()}'")
    except Exception as e:
        logger.error(f"Failed to save the tracking results to: '{tracks_txt_file.resolve()}' due to: {e}")

    try:
        if transforms.size != 0 and config['main']['extraction']['save_stab']:
            frame_nums = transforms[:, 0].astype(int)
            matrices = transforms[:, 1:].reshape((-1, 3, 3))
            if not np.all(np.diff(frame_nums) == 1):
                logger.warning(f"Missing frame ids found in: '{transf_txt_file}'.")
            if not np.all(np.linalg.det(matrices) > 0):
                logger.warning(f"Invalid transforms found in: '{transf_txt_file}'.")
            tables.savetxt(transf_txt_file, transforms, 16)        # fmt='%.16g'
    except Exception as e:
        logger.error(f"Failed to save the video stabilization results to: '{transf_txt_file.resolve()}' due to: {e}")
    else:
        logger.info(f"Video stabilization results saved to: '{transf_txt_file.resolve()}'")

    metadata = _serializable(_build_run_metadata(config, save_dir))
    with open(info_yaml_file, 'w') as f:
        yaml.dump(metadata, f, default_flow_style=False, sort_keys=False)
    logger.info(f"Video info and configs saved to: '{info_yaml_file.resolve()}'")


def _build_run_metadata(config: dict, save_dir: Path) -> dict:
    """Same sections as the reference's metadata file (extract.py:526-568)."""
    main, ul, args = config['main'], config['ultralytics'], config['main']['args']
    active_classes = ul.get('classes') or []
    class_mapping = main.get('class_names', {})
    return {
        'run': {'geotrax_version': f'geotrax_amd-{__version__}', 'timestamp': datetime.datetime.now().isoformat(timespec='seconds'),
                'source': str(args.source), 'config': str(getattr(args, 'cfg', None)), 'output_folder': str(save_dir)},
        'model': {'configured': main.get('model_configured'), 'resolved': ul.get('model')},
        'class_names': {'source': main.get('class_names_source', 'unknown'),
                        'mapping': {k: class_mapping[k] for k in sorted(active_classes) if k in class_mapping}},
        'extraction': {k: v for k, v in main.get('extraction', {}).items() if k != 'model'},
        'processing': main.get('processing', {}),
        'output': main.get('output', {}),
        'detection': {k: v for k, v in ul.items() if k in _INFERENCE_KEYS},
        'tracker': {'active': main.get('tracker_active'), 'params': main.get('tracker_params', {})},
        'stabilo': config['stabilo'],
        'georef': config['georef'],
        'paths': {'ortho_folder': getattr(args, 'ortho_folder', None), 'master_folder': getattr(args, 'master_folder', None),
                  'segmentation_folder': getattr(args, 'segmentation_folder', None)},
        'visualization': main.get('visualization', {}),
        'plotting': main.get('plotting', {}),
        'batch': main.get('batch', {}),
    }


def add_common_args(group) -> None:
    """The shared flags of every geotrax stage (cli_utils.add_common_args :16-32)."""
    group.add_argument('--cfg', '-c', type=Path, default=None,
                       help='Pipeline config: a bundled preset name (default, confident, lenient, stable) or a path to a config file.')
    group.add_argument('--output-folder', '-of', type=str, default=None, help='Root folder for outputs.')
    group.add_argument('--log-path', '-lp', type=Path, default=None,
                       help='Where to write detailed logs: a directory (<stage>.log inside it) or a full file path. '
                            'Defaults to $XDG_STATE_HOME/geo-trax/logs (~/.local/state/geo-trax/logs).')
    group.add_argument('--verbose', '-v', action='store_true', help='Set print verbosity level to INFO.')


def add_processing_args(group) -> None:
    """Same flags, spelling and defaults as the reference (extract.py:571-584)."""
    group.add_argument('--model', '-m', nargs='+', default=None, metavar='MODEL')
    group.add_argument('--class-names', '-cn', nargs='+', default=None, metavar='ID=NAME|FILE')
    group.add_argument('--conf', '-co', type=float, default=None)
    group.add_argument('--classes', '-cls', nargs='+', type=int, default=None)
    group.add_argument('--cut-frame-left', '-cfl', type=int, default=None)
    group.add_argument('--cut-frame-right', '-cfr', type=int, default=None)
    group.add_argument('--interpolate', action=argparse.BooleanOptionalAction, default=None)


def parse_cli_args(argv=None) -> argparse.Namespace:
    parser = argparse.ArgumentParser(prog='geotrax extract', description='Vehicle Detection, Tracking, and Stabilization Pipeline')
    parser.add_argument('source', type=str, help='Path to the input video / frame source.')
    add_common_args(parser.add_argument_group('Optional arguments'))
    add_processing_args(parser.add_argument_group('Processing arguments'))
    return parser.parse_args(argv)


def default_log_dir() -> Path:
    """logging_utils.default_log_dir (:63-72), Linux branch."""
    import os

    return Path(os.environ.get('XDG_STATE_HOME') or (Path.home() / '.local' / 'state')) / 'geo-trax' / 'logs'


def setup_logger(name: str, verbose: bool = False, log_path=None, dry_run: bool = False) -> logging.Logger:
    """logging_utils.setup_logger (:75-110): console at WARNING (INFO with --verbose), plus a file handler at INFO.
    `log_path` may be a directory (<stage>.log inside it) or a full file path; default: default_log_dir()."""
    logger = logging.getLogger(name)
    logger.setLevel(logging.INFO)
    logger.propagate = False
    for h in list(logger.handlers):
        logger.removeHandler(h)
        h.close()
    fmt = logging.Formatter('%(asctime)s - %(levelname)s - %(name)s:%(module)s:%(funcName)s - %(message)s')
    console = logging.StreamHandler()
    console.setFormatter(fmt)
    console.setLevel(logging.INFO if verbose else logging.WARNING)
    logger.addHandler(console)
    if not dry_run:
        stage = f"{name.split('.')[-1]}.log"
        if log_path is None:
            file = default_log_dir() / stage
        else:
            log_path = Path(log_path)
            file = log_path / stage if log_path.is_dir() else log_path
        try:
            file.parent.mkdir(parents=True, exist_ok=True)
            fh = logging.FileHandler(file)
        except OSError as e:                                # a read-only home must not stop the run
            logger.warning(f"Cannot write the log file '{file}': {e}")
        else:
            fh.setFormatter(fmt)
            fh.setLevel(logging.INFO)
            logger.addHandler(fh)
            print(f"Saving logs to: {file}")
    return logger


def main(argv=None) -> None:
    args = parse_cli_args(argv)
    logger = setup_logger('geotrax_amd.extract', args.verbose, args.log_path)
    try:
        detect_track_stabilize(args, logger)
    finally:
        if frame_sharding_active():                      # the launcher-started run created a process group: leave it cleanly
            from . import distributed as D

            D.shutdown_process_group()


if __name__ == '__main__':
    main()
