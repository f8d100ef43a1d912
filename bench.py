#!/usr/bin/env python3
"""bench.py -- 4K frames/s through the MI355X extraction hot path (detect + track + stabilize).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Either spelling gives an N-rank line: without a launcher and N > 1 the script starts the N ranks itself (a child
`torch.distributed.run`, before this process touches HIP) and forwards rank 0's line; fewer than N visible GPUs, or a launcher
whose WORLD_SIZE is not N, is an error (exit 2) -- never a smaller run printed as if it were the one asked for.

A step = one pass of the hot path over one batch of --batch consecutive 3840x2160 synthetic frames
per rank (default 2; --batch 1 is the frame-at-a-time pass), the frames already resident in HBM when
the timed region starts. Frames of a batch share the detector launches (every kernel covers the whole
batch); tracker and stabilizer take them one by one in clip order, so per-frame results do not depend
on the batch size (tests/test_detector_gpu.py::test_detector_batch_equals_single):

  N = 1   the reference's per-frame order (geotrax/extract.py:145-197): HIP detector -> host C++
          tracker -> HIP stabilizer (mask from the tracker's boxes) -> box warp.
  N > 1   the clip is dealt to the ranks in runs of --gather-every consecutive batches (contiguous
          ranges per rank): every rank detects and stabilizes its K batches (mask from the raw detections);
          every --gather-every steps the fixed-stride per-frame records of those steps are gathered to
          rank 0 over RCCL, where a second host thread runs the tracker over them in clip order and
          warps the boxes while the GPUs carry on -- all inside the timed region. No other collective.

Weak scaling: per-rank work is fixed, value = N*K frames / max-over-ranks time. ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os

os.environ.setdefault("OMP_NUM_THREADS", "1")   # torch is only plumbing here; 256 idle OpenMP spinners starve the host stages under a CPU quota
import queue
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "geo-trax_amd"))
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec; dense MFMA peaks ~2.5 PF fp16/bf16, 157.3 TF fp32
HBM_PEAK_GBS = 8000.0
# "f32s" = fp32 activations, convolutions as split-f16x3 (three fp16 MFMAs per product): a useful FLOP costs three
# fp16 MFMA FLOPs, so the roof for ALGORITHMIC FLOP/s is the dense fp16 peak / 3
MFMA_PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3, "f32s": 2500.0 / 3.0}
DTYPE_NAME = {"f16": "f16", "f32": "f32", "f32s": "f32 (split-f16x3 MFMA)"}
MODEL_LABEL = {"yolov8s": "YOLOv8s", "rtdetr-l": "RT-DETR-l (the reference's RTDETR branch, extract.py:222-225; frame stretched to imgsz x imgsz, no NMS)"}
# 2 x MAC over the 63 convolutions of YOLOv8s per frame at the two network inputs of a 3840 x 2160 frame
NOMINAL_GFLOP = {("yolov8s", 1920, 1920): 255.9, ("yolov8s", 1088, 1920): 145.0}
TRACKER_LABEL = {"bytetrack": "ByteTrack", "botsort": "BoT-SORT (GPU GMC)", "ocsort": "OC-SORT", "deepocsort": "Deep OC-SORT motion half (GPU GMC)",
                 "fasttrack": "FastTracker", "tracktrack": "TrackTrack (GPU GMC; association thresholds of the other trackers: the seeded boxes sit just above conf 0.25)"}
GMC_TRACKERS = ("botsort", "deepocsort", "tracktrack")   # trackers that take a camera-motion warp per frame (bench runs them with gmc_method: sparseOptFlow)
H, W = 2160, 3840
# seeded weights of the bench: only the stride-8 head fires, DFL biases give ~120 x 60 px boxes in 4K, and the class
# branch is spatially smooth so that candidates come in clusters and NMS suppresses about half of them
SYNTH_KW = dict(seed=0, nc=4, scale="s", level_bias=(0.0, -1e4, -1e4), box_weight_scale=0.002, smooth_cls=True, box_decay=(0.2, 0.3, 0.2, 0.3))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="extract", choices=["extract", "detect", "register", "georef", "warp", "extract+georef", "cli"],
                    help="extract = detect+track+stabilize (BASELINE metric / configs[2]); detect = YOLOv8s only (configs[1]); "
                         "register = RootSIFT registration of a 4K frame pair, the once-per-video step of configs[3] (single GPU); "
                         "georef = the per-row transform chain of configs[3] (frame px -> orthophoto px -> lat/lon -> local metres); "
                         "warp = perspective warp of resident 4K frames (visualize.py:285-289, SURVEY 8f N3); "
                         "extract+georef = BASELINE configs[3]: the extract stream followed by the georeference stage (registration against a synthetic "
                         "orthophoto, row chain, kinematics, CSV), unpaced and as a 30 fps stream; "
                         "cli = the product from a file: a 150-frame 3840x2160 clip written to local disk as .y4m and .npy, then "
                         "geotrax_amd.extract.track_with_model on the file (wall-clock frames/s and the reference's own convention)")
    ap.add_argument("--ortho", type=int, default=0,
                    help="--workload register: register a 3840x2160 frame against a synthetic orthophoto cut-out of this width (the reference's size is "
                         "15000, default.yaml:154, with max_features 250000) instead of a 4K frame pair")
    ap.add_argument("--cli-frames", type=int, default=150, help="--workload cli: frames of the clip (the reference's 5 s clip has 150)")
    ap.add_argument("--cli-formats", default="y4m,npy", help="--workload cli: which containers to write and measure")
    ap.add_argument("--cli-dir", default=None, help="--workload cli: where the clip is written (default: a fresh directory under the system's temp dir)")
    ap.add_argument("--cli-compare-sync", type=int, default=1, help="--workload cli: also run with the synchronous reader (GTX_FEEDER=0), for the comparison")
    ap.add_argument("--model", default="yolov8s", choices=["yolov8s", "rtdetr-l"],
                    help="detector graph of the seeded weights: yolov8s (BASELINE's model) or rtdetr-l (the reference's RTDETR branch, extract.py:222-225; SURVEY N4)")
    ap.add_argument("--half", type=int, default=0, help="ultralytics.half: 0 = fp32 activations (the reference default, default.yaml:245), 1 = fp16 activations + fp16 MFMA")
    ap.add_argument("--no-f16-line", action="store_true", help="skip the secondary fp16 measurement (N = 1, --half 0 runs add a shorter --half 1 pass and report it under 'f16')")
    ap.add_argument("--fp32", default=None, choices=["exact", "split"],
                    help="--half 0 only: exact = v_mfma_f32_32x32x2_f32; split = split-f16x3 (hi+lo fp16 operands, 3 fp16 MFMAs per product, "
                         "fp32 accumulate). Default: the library's default (geotrax_amd.detector.FP32_SPLIT_DEFAULT)")
    ap.add_argument("--rect", type=int, default=0, help="ultralytics.rect (reference config: false -> 1920x1920 input)")
    ap.add_argument("--imgsz", type=int, default=1920)
    ap.add_argument("--tracker", default=None, choices=["bytetrack", "botsort", "ocsort", "deepocsort", "fasttrack", "tracktrack"],
                    help="default: bytetrack at N = 1 (BASELINE configs[2]), botsort at N > 1 (configs[4]; the reference's own default, default.yaml:362)")
    ap.add_argument("--gmc-method", default="sparseOptFlow", choices=["sparseOptFlow", "orb", "sift", "ecc"],
                    help="camera-motion compensation of the trackers that take a warp (botsort, deepocsort, tracktrack): the reference's default "
                         "(default.yaml:374) or one of its other choices (N = 1 only: the frame-sharded run takes sparseOptFlow)")
    ap.add_argument("--batch", type=int, default=2, help="frames per detector pass (= per step)")
    ap.add_argument("--det-streams", type=int, default=2, help="detector instances (own HIP stream and activation buffers each) taking batches round-robin")
    ap.add_argument("--stab-streams", type=int, default=4, help="stabilizer instances (own HIP stream each) working on consecutive frames")
    ap.add_argument("--frames", type=int, default=6, help="distinct synthetic frames kept in HBM per rank (played ping-pong)")
    ap.add_argument("--detections", type=int, default=132, help="boxes per frame the seeded weights are calibrated to (golden clip: 132)")
    ap.add_argument("--candidates", type=int, default=0,
                    help="calibrate to this many anchors above conf on the probe frame instead of to --detections (SURVEY 8d: a trained YOLOv8 fires "
                         "1-3 k anchors in ~132 clusters; the seeded weights' boxes sit on their anchors, so clusters thin out ~2:1 only and 2 000 "
                         "candidates leave ~1 000 boxes -- the post-processing load of the secondary key `nms_load`)")
    ap.add_argument("--trace-every", type=int, default=8, help="HIP-event timing of every launch on every n-th detector pass inside the timed region (roofline); 0 = off")
    ap.add_argument("--gather-every", type=int, default=8, help="N > 1: steps between two gathers of per-frame records to rank 0")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo lets two ranks share one GPU to test the sharded path)")
    ap.add_argument("--sharding", default="frames", choices=["frames", "videos"],
                    help="N > 1: frames = batches of ONE clip dealt to the ranks, tracks gathered to rank 0 (BASELINE north star); "
                         "videos = every rank runs the whole reference-order pipeline on its own clip, no data-path collective (SURVEY 8e best case)")
    ap.add_argument("--host-frames", action="store_true", help="N = 1: feed the engine frames from host memory instead of frames resident in HBM (the PCIe-inclusive rate: a different measurement, labelled as such in the line, never the headline)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC summary instead of two rocprofv3 --pmc child passes of this run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, run one all_reduce over the chosen backend and print {launch_check, n_gpus, ranks_seen} -- no GPU work "
                         "(the CPU-side test of the --gpus N self-launch)")
    return ap.parse_args()


def visible_gpus() -> int:
    """Devices this process could use, WITHOUT initialising HIP (torch.cuda.device_count() only counts on this image): the
    self-launching parent must not touch the GPU, it starts other programs."""
    import torch

    return int(torch.cuda.device_count())


def self_launch(args) -> int | None:
    """`python bench.py --gpus N` with N > 1 and no launcher: start the N ranks ourselves, as a CHILD process
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>`),
    before anything in this process has touched HIP, forward the child's output (rank 0's one JSON line) and return its status.
    Fewer than N visible devices on the RCCL backend is an error, not a smaller run. Returns None when there is nothing to
    launch (N = 1, or already under a launcher)."""
    if "RANK" in os.environ or "LOCAL_RANK" in os.environ:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != args.gpus and os.environ.get("GTX_BENCH_FORCE_DIST") != "1":
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
            return 2
        return None
    if args.gpus <= 1:
        return None
    if args.workload not in ("extract", "detect"):
        print(f"bench.py: --workload {args.workload} is a single-GPU measurement (replicas only); run it with --gpus 1", file=sys.stderr)
        return 2
    if args.backend == "nccl":
        have = visible_gpus()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs for one rank per GPU over RCCL, this host shows {have} "
                  "(--backend gloo lets ranks share a GPU: a test of the sharded path, not a measurement)", file=sys.stderr)
            return 2
    import socket
    import subprocess

    with socket.socket() as sk:                              # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode          # stdout / stderr are inherited: rank 0's JSON line is the child's


def launch_check(args) -> int:
    """--launch-check: every rank joins the process group and adds 2**rank into one number."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    seen = 1
    if world > 1:
        if args.backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            t = torch.tensor([2 ** rank], dtype=torch.int64, device=f"cuda:{local}")
        else:
            dist.init_process_group("gloo")
            t = torch.tensor([2 ** rank], dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": bin(seen).count("1"), "backend": args.backend}), flush=True)
    return 0


def fp32_split(args):
    """None = the library default; only meaningful with --half 0."""
    return None if args.fp32 is None else args.fp32 == "split"


def xywh_of(b):
    return np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float32) \
        if len(b) else None


def calibrated_detector(ctx, frame, args, target):
    """Seeded weights know nothing about vehicles: shift the class-logit bias until one probe frame
    yields about `target` boxes after NMS (the golden clip has 127-136 per frame)."""
    from geotrax_amd.detector import Detector
    from geotrax_amd.weights import calibrate_cls_bias, synthetic_yolov8

    kw = dict(imgsz=args.imgsz, conf=0.25, iou=0.7, max_det=1000, classes=[0, 1, 2, 3], agnostic_nms=True,
              half=bool(args.half), fp32_split=fp32_split(args), rect=bool(args.rect), max_batch=max(args.batch, 1), ctx=ctx)
    if args.model == "rtdetr-l":            # seeded RT-DETR-l; the last score head shifted so that `target` of the 300 queries clear conf
        from geotrax_amd.weights import calibrate_rtdetr_scores, synthetic_rtdetr

        base = synthetic_rtdetr(seed=0, nc=4)
        det = Detector(base, (H, W), **kw)
        det.detect(frame)
        weights = calibrate_rtdetr_scores(base, det.raw_output(logits=True)[:, 4:], 0.25, min(target, 280))
        det.close()
        det = Detector(weights, (H, W), **kw)
        n_det = len(det.detect(frame))
        return det, weights, n_det, n_det
    base = synthetic_yolov8(**SYNTH_KW)   # vehicle-sized boxes from the stride-8 head, candidates in clusters
    det = Detector(base, (H, W), **kw)
    det.detect(frame)
    logits = det.raw_output(logits=True)[:, 4:]
    det.close()
    if getattr(args, "candidates", 0) > 0:                  # a given candidate count: one shift of the class bias, whatever is left after NMS
        weights = calibrate_cls_bias(base, logits, 0.25, args.candidates)
        det = Detector(weights, (H, W), **kw)
        n_det = len(det.detect(frame))
        return det, weights, n_det, int((det.raw_output()[:, 4:].max(1) > 0.25).sum())
    cand, weights, n_det, n_cand = 4 * target, base, 0, 0
    for _ in range(4):
        weights = calibrate_cls_bias(base, logits, 0.25, cand)
        det = Detector(weights, (H, W), **kw)
        n_det = len(det.detect(frame))
        n_cand = int((det.raw_output()[:, 4:].max(1) > 0.25).sum())
        if 0.85 * target <= n_det <= 1.15 * target:
            break
        det.close()
        cand = max(int(cand * target / max(n_det, 1)), 8)
    else:
        det = Detector(weights, (H, W), **kw)
    return det, weights, n_det, n_cand


def pmc_traffic(kernel, batch):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r01_pmc_traffic.json,
    written by tools/pmc_summary.py from two rocprofv3 --pmc runs of this command: FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for 16-B-per-lane reads on gfx950, plus WRITE_SIZE), or None when
    no summary for this batch size is present."""
    files = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))      # the newest round's collection
    try:
        f = files[-1]
        rec = json.loads(f.read_text())
        if rec.get("batch") != batch:
            return None
        return rec["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError, IndexError):
        return None


def live_traffic(kernel, args, B):
    """HBM bytes per launch of `kernel`, measured NOW: two child runs of this command under `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (separate passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes; the program itself
    directly behind `--`), the counter averaged over the kernel's dispatches, bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
    (gfx950 counts a 128-byte request as 64 in FETCH_SIZE). None when rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    sys.path.insert(0, str(ROOT / "tools"))
    from pmc_summary import family

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not Path(prof).exists():
        return None
    kb = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"gtx_pmc_{counter.lower()}_")
        cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, str(ROOT / "bench.py"), "--model", args.model, "--steps", "6", "--warmup", "2",
               "--no-cpu-baseline", "--no-profile", "--no-f16-line", "--workload", args.workload, "--tracker", args.tracker, "--batch", str(B), "--det-streams",
               str(args.det_streams), "--stab-streams", str(args.stab_streams), "--frames", str(args.frames), "--detections", str(args.detections),
               "--imgsz", str(args.imgsz), "--rect", str(args.rect)] + (["--fp32", args.fp32] if args.fp32 else [])
        try:
            subprocess.run(cmd, capture_output=True, text=True, timeout=150, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"})
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            n = v = 0.0
            for r in csv.DictReader(open(files[0])):
                if r["Counter_Name"] == counter and family(r["Kernel_Name"]) == kernel:
                    n += 1
                    v += float(r["Counter_Value"])
            kb[counter] = v / n if n else None
        except Exception:
            kb[counter] = None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if kb.get("FETCH_SIZE") is None or kb.get("WRITE_SIZE") is None:
        return None
    return round((2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0)


def host_cores() -> int:
    """Host cores this job may use: the cgroup quota when there is one (the GPU boxes give 16 of 256), else the affinity mask."""
    cores = len(os.sched_getaffinity(0))
    for f in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = Path(f).read_text().split()[:2]
            if quota != "max":
                cores = max(1, min(cores, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline(weights, ref_frame, frame, args):
    """The CPU restatement of the path on ONE 4K frame, timed on the host cores: torch-CPU YOLOv8s
    + numpy NMS (oracle/yolov8_ref.py), numpy ByteTrack (oracle/bytetrack_ref.py), and the C restatement of the stabilizer
    (oracle/stabilo_ref.c: ORB-style keypoints + Hamming matcher + RANSAC homography, OpenMP) -- with every stage on all the
    cores the job may use (`value`) and on one thread (`one_thread`). The reference frame's keypoints are prepared outside the
    timed region, as in steady state. Both conventions of BASELINE.md 2.3 are given: wall clock (detector + tracker +
    stabilizer) and the reference's 1000 n / (sum detect + sum stabilize)."""
    import torch
    from oracle import stabilo_c
    from oracle.bytetrack_ref import ByteTrackRef
    from oracle.yolov8_ref import YoloV8Ref, detect

    cores = host_cores()
    rt = args.model == "rtdetr-l"
    if rt:
        from oracle import rtdetr_ref

        model = rtdetr_ref.RtDetrRef(weights)
    else:
        model = YoloV8Ref(weights, emulate_half=False)
    stab_cfg = dict(downsample_ratio=0.5, max_features=2000, ref_multiplier=2.0, filter_ratio=0.9, ransac_threshold=2.0,
                    mask_use=True, mask_margin_ratio=0.15, fast_threshold=20, n_levels=8, scale_factor=1.2, seed=0)

    def detect_s(threads):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        out = rtdetr_ref.detect(model, frame, args.imgsz, 0.25, [0, 1, 2, 3], 1000) if rt else detect(model, frame, args.imgsz, bool(args.rect), 0.25, 0.7, [0, 1, 2, 3], True, 1000)
        return time.perf_counter() - t0, out

    t_det, (xyxy, conf, cls) = detect_s(cores)
    t_det1 = detect_s(1)[0] if cores > 1 else t_det
    torch.set_num_threads(cores)
    t_trk = t_stab = t_stab1 = 0.0
    if args.workload == "extract":
        trk = ByteTrackRef()
        t0 = time.perf_counter()
        rows = trk.update(xyxy, conf, cls)
        t_trk = time.perf_counter() - t0
        boxes = xywh_of(rows[:, :4]) if len(rows) else None
        st = stabilo_c.StabilizerC(stab_cfg, (H, W), n_hyp=2048)       # the hypothesis count the GPU stabilizer uses
        stabilo_c.set_threads(cores)
        st.set_ref_frame(ref_frame, None)
        t0 = time.perf_counter()
        st.stabilize(frame, boxes)
        t_stab = time.perf_counter() - t0
        stabilo_c.set_threads(1)
        t0 = time.perf_counter()
        st.stabilize(frame, boxes)
        t_stab1 = time.perf_counter() - t0
        stabilo_c.set_threads(cores)

    def both(td, ts):
        return {"value": 1.0 / (td + t_trk + ts), "reference_convention_fps": 1.0 / (td + ts), "stage_s": {"detect": td, "track": t_trk, "stabilize": ts}}

    allc, one = both(t_det, t_stab), both(t_det1, t_stab1)
    return dict(value=allc["value"], unit="frames/s", cores=cores, kind="port",
                reference_convention_fps=allc["reference_convention_fps"], stage_s=allc["stage_s"],
                one_thread=dict(value=one["value"], unit="frames/s", cores=1, reference_convention_fps=one["reference_convention_fps"], stage_s=one["stage_s"]),
                sample=f"1 synthetic 3840x2160 frame through the CPU restatement of the path ({args.workload}): fp32 torch-CPU detector, numpy tracker, "
                       f"C stabilizer (oracle/stabilo_ref.c, OpenMP; 2000 / 4000 features, 2048 hypotheses) -- every stage on {cores} threads (`value`) "
                       "and on 1 thread (`one_thread`); `value` = wall clock incl. tracker, `reference_convention_fps` = 1 / (detect + stabilize), extract.py:207")


def bench_register(args):
    """--workload register: one step = estimate_homography's GPU path on a 4K frame pair (host images in, H out).
    The roofline object prices the stage SURVEY 8d calls MFMA-bound, the brute-force 2-NN, at its stated size
    (250 000 x 250 000 x 128): the synthetic frames only give ~10 k keypoints each."""
    from geotrax_amd import _lib, ops
    from geotrax_amd.registration import register_once
    from geotrax_amd.synth import make_scene

    if args.ortho:
        return bench_register_ortho(args)
    ctx = _lib.Context(0)
    scene = make_scene(seed=0, h=H, w=W)
    a, b = scene.render(0, 150), scene.render(40, 150)
    kw = dict(max_features=250000, filter_ratio=0.55, ransac_epipolar_threshold=3.0, ransac_max_iter=10000, ransac_confidence=0.999999,
              rsift_eps=1e-8, ctx=ctx)
    steps, warm = min(args.steps, 40), min(args.warmup, 4)
    for _ in range(max(warm, 1)):
        Hm, stats, tm = register_once(b, a, **kw)
    t0 = time.perf_counter()
    for _ in range(steps):
        Hm, stats, tm = register_once(b, a, **kw)
    elapsed = time.perf_counter() - t0
    ys, xs = np.meshgrid(np.linspace(0, H - 1, 9), np.linspace(0, W - 1, 16), indexing="ij")
    P = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    pa, pb = Hm @ P, np.linalg.inv(scene.camera(40, 150)) @ P
    err = float(np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max())
    rng = np.random.default_rng(0)
    n = 250000
    d = rng.gamma(0.6, 1.0, (n, 128)).astype(np.float32)
    d /= d.sum(1, keepdims=True)
    d = np.sqrt(d)
    *_, ms = ops.match_2nn(d, d[rng.permutation(n)], iters=3, ctx=ctx)
    flops = 2.0 * n * n * 128
    out = {"metric": "4K frame-pair registrations/sec (RootSIFT + 2-NN + robust homography)", "value": steps / elapsed, "unit": "registrations/s",
           "n_gpus": 1, "steps": steps, "warmup": warm, "ms_per_step": 1000.0 * elapsed / steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32 keypoints/descriptors, f16 MFMA matching", "data": "synthetic",
           "config": {"workload": "estimate_homography GPU path on two 3840x2160 frames (frame 40 -> frame 0 of the synthetic clip), host images in",
                      "keypoints": [int(stats[0]), int(stats[1])], "good_matches": int(stats[2]), "inliers": int(stats[3]),
                      "stage_ms": {"detect_describe": float(tm[0]), "match": float(tm[1]), "ratio": float(tm[2]), "fit": float(tm[3])},
                      "max_grid_error_px_vs_known_camera": err},
           "roofline": {"bound": "mfma", "kernel": "match2nn_kernel", "achieved": flops / (ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS["f16"],
                        "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS["f16"], "traffic": None,
                        "avg_launch_us": 1000.0 * ms, "flops_per_launch": flops,
                        "timing": "HIP events around 3 passes of 250000 x 250000 x 128 (gtx_op_match_2nn), incl. the exact-distance finish kernel"}}
    print(json.dumps(out), flush=True)


def bench_register_ortho(args):
    """--workload register --ortho N: K11 at the reference's size (geotrax/utils/registration.py:21-95 with cfg/default.yaml:154,158-168:
    a 15 000-px orthophoto cut-out, max_features 250 000, ratio 0.55, 3 px, 10 000 iterations): one 3840x2160 frame registered against
    a synthetic N x N orthophoto whose frame -> orthophoto mapping is known. Reports keypoints found, time per stage, the HBM the
    registration holds, the 2-NN kernel's measured TFLOP/s at the keypoint counts reached, and a roofline object for the stage that
    takes the time at this size: the SIFT pyramid (HBM-bound)."""
    import ctypes as C

    from geotrax_amd import _lib, ops
    from geotrax_amd.registration import Sift, register_once
    from geotrax_amd.synth import make_scene

    ctx = _lib.Context(0)
    N = int(args.ortho)
    t0 = time.perf_counter()
    scene = make_scene(seed=0, h=H, w=W)
    frame = scene.render(0, 150)
    ortho, A = scene.orthophoto_large(size=N, scale=1.3, angle=0.2)
    t_make = time.perf_counter() - t0

    def mem():
        f, t = C.c_size_t(), C.c_size_t()
        _lib.check(_lib.load().gtx_device_mem_info(0, C.byref(f), C.byref(t)))
        return f.value, t.value

    free0, total = mem()
    # the detector stage of the orthophoto on its own: GPU time by stage (HIP events), then its buffers are given back
    sift = Sift((N, N), ctx=ctx)
    free_sift, _ = mem()
    sift.detect_and_compute(ortho, max_features=250000, cap=8)            # cap: rows copied back to the host (the count is what matters here)
    k_ortho = sift.detect_and_compute(ortho, max_features=250000, cap=8)
    st = sift.stage_ms()
    sift.close()
    base_px = st["base_pixels"]
    # one write and one read of every pyramid image (6 Gaussian + 5 DoG layers of every octave: 11 x 4/3 fp32 images of the doubled
    # base) + the BGR upload: the traffic a pyramid that keeps every layer cannot go below. The separable blurs as built move ~2 x that.
    pyr_bytes = base_px * 4.0 * 11 * (4.0 / 3.0) * 2 + N * N * 3.0
    kw = dict(max_features=250000, filter_ratio=0.55, ransac_epipolar_threshold=3.0, ransac_max_iter=10000, ransac_confidence=0.999999,
              rsift_eps=1e-8, ctx=ctx)
    Hm, stats, tm = register_once(frame, ortho, **kw)                    # first call: allocates the pyramids (kept by the library)
    steps = max(min(args.steps, 3), 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        Hm, stats, tm = register_once(frame, ortho, **kw)
    elapsed = time.perf_counter() - t0
    free1, _ = mem()
    ys, xs = np.meshgrid(np.linspace(0, H - 1, 9), np.linspace(0, W - 1, 16), indexing="ij")
    P = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    err = None
    if Hm is not None:
        pa, pb = Hm @ P, A @ P
        err = float(np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max())
    nq, nt = int(stats[0]), int(stats[1])
    rng = np.random.default_rng(0)
    d = np.sqrt(rng.gamma(0.6, 1.0, (max(nq, nt, 2), 128)).astype(np.float32))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    *_, ms = ops.match_2nn(d[:max(nq, 1)], d[rng.permutation(max(nt, 2))], iters=3, ctx=ctx)
    mflops = 2.0 * nq * nt * 128
    out = {"metric": "frame-to-orthophoto registrations/sec at the reference's size (RootSIFT + 2-NN + robust homography)", "value": steps / elapsed,
           "unit": "registrations/s", "n_gpus": 1, "steps": steps, "warmup": 1, "ms_per_step": 1000.0 * elapsed / steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32 keypoints/descriptors, f16 MFMA matching", "data": "synthetic",
           "config": {"workload": f"estimate_homography GPU path: one 3840x2160 frame against a {N}x{N} synthetic orthophoto cut-out, max_features 250000 "
                                  "(geotrax/cfg/default.yaml:154,158-168), host images in",
                      "keypoints": {"frame": nq, "orthophoto": nt, "orthophoto_detect_pass": int(k_ortho["count"])},
                      "good_matches": int(stats[2]), "inliers": int(stats[3]),
                      "stage_ms": {"detect_describe_both_images": float(tm[0]), "match": float(tm[1]), "ratio": float(tm[2]), "fit": float(tm[3])},
                      "orthophoto_sift_stage_ms": {"pyramid": st["pyramid"], "extrema_refine_orient": st["keypoints"], "describe": st["describe"]},
                      "max_grid_error_px_vs_known_mapping": err,
                      "hbm_gb": {"total": total / 1e9, "orthophoto_sift_buffers": (free0 - free_sift) / 1e9, "held_after_registration": (free0 - free1) / 1e9},
                      "match_2nn": {"nq": nq, "nt": nt, "ms": float(ms), "tflops": mflops / (ms * 1e-3) / 1e12 if ms > 0 else None,
                                    "frac_of_fp16_mfma_peak": mflops / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS["f16"] if ms > 0 else None,
                                    "note": "gtx_op_match_2nn on random unit descriptors of the keypoint counts reached, incl. the exact-distance finish"},
                      "host_seconds_to_render_the_orthophoto": t_make},
           "roofline": {"bound": "hbm", "kernel": "sift pyramid (gray, upscale, blur_h / blur_v, down, sub kernels of csrc/sift.hip)",
                        "achieved": pyr_bytes / (st["pyramid"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": pyr_bytes / (st["pyramid"] * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "bytes_per_pass": pyr_bytes, "pass_ms": st["pyramid"],
                        "timing": "HIP events around the pyramid stage of the orthophoto's detect pass (gtx_sift_stage_ms); algorithmic bytes = every "
                                  "pyramid image written once and read once + the BGR upload"}}
    print(json.dumps(out), flush=True)


def bench_georef(args):
    """--workload georef: one step = gtx_op_georef_points over the rows of a track table (host arrays in and out, as
    the georeference stage holds them): 2 M rows, i.e. ~100 golden clips' worth (19 817 rows each). The kernel is
    HBM-bound by construction (16 B in, 48 B out per row); what this line measures includes the PCIe copies."""
    from geotrax_amd import _lib
    from geotrax_amd.georeference import apply_homography, geo2local, ortho2geo, transform_points

    ctx = _lib.Context(0)
    n = 2_000_000
    rng = np.random.default_rng(0)
    x, y = rng.uniform(0, W, n), rng.uniform(0, H, n)
    Hm = np.array([[1.91, 0.08, 5120.5], [-0.07, 1.88, 3310.25], [1.3e-6, -0.9e-6, 1.0]])
    ortho = (126.6412, 37.3951, 2.4e-7, -1.9e-7, 1.1e-9, -0.7e-9)
    steps, warm = min(args.steps, 20), min(args.warmup, 3)
    for _ in range(max(warm, 1)):
        out = transform_points(x, y, Hm, ortho, "EPSG:4326", "EPSG:5186", ctx=ctx)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = transform_points(x, y, Hm, ortho, "EPSG:4326", "EPSG:5186", ctx=ctx)
    elapsed = time.perf_counter() - t0
    m = 200_000                                                   # CPU baseline: the host chain (reference-like numpy) on a bounded sample
    t0 = time.perf_counter()
    ox, oy = apply_homography(x[:m], y[:m], Hm)
    lat, lon = ortho2geo(ox, oy, ortho)
    xl, yl = geo2local(lat, lon, "EPSG:4326", "EPSG:5186")
    cpu_s = time.perf_counter() - t0
    err = float(max(np.abs(out["x_local"][:m] - xl).max(), np.abs(out["y_local"][:m] - yl).max()))
    print(json.dumps({"metric": "track rows/sec through the georeference transform chain", "value": n * steps / elapsed, "unit": "rows/s", "n_gpus": 1,
                      "steps": steps, "warmup": warm, "ms_per_step": 1000.0 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "f64", "data": "synthetic",
                      "config": {"workload": f"georeference.py:173-177 row chain on {n} rows: homography -> orthophoto geotransform -> Korea 2000 central belt (EPSG:5186), "
                                             "host arrays in and out (PCIe inclusive)", "max_abs_diff_vs_host_chain_m": err},
                      "cpu_baseline": {"value": m / cpu_s, "unit": "rows/s", "cores": 1, "kind": "port",
                                       "sample": f"{m} rows through the host numpy chain (apply_homography, ortho2geo, geo2local)"}}), flush=True)


def bench_warp(args):
    """--workload warp: one step = gtx_warp_frame_dev on a 3840x2160 BGR frame resident in HBM (24.9 MB read + 24.9 MB
    written): the frame warp of the reference's visualisation modes 1/4. HBM-bound by construction; the roofline object
    prices it against 8 TB/s. Timed as K back-to-back launches on one stream between two synchronisations."""
    from geotrax_amd import _lib
    from geotrax_amd.synth import make_scene
    from geotrax_amd.warp import FrameWarper

    ctx = _lib.Context(0)
    scene = make_scene(seed=0, h=H, w=W)
    frame = scene.render(40, 150)
    Hm = np.linalg.inv(scene.camera(40, 150)) @ scene.camera(0, 150)        # frame 40 -> frame 0, the matrix extract writes
    wp = FrameWarper((H, W), ctx=ctx)
    ctx.dev_upload(wp.src, frame)
    steps, warm = max(args.steps, 1), max(args.warmup, 1)
    for _ in range(warm):
        wp.warp_dev(wp.src, Hm, wp.dst)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wp.warp_dev(wp.src, Hm, wp.dst)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    out_frame = np.empty_like(frame)
    ctx.dev_download(out_frame, wp.dst)
    nbytes = 2.0 * frame.nbytes
    gbs = nbytes * steps / elapsed / 1e9
    line = {"metric": "4K frames/sec through the perspective frame warp", "value": steps / elapsed, "unit": "frames/s", "n_gpus": 1,
            "steps": steps, "warmup": warm, "ms_per_step": 1000.0 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (f64 coordinates)", "data": "synthetic",
            "config": {"workload": "cv2.warpPerspective equivalent (INTER_LINEAR, constant border) on a 3840x2160 BGR frame resident in HBM, "
                                   "homography of frame 40 of the synthetic clip"},
            "roofline": {"bound": "hbm", "kernel": "warp_kernel", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": None, "avg_launch_us": 1e6 * elapsed / steps, "bytes_per_launch": nbytes,
                         "timing": f"{steps} back-to-back launches on one stream between two synchronisations (launch gaps included)"}}
    if not args.no_cpu_baseline:
        from oracle.warp_ref import warp_perspective as ref

        t0 = time.perf_counter()
        want = ref(frame, Hm)
        cpu_s = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "frames/s", "cores": 1, "kind": "port",
                                "sample": "one 3840x2160 frame through oracle/warp_ref.py (vectorised numpy restatement of cv2.warpPerspective)"}
        line["config"]["pixels_differing_from_oracle"] = int(np.count_nonzero(out_frame != want))
    print(json.dumps(line), flush=True)


def bench_extract_georef(args):
    """--workload extract+georef (BASELINE configs[3]): one clip through the product's extract engine
    (detect + track + stabilize), its rows through aggregate/post-processing, then the georeference stage on the result:
    RootSIFT registration of the reference frame against a synthetic orthophoto (a zoomed, rotated render of the same
    scene, ground truth known), the per-row chain frame px -> orthophoto px -> lat/lon -> local metres on the GPU,
    dimensions / visibility / kinematics and the CSV. Measured twice: unpaced (`value`: clip frames / (extract + georeference
    time)) and as a 30 fps stream (frames become available every 1/30 s; reported under `paced`: sustained rate, worst
    frame-in -> result-out latency, and the georeference tail after the last frame)."""
    import logging
    import tempfile

    from geotrax_amd import _lib
    from geotrax_amd import georef_stage as gs
    from geotrax_amd import georeference as gr
    from geotrax_amd.engine import ExtractEngine
    from geotrax_amd.postprocess import aggregate_results, postprocess_tracks
    from geotrax_amd.registration import estimate_homography
    from geotrax_amd.synth import make_scene
    from geotrax_amd.tracker import Tracker

    logger = logging.getLogger("bench.georef")
    logger.setLevel(logging.ERROR)
    from geotrax_amd.engine import StreamPlan

    ctx = StreamPlan.get(0, max(args.det_streams, 1), max(args.stab_streams, 1)).take("d")   # the first detector's place in the engine's stream plan
    scene = make_scene(seed=0, h=H, w=W)
    n_pool = max(args.frames, 2)
    frames = [scene.render(t, 150) for t in range(n_pool)]
    ortho, A_true = scene.orthophoto()
    ortho_params = (126.6412, 37.3951, 2.4e-7, -1.9e-7, 0.0, 0.0)
    args.tracker = args.tracker or "bytetrack"
    det, weights, n_det, n_cand = calibrated_detector(ctx, frames[0], args, args.detections)
    B = max(args.batch, 1)
    order = list(range(n_pool)) + list(range(n_pool - 2, 0, -1))
    seq = order + [order[i % len(order)] for i in range(B - 1)]
    fbytes = frames[0].nbytes
    pool = ctx.dev_alloc(fbytes * len(seq))
    for i, t in enumerate(seq):
        ctx.dev_upload(pool + i * fbytes, frames[t])
    det_kw = dict(imgsz=args.imgsz, conf=0.25, iou=0.7, max_det=1000, classes=[0, 1, 2, 3], agnostic_nms=True, half=bool(args.half),
                  fp32_split=fp32_split(args), rect=bool(args.rect))
    tracker = Tracker(args.tracker)
    engine = ExtractEngine(weights, (H, W), det_kw, tracker, {}, device=0, batch=B, det_streams=args.det_streams, stab_streams=args.stab_streams,
                           gmc=args.tracker in GMC_TRACKERS, detectors=[det])
    gcfg = dict(transformation=dict(source_crs="epsg:4326", target_crs="epsg:5186", cutout_width_px=None),
                filtering=dict(filter_type="gaussian", kernel_size=14, min_traj_length=15, visibility_margin=4))
    matching = dict(detector_name="rsift", matcher_name="bf", filter_type="ratio", sift_enable_precise_upscale=True, max_features=250000, filter_ratio=0.55,
                    ransac_method=38, ransac_epipolar_threshold=3.0, ransac_max_iter=10000, ransac_confidence=0.999999, rsift_eps=1e-8)
    main_cfg = {"main": {"extraction": {"min_track_length": 3, "interpolate": False, "dimension_estimation": dict(
        gsd=0.02725, eps=4, r0=1.25, theta_bar=15, tau_c={0: 1.83, 1: 2.85, 2: 1.70, 3: 1.80, -1: 1.70})},
        "args": argparse.Namespace(source="synthetic.mp4", interpolate=False), "tracker": {"active": args.tracker, args.tracker: {"track_buffer": 30}}}}
    n_steps = max(min(args.steps, 75), 2)                      # 75 steps x 2 frames = the 150 frames of the reference's 5 s clip
    outdir = Path(tempfile.mkdtemp(prefix="gtx_bench_georef_"))

    def one_pass(pace_fps):
        tracker.reset()
        engine.reset()
        cols = dict(frame=[], ids=[], raw=[], stab=[], cls=[], conf=[], tr=[])
        avail, lat = {}, []
        t_start = time.perf_counter()

        def batches():
            for k in range(n_steps):
                if pace_fps:                                    # the last frame of batch k exists at (k*B + B - 1) / fps
                    due = t_start + (k * B + B - 1) / pace_fps
                    while time.perf_counter() < due:
                        time.sleep(max(min(due - time.perf_counter(), 0.002), 0))
                for b in range(B):                              # a frame exists from its own arrival time on, not from its batch's
                    avail[k * B + b] = min(time.perf_counter(), t_start + (k * B + b) / pace_fps) if pace_fps else time.perf_counter()
                yield pool + ((k * B) % len(order)) * fbytes

        for r in engine.run(batches(), paced=bool(pace_fps)):
            lat.append(time.perf_counter() - avail[r.index])
            if r.xywh is not None:
                n = len(r.xywh)
                cols["frame"].append(np.full((n, 1), r.index, dtype=np.uint32))
                cols["ids"].append(np.full((n, 1), -1) if r.ids is None else np.asarray(r.ids).astype(np.uint16).reshape(-1, 1))
                cols["raw"].append(r.xywh.astype(np.float32))
                cols["stab"].append(r.xywh_stab)
                cols["cls"].append(np.asarray(r.cls).astype(np.uint8).reshape(-1, 1))
                cols["conf"].append(np.asarray(r.conf).astype(np.float32).reshape(-1, 1))
            if r.index > 0 and r.H is not None:
                cols["tr"].append(np.hstack((np.array([[r.index]]), r.H.reshape(1, -1))))
        ctx.synchronize()
        t_extract = time.perf_counter() - t_start
        t0 = time.perf_counter()
        tracks, transforms = aggregate_results(cols["frame"], cols["ids"], cols["raw"], cols["stab"], cols["cls"], cols["conf"], cols["tr"], logger)
        tracks = postprocess_tracks(tracks, main_cfg, logger, (W, H))
        t_post = time.perf_counter() - t0
        t0 = time.perf_counter()
        Hm, inl, nm, nkp = estimate_homography(frames[0], ortho, logger, ctx=ctx, **matching)
        t_reg = time.perf_counter() - t0
        t0 = time.perf_counter()
        df = gs.georeference_tracks(tracks[:, 1].astype(int), tracks[:, 0].astype(int), tracks[:, 2:6].astype(np.float64), tracks[:, 6].astype(np.float64),
                                    tracks[:, 7].astype(np.float64), tracks[:, 10].astype(int), tracks[:, 12:14].astype(np.float64), None, np.array([]),
                                    (H, W), 30.0, Hm, ortho_params, None, gcfg, logger, ctx=ctx)
        gr.save_georeferenced_data(outdir / "synthetic.csv", df, logger)                    # the product's writers (geotrax_amd/tables.py)
        gr.save_homography(outdir / "synthetic_geo_transf.txt", Hm, logger)
        t_geo = time.perf_counter() - t0
        ys, xs = np.meshgrid(np.linspace(0, H - 1, 9), np.linspace(0, W - 1, 16), indexing="ij")
        P = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
        pa, pb = Hm @ P, A_true @ P
        err = float(np.abs(pa[:2] / pa[2] - pb[:2] / pb[2]).max())
        return dict(t_extract=t_extract, t_post=t_post, t_reg=t_reg, t_geo=t_geo, rows=int(len(tracks)), csv_rows=int(len(df)), vehicles=int(df["Vehicle_ID"].nunique()),
                    lat_max=float(max(lat)), lat_med=float(np.median(lat)), err=err, kp=[int(v) for v in nkp], inliers=int(inl), matches=int(nm))

    one_pass(0)                                                 # warm-up: kernels loaded, SIFT pyramids allocated
    u = one_pass(0)
    p = one_pass(30.0)
    n_fr = n_steps * B
    tail_u = u["t_post"] + u["t_reg"] + u["t_geo"]
    tail_p = p["t_post"] + p["t_reg"] + p["t_geo"]
    dt = "f16" if args.half else ("f32s" if det.fp32_split else "f32")
    line = {"metric": "4K frames/sec through detect+stabilize+track, then orthophoto georeference", "value": n_fr / (u["t_extract"] + tail_u), "unit": "frames/s",
            "n_gpus": 1, "steps": n_steps, "warmup": n_steps, "ms_per_step": 1000.0 * (u["t_extract"] + tail_u) / n_steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAME[dt], "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: full extract ({args.tracker}) of a {n_fr}-frame 3840x2160 clip + georeference stage "
                                   "(RootSIFT registration against a 4800x4800 synthetic orthophoto, row chain on the GPU, kinematics, CSV)",
                       "frames": n_fr, "track_rows": u["rows"], "csv_rows": u["csv_rows"], "vehicles": u["vehicles"],
                       "extract_s": u["t_extract"], "postprocess_s": u["t_post"], "registration_s": u["t_reg"], "row_chain_kinematics_csv_s": u["t_geo"],
                       "registration": {"keypoints": u["kp"], "matches": u["matches"], "inliers": u["inliers"], "max_grid_error_px_vs_known_orthophoto": u["err"]}},
            "paced": {"stream_fps": 30.0, "sustained_fps": n_fr / p["t_extract"], "frame_latency_ms": {"median": 1000 * p["lat_med"], "max": 1000 * p["lat_max"]},
                      "georeference_tail_ms": 1000 * tail_p,
                      "note": "frames become available every 1/30 s; latency = a frame's own arrival -> its tracked, stabilized result leaves the engine (the first frame of a batch of 2 waits a frame period for the second)"}}
    print(json.dumps(line), flush=True)
    engine.close()


def bench_cli(args):
    """--workload cli: what a `geotrax extract` user gets from a file. A 150-frame 3840x2160 clip of the
    synthetic scene (ping-pong over --frames distinct renders, like the resident pool of the default workload) is written to
    local disk as .y4m (I420, 12.4 MB per frame) and .npy (BGR, 24.9 MB per frame) together with the calibrated weights and a
    config file; then the product's own loop, geotrax_amd.extract.track_with_model, runs on each file: once to warm up
    (kernels loaded, file in the page cache), then timed. Reported per container: wall-clock frames/s from the open file to
    the aggregated tables (reader + PCIe + GPU + tracker), and the reference's own convention 1000 n / (sum det_ms + sum
    stab_ms) (extract.py:204-207), with the read-ahead feeder (the product default) and with the synchronous reader."""
    import logging
    import shutil
    import tempfile

    import yaml
    from geotrax_amd import _lib
    from geotrax_amd import extract as ex
    from geotrax_amd.config_utils import DEFAULT_CFG, load_config_all
    from geotrax_amd.frames import bgr_to_i420, write_y4m
    from geotrax_amd.synth import make_scene
    from geotrax_amd.weights import save_weights

    logger = logging.getLogger("bench.cli")
    logger.setLevel(logging.ERROR)
    args.tracker = args.tracker or "bytetrack"
    from geotrax_amd.engine import StreamPlan

    plan = StreamPlan.get(0, max(args.det_streams, 1), max(args.stab_streams, 1))   # the streams the product loop below will run on
    ctx = plan.take("d")
    scene = make_scene(seed=0, h=H, w=W)
    n_pool = max(args.frames, 2)
    frames = [scene.render(t, 150) for t in range(n_pool)]
    order = list(range(n_pool)) + list(range(n_pool - 2, 0, -1))
    seq = [order[i % len(order)] for i in range(max(args.cli_frames, 2))]
    det, weights, n_det, n_cand = calibrated_detector(ctx, frames[0], args, args.detections)
    dt = "f16" if args.half else ("f32s" if det.fp32_split else "f32")
    det.close()
    plan.give_back(ctx)
    root = Path(args.cli_dir) if args.cli_dir else Path(tempfile.mkdtemp(prefix="gtx_bench_cli_"))
    root.mkdir(parents=True, exist_ok=True)
    wpath = root / "weights.safetensors"
    save_weights(weights, wpath)
    wpath.with_suffix(".names.yaml").write_text("{0: car, 1: bus, 2: truck, 3: motorcycle}\n")
    cfg = yaml.safe_load(DEFAULT_CFG.read_text())
    cfg["ultralytics"].update(imgsz=args.imgsz, half=bool(args.half), max_det=1000, rect=bool(args.rect), conf=0.25, iou=0.7, classes=[0, 1, 2, 3], agnostic_nms=True)
    cfg["tracker"]["active"] = args.tracker
    cfg["extraction"]["model"] = str(wpath)
    cfg.setdefault("engine", {}).update(batch=max(args.batch, 1), det_streams=args.det_streams, stab_streams=args.stab_streams)
    if args.fp32 is not None:
        cfg["engine"]["fp32_split"] = args.fp32 == "split"
    cfg_path = root / "cfg.yaml"
    cfg_path.write_text(yaml.safe_dump(cfg))
    t0 = time.perf_counter()
    files = {}
    for fmt in [f for f in args.cli_formats.split(",") if f]:
        path = root / f"clip.{fmt}"
        if fmt == "y4m":
            planes = [bgr_to_i420(f) for f in frames]
            write_y4m(path, [planes[t] if k else frames[t] for k, t in enumerate(seq)])   # (the first item tells write_y4m the size)
        elif fmt == "npy":
            out = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint8, shape=(len(seq), H, W, 3))
            for k, t in enumerate(seq):
                out[k] = frames[t]
            out.flush()
            del out
        else:
            raise SystemExit(f"--cli-formats: unknown container '{fmt}'")
        files[fmt] = path
    t_write = time.perf_counter() - t0

    def one_run(path, feeder_on):
        os.environ["GTX_FEEDER"] = "1" if feeder_on else "0"
        a = argparse.Namespace(source=str(path), cfg=cfg_path, output_folder=None, log_path=None, verbose=False, model=None, class_names=None,
                               conf=None, classes=None, cut_frame_left=None, cut_frame_right=None, interpolate=None)
        model = ex.load_detector(a, logger)
        config = load_config_all(a, logger, model_names=model.names)
        t0 = time.perf_counter()
        tracks, transforms = ex.track_with_model(model, config, logger)
        wall = time.perf_counter() - t0
        lr = dict(getattr(model, "last_run", {}) or {})
        assert len(tracks) and lr.get("frames") == len(seq), (len(tracks), lr)
        import zlib

        digest = zlib.crc32(np.ascontiguousarray(tracks).tobytes()) ^ zlib.crc32(np.ascontiguousarray(transforms).tobytes())
        return dict(digest=digest, frames_per_s=lr["loop_fps"], frames_per_s_incl_setup=lr["wall_fps"], engine_setup_s=lr["engine_setup_s"], reader_setup_s=lr["reader_setup_s"],
                    loop_s=lr["loop_s"], reference_convention_fps=lr["reference_convention_fps"], det_ms_per_frame=lr["det_ms"],
                    stab_ms_per_frame=lr["stab_ms"], track_with_model_s=wall, track_rows=int(len(tracks)), transforms=int(len(transforms)))

    res = {}
    for fmt, path in files.items():
        warm = one_run(path, True)                              # warm-up: kernels, allocator, page cache
        runs = [one_run(path, True) for _ in range(2)]
        r = max(runs, key=lambda d: d["frames_per_s"])
        r["all_runs_frames_per_s"] = [warm["frames_per_s"]] + [x["frames_per_s"] for x in runs]   # the first one is the cold run
        # the pipeline is asynchronous end to end (reader threads, copy stream, 2 + 4 GPU streams, three host stages): its
        # output must not depend on timing -- the three runs' tables are the same bytes
        r["deterministic"] = len({warm["digest"], runs[0]["digest"], runs[1]["digest"]}) == 1
        r["bytes_per_frame"] = int(path.stat().st_size // len(seq))
        r["file_gbs"] = r["bytes_per_frame"] * r["frames_per_s"] / 1e9
        if args.cli_compare_sync:
            s_ = one_run(path, False)
            r["deterministic"] = r["deterministic"] and s_["digest"] == r["digest"]      # ... and not on the reader either
            r["synchronous_reader"] = {"frames_per_s": s_["frames_per_s"], "frames_per_s_incl_setup": s_["frames_per_s_incl_setup"], "reference_convention_fps": s_["reference_convention_fps"],
                                       "note": "GTX_FEEDER=0: f.read() + pageable upload on the detector stage thread (round 3's reader)"}
        res[fmt] = r
    os.environ.pop("GTX_FEEDER", None)
    if not args.cli_dir:
        shutil.rmtree(root, ignore_errors=True)
    head = res.get("y4m") or next(iter(res.values()))
    line = {"metric": "4K frames/sec through detect+stabilize+track, from a file", "value": head["frames_per_s"], "unit": "frames/s", "n_gpus": 1,
            "steps": len(seq) // max(args.batch, 1), "warmup": len(seq) // max(args.batch, 1), "ms_per_step": 1000.0 * max(args.batch, 1) / head["frames_per_s"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAME[dt],
            "data": "synthetic (a clip file on local disk, in the page cache after the warm-up run; read, uploaded and converted inside the timed region)",
            "config": {"workload": f"geotrax_amd.extract.track_with_model on a {len(seq)}-frame 3840x2160 clip file ({', '.join(files)}): YOLOv8s + {TRACKER_LABEL[args.tracker]} + "
                                   "homography stabilization, the product's reader -> engine -> aggregate path (BASELINE configs[2] from a file)",
                       "imgsz": args.imgsz, "rect": bool(args.rect), "half": bool(args.half), "detections_per_frame_calibrated": n_det,
                       "clip_write_s": t_write, "distinct_frames": n_pool,
                       "reader": "read-ahead feeder: pread into pinned slots on the library's threads, async upload + I420->BGR on a copy stream (geotrax_amd/feeder.py, csrc/feeder.cpp)",
                       "reference_convention": "1000 n / (sum of per-frame detector ms + sum of per-frame stabilizer ms), extract.py:204-207: what the reference's third log line "
                                               "would print for these stage times; its stages run one after the other, here they overlap"},
            "from_file": res}
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    rc = self_launch(args)                                  # --gpus N > 1 without a launcher: N ranks as a child process, nothing of HIP touched here
    if rc is not None:
        sys.exit(rc)
    if args.launch_check:
        sys.exit(launch_check(args))
    if args.workload == "cli":
        return bench_cli(args)
    if args.workload == "extract+georef":
        return bench_extract_georef(args)
    if args.workload == "warp":
        return bench_warp(args)
    if args.workload == "register":
        return bench_register(args)
    if args.workload == "georef":
        return bench_georef(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.tracker is None:
        args.tracker = "botsort" if world > 1 else "bytetrack"
    dist = None
    force_dist = os.environ.get("GTX_BENCH_FORCE_DIST") == "1"   # test hook: the N > 1 code path with one rank
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist

        if args.backend == "nccl":
            from geotrax_amd.engine import StreamPlan

            StreamPlan.get(local, max(args.det_streams, 1), max(args.stab_streams, 1) if args.workload == "extract" else 0)   # before RCCL creates its streams
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:                                                   # test mode: ranks may share a GPU, collectives on the host
            local = local % max(torch.cuda.device_count(), 1)
            dist.init_process_group("gloo")

    cdev = f"cuda:{local}" if args.backend == "nccl" else "cpu"     # where the collectives' tensors live
    from geotrax_amd import _lib
    from geotrax_amd.distributed import gather_records, pack_frame_record, pin_to_core, reserve_replay_core
    from geotrax_amd.geometry import warp_boxes
    from geotrax_amd.synth import make_scene
    from geotrax_amd.tracker import Tracker

    from geotrax_amd.engine import StreamPlan

    # the first detector's context, from the device's stream plan (geotrax_amd/engine.py: all of the engine's streams are created
    # here, at once, in the order that gives each detector a hardware queue of its own; the engine below takes the rest)
    extract_like = args.workload == "extract"
    ctx = StreamPlan.get(local, max(args.det_streams, 1), max(args.stab_streams, 1) if extract_like and os.environ.get("GTX_BENCH_NO_STAB") != "1" else 0).take("d")
    scene = make_scene(seed=0 if args.sharding == "frames" else rank, h=H, w=W)   # frames: one clip, ranks take different batches of it; videos: a clip per rank
    n_pool = max(args.frames, 2)
    frames = [scene.render(t, 150) for t in range(n_pool)]    # every rank holds the clip; it processes its own batches of it
    ref_frame = frames[0]
    if dist is None:
        det, weights, n_det, n_cand = calibrated_detector(ctx, ref_frame, args, args.detections)
    else:
        # rank 0 prepares the weights (here: calibrates the seeded set) and broadcasts them, one flat fp32 buffer
        # over RCCL (~44 MB); every rank then builds its detector from the same bytes
        import torch
        from geotrax_amd.detector import Detector

        from geotrax_amd.distributed import broadcast_weights

        meta = torch.zeros(2, dtype=torch.int64)
        weights = None
        if rank == 0:
            det, weights, n_det, n_cand = calibrated_detector(ctx, ref_frame, args, args.detections)
            det.close()
            meta[0], meta[1] = n_det, n_cand
        weights, _ = broadcast_weights(weights, None, dist, cdev)   # the product's own exchange (geotrax_amd/distributed.py)
        meta = meta.to(cdev)
        dist.broadcast(meta, src=0)
        n_det, n_cand = int(meta[0]), int(meta[1])
        det = Detector(weights, (H, W), imgsz=args.imgsz, conf=0.25, iou=0.7, max_det=1000, classes=[0, 1, 2, 3], agnostic_nms=True,
                       half=bool(args.half), fp32_split=fp32_split(args), rect=bool(args.rect), max_batch=max(args.batch, 1), ctx=ctx)
    # the ping-pong playback (0..n-1, n-2..1: continuous motion) laid out contiguously in HBM, plus the
    # first B-1 frames again, so that every batch of B consecutive frames is one contiguous range
    B = max(args.batch, 1)
    order = list(range(n_pool)) + list(range(n_pool - 2, 0, -1))
    seq = order + [order[i % len(order)] for i in range(B - 1)]
    fbytes = frames[0].nbytes
    pool = ctx.dev_alloc(fbytes * len(seq))
    for i, t in enumerate(seq):
        ctx.dev_upload(pool + i * fbytes, frames[t])

    CH = max(args.gather_every, 1)                               # steps per contiguous run of a rank = steps per gather
    sharded_run = (world > 1 or force_dist) and args.sharding == "frames" and args.workload == "extract"

    def global_batch(k):
        """Local step k -> global batch of the playback. Frame sharding deals the clip in runs of CH consecutive batches:
        within a gather interval rank r holds batches [r*CH, (r+1)*CH) of that interval, so its frames continue each other
        (BoT-SORT's GMC needs the previous frame: one priming frame per run instead of one per batch) and the gathered
        records are in clip order rank by rank. Warm-up and timed region are separate epochs."""
        if args.sharding != "frames" or world == 1:
            return k                                             # videos: every rank plays its own clip from the start
        base, kk, total = (0, k, args.warmup) if k < args.warmup else (args.warmup * world, k - args.warmup, args.steps)
        c, j = divmod(kk, CH)
        n_c = min(CH, total - c * CH)                            # the last run of an epoch may be shorter
        return base + c * CH * world + rank * n_c + j

    def batch_ptr(k):
        i0 = (global_batch(k) * B) % len(order)
        return pool + i0 * fbytes

    def batch_item(k):
        """What the engine is fed for local step k. Frame-sharded BoT-SORT run: the first batch of a run does not continue
        the rank's previous one, so the GMC is primed with the frame that precedes it in the clip (every rank holds the clip)."""
        if not shard_gmc:
            return batch_ptr(k)
        kk = k if k < args.warmup else k - args.warmup
        if kk % CH != 0 and world > 1:
            return batch_ptr(k)                                  # continues the run
        g = global_batch(k)
        return batch_ptr(k), (None if g == 0 else pool + ((g * B - 1) % len(order)) * fbytes)

    extract = args.workload == "extract"
    sharded = (world > 1 or force_dist) and args.sharding == "frames"
    tracker = Tracker(args.tracker)                              # N > 1: used by rank 0's replay thread only
    max_det = 1000
    # The product's own engine (geotrax_amd/engine.py, what `geotrax_amd.extract` runs): detector streams take the
    # batches round-robin, the tracker works in clip order, stabilizers (and BoT-SORT's GMC) run on their own streams.
    # N > 1: a shard rank's engine has no tracker (mask from the raw detections); rank 0 tracks the
    # gathered records. BoT-SORT there: each rank runs the GMC over [frame before its batch, its batch] and ships the
    # warps in the records; rank 0's tracker applies them in clip order.
    from geotrax_amd.engine import ExtractEngine

    det_kw = dict(imgsz=args.imgsz, conf=0.25, iou=0.7, max_det=max_det, classes=[0, 1, 2, 3], agnostic_nms=True, half=bool(args.half),
                  fp32_split=fp32_split(args), rect=bool(args.rect))
    stab_kw = {} if extract else None
    if os.environ.get("GTX_BENCH_NO_STAB") == "1":               # experiment: the extract loop without its stabilizer stage (tracker only)
        stab_kw = None
    shard_gmc = sharded and extract and args.tracker in GMC_TRACKERS
    if shard_gmc and args.gmc_method != "sparseOptFlow":
        raise SystemExit(f"bench: --gmc-method {args.gmc_method} runs unsharded only (a shard rank primes its GMC with the frame before its batch, which sparseOptFlow takes)")
    engine = ExtractEngine(weights, (H, W), det_kw, None if (sharded or not extract) else tracker, stab_kw, device=local, batch=B,
                           det_streams=args.det_streams, stab_streams=args.stab_streams,
                           gmc=(args.gmc_method if extract and args.tracker in GMC_TRACKERS else False), detectors=[det],
                           feeder_stream=bool(args.host_frames and world == 1))
    n_det_streams, n_stab, gmc = len(engine.dets), len(engine.stabs), engine.gmc
    host_feed = None
    if args.host_frames and world == 1:
        # the PCIe-inclusive measurement: the frames start as pageable host arrays and reach the engine the way the product brings
        # any host-side source in (geotrax_amd.feeder, memory mode: three of the library's threads copy them into the pinned ring, the copy stream
        # uploads them, the detector streams wait for the upload events) -- warm-up and timed steps from one feeder
        from geotrax_amd.feeder import FrameFeeder

        host_feed = FrameFeeder((H, W), kind="bgr", batch=B, ring=len(engine.dets) + 4, device=local, ctx=engine.feeder_ctx)
        host_feed.open_memory([frames[seq[(i // B * B) % len(order) + i % B]] for i in range((args.warmup + args.steps) * B)], n_threads=int(os.environ.get('GTX_BENCH_COPY_THREADS', '3')))
        host_batches = host_feed.batches(len(engine.dets))
    if extract and stab_kw is not None:
        engine.set_reference(ref_frame)                          # every rank registers against frame 0 of the clip
    records = []

    def run(k0, n_steps, sharded):
        """n_steps batches through the engine; per-frame results are identical to the frame-at-a-time order."""
        n_rows = 0
        import itertools

        src = itertools.islice(host_batches, n_steps) if host_feed is not None else (batch_item(k0 + k) for k in range(n_steps))
        stamps = [] if os.environ.get("GTX_BENCH_STAMPS") == "1" else None   # experiment: when each frame's result leaves the engine
        t_run = time.perf_counter()
        for r in engine.run(src):
            n_rows = len(r.xyxy)
            if stamps is not None:
                stamps.append(time.perf_counter() - t_run)
            if sharded and extract:
                records.append(pack_frame_record(max_det, r.xyxy, r.conf, r.cls, r.H, r.gmc, with_gmc=shard_gmc))
                if live[0]:
                    gather_ready()
        if stamps:
            print("result stamps (ms since run()): " + " ".join(f"{1e3 * t:.2f}" for t in stamps) + f" | generator done {1e3 * (time.perf_counter() - t_run):.2f}", file=sys.stderr)
        if engine.prof and stamps is not None:
            print("stage marks (ms): " + " ".join(f"{w}{i}@{1e3 * t:.2f}" for w, i, t in sorted(engine.marks, key=lambda m: m[2])), file=sys.stderr)
        if engine.prof:                                          # GTX_ENGINE_PROF=1: where the host stages wait
            print("engine host stages (s): " + ", ".join(f"{k} {v:.4f}" for k, v in sorted(engine.prof.items())), file=sys.stderr)
        return n_rows

    # ---- N > 1: chunked gather to rank 0 + tracker replay on a second host thread
    sent = [0]                                                   # steps of this rank already handed to a gather
    replay_q = queue.Queue()
    replay_rows = [0]
    comm_stream = None
    if dist is not None and args.backend == "nccl":
        import torch

        comm_stream = torch.cuda.Stream()                        # non-blocking: torch copies/collectives never touch the kernels' streams

    def gather_ready(final=False):
        """Hands every complete chunk of --gather-every steps (all of the rest when final) to one gather."""
        import contextlib

        import torch

        chunk = max(args.gather_every, 1)
        while True:
            done_steps = len(records) // B
            n = min(chunk, done_steps - sent[0])
            if n <= 0 or (n < chunk and not final):
                return
            block = np.stack(records[sent[0] * B:(sent[0] + n) * B])          # [n*B, stride], step-major
            sent[0] += n
            with (torch.cuda.stream(comm_stream) if comm_stream is not None else contextlib.nullcontext()):
                # the product's exchange (geotrax_amd.distributed.gather_records): records packed to their real length, one
                # all-gather of the packed sizes, one padded gather to rank 0
                blocks, _ = gather_records(block, False, dist, torch.device(cdev), max_det, shard_gmc)
                if rank == 0:
                    # clip order: every rank holds one contiguous run of the interval -> rank-major is clip order
                    replay_q.put(np.concatenate(blocks))

    replay_core = reserve_replay_core(world) if sharded_run else None   # every rank's threads stay off the core rank 0's replay thread takes

    def replay_worker():
        pin_to_core(replay_core)
        while True:
            item = replay_q.get()
            if item is None:
                return
            replay_rows[0] = replay_tracker(item)

    def replay_tracker(all_records):
        # one C call for the whole gathered run (gtx_tracker_replay: tracker.update on every frame, empty or not, with the
        # rank's camera-motion warp for BoT-SORT), then the box warp of every frame's tracks by that frame's homography
        per, bx, ids = tracker.replay(all_records, max_det, with_gmc=shard_gmc)[:3]
        o = 0
        for f, k in enumerate(per):
            if k and all_records[f, -10] > 0:
                warp_boxes(all_records[f, -9:].reshape(3, 3), xywh_of(bx[o:o + k]))
            o += k
        return int(per[-1]) if len(per) else 0

    n_tracks = 0
    live = [False]                                               # gathers only inside the timed region
    if args.warmup > 0:
        n_tracks = run(0, args.warmup, sharded)
    records.clear()
    tracker.reset()
    engine.reset(keep_reference=True)

    def barrier():
        ctx.synchronize()
        if dist is not None:
            import torch

            dist.barrier()
            if args.backend == "nccl":
                torch.cuda.synchronize()

    if rank == 0 and not args.no_profile:
        for d in engine.dets:
            d.trace(args.trace_every)
    worker = None
    if sharded and extract:
        live[0] = True
        if rank == 0:
            worker = threading.Thread(target=replay_worker, daemon=True)
            worker.start()
    barrier()
    t0 = time.perf_counter()
    failure = None
    try:
        n_tracks = run(args.warmup, args.steps, sharded)
        ctx.synchronize()
    except Exception as e:                                       # this rank's shard is lost; it still joins every collective below
        failure = f"rank {rank}: {type(e).__name__}: {e}"
        if sharded and extract:                                  # empty records for the steps it did not finish (same gather count on every rank)
            empty = pack_frame_record(max_det, np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32), None, None, with_gmc=shard_gmc)
            while len(records) < args.steps * B:
                records.append(empty)
    if sharded and extract:
        gather_ready(final=True)                                 # the steps since the last full chunk
        if worker is not None:
            replay_q.put(None)
            worker.join()                                        # the tracker has seen every frame of every rank
            n_tracks = replay_rows[0]
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed, 1.0 if failure else 0.0], device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if float(t[1].item()) > 0 and failure is None:
            failure = "another rank failed (see its stderr)"
    if failure is not None:
        print(f"bench: {failure}", file=sys.stderr, flush=True)
    barrier()
    if host_feed is not None:
        host_feed.close()

    if rank == 0:
        dt = "f16" if args.half else ("f32s" if det.fp32_split else "f32")
        out = {
            "metric": "4K frames/sec through detect+stabilize+track", "value": (args.steps * B * world / elapsed) if failure is None else 0.0, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAME[dt], "data": "synthetic" + (" (frames uploaded from host memory inside the timed region: PCIe-inclusive)" if getattr(args, "host_frames", False) else ""),
            "config": {
                "workload": (f"full extract: {MODEL_LABEL[args.model]} + {TRACKER_LABEL[args.tracker]} + homography stabilization on 3840x2160 frames, "
                             f"{B} frame(s) per step (BASELINE {'configs[2]' if world == 1 else 'configs[4]: frames of one clip over the ranks'}; metric 'detect+stabilize+track')" if extract else
                             f"{MODEL_LABEL[args.model]} HIP inference only, 3840x2160 frames, batch={B} (BASELINE configs[1] is batch=1)"),
                "imgsz": args.imgsz, "rect": bool(args.rect), "net_input": list(det.net_hw), "half": bool(args.half),
                "arithmetic": {"f16": "fp16 activations and weights, fp16 MFMA, fp32 accumulate",
                               "f32": "fp32 activations, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                               "f32s": "fp32 activations in HBM; conv operands split into hi+lo fp16 parts, three fp16 MFMAs per product, fp32 accumulate "
                                       "(22 significand bits per operand; passes the fp32 parity assertions of tests/test_detector_gpu.py unchanged)"}[dt],
                "detect_head": (("box branch (cv2[l][0], cv2[l][1]) evaluated at the score gate's candidates only, the dense layers' values bit for bit "
                                 "(csrc/head_sparse.hip; every other layer dense); batches over 8192 candidates per image finish on the dense layers: "
                                 f"{det.sparse_box()[1]} such in this run") if det.sparse_box()[0] else "dense"),
                "letterbox_padding": ((lambda ps: (f"rect = false puts the 16:9 frame in {round(H * min(args.imgsz / H, args.imgsz / W))} of {det.net_hw[0]} input rows; activation rows out of reach "
                                                    f"of them are constants of the checkpoint, computed once when the detector is created and left out of the launches afterwards: "
                                                    f"{ps[1]} of {ps[2]} tile rows per image and pass (results bit-identical to computing them every pass; the `every_row_every_pass` "
                                                    f"key is this line without it)") if ps[0] else "every row computed in every pass")(det.pad_skip())),
                "tracker": args.tracker + (f" + {args.gmc_method} GMC on the GPU" + (" (per shard rank, primed with the frame before each batch)" if shard_gmc else "") if gmc is not None else ""), "stabilo": "orb 2000/4000 features, ratio 0.9, ransac 2 px, downsample 0.5, mask on",
                "weights": ("seeded synthetic RT-DETR-l (no checkpoint reachable; rtdetr-l.yaml topology, 32 M parameters): the last decoder score head shifted on one frame so that about the golden clip's box count of the 300 queries clears conf"
                            if args.model == "rtdetr-l" else
                            "seeded synthetic YOLOv8s (no checkpoint reachable): class bias calibrated on one frame to the golden clip's box count, only the stride-8 head fires so boxes are vehicle-sized (~100 px in 4K)"),
                "detections_per_frame": n_det, "candidates_per_frame": n_cand, "tracks_last_step": int(n_tracks),
                "nms_path": "none (RT-DETR: the decoder's 300 queries, score threshold + class filter + descending order in rt_post_kernel)" if args.model == "rtdetr-l" else ("one workgroup per image (nms_small_kernel: <= 4096 candidates, the product's path for any frame of the golden clip's kind)"
                             if n_cand <= 4096 else "rank / mask / resolve kernels (> 4096 candidates)"),
                "frames_per_step": B, "frames_per_rank_in_hbm": len(seq),
                "pipeline": f"{n_det_streams} detector stream(s) take batches round-robin and stay in flight while tracker/stabilizers work through the collected batch; {n_stab} stabilizer streams (submit/collect C ABI)",
                "sharding": "none (reference per-frame order)" if world == 1 else
                            "one clip per rank, reference per-frame order on every rank, no data-path collective" if args.sharding == "videos" else
                            f"runs of {args.gather_every} consecutive batches dealt round-robin to the ranks; their records gathered to rank 0 once "
                            "per run (RCCL), tracker replayed there in clip order on a second host thread",
            },
        }
        if failure is not None:
            out["error"] = failure + " -- the measurement is void (a rank that fails still joins the gathers, so nobody hangs)"
        if not args.no_profile:
            # kernel durations as they were inside the timed region: HIP events in front of every launch of
            # every --trace-every-th pass, on the stream the kernels were launched on (gtx_detector_trace)
            merged = {}
            for d in engine.dets:
                for f in d.trace_report():
                    m = merged.setdefault(f["kernel"], dict(kernel=f["kernel"], launches=0, total_ms=0.0, flops=0.0, bytes=0.0))
                    for key in ("launches", "total_ms", "flops", "bytes"):
                        m[key] += f[key]
                d.trace(0)
            fam = sorted(merged.values(), key=lambda d: -d["total_ms"])
            if fam:
                top = fam[0]
                tflops = top["flops"] / (top["total_ms"] * 1e-3) / 1e12
                gbs = top["bytes"] / (top["total_ms"] * 1e-3) / 1e9
                # which roof bounds the family: algorithmic intensity against the machine balance (peak FLOP/s / peak B/s)
                balance = MFMA_PEAK_TFLOPS[dt] * 1e12 / (HBM_PEAK_GBS * 1e9)
                hbm_bound = top["flops"] / max(top["bytes"], 1.0) < balance
                out["roofline"] = {"bound": "hbm" if hbm_bound else "mfma", "kernel": top["kernel"],
                                   "achieved": gbs if hbm_bound else tflops, "peak": HBM_PEAK_GBS if hbm_bound else MFMA_PEAK_TFLOPS[dt],
                                   "unit": "GB/s" if hbm_bound else "TFLOP/s",
                                   "frac": (gbs / HBM_PEAK_GBS) if hbm_bound else (tflops / MFMA_PEAK_TFLOPS[dt]),
                                   "traffic": pmc_traffic(top["kernel"], B),
                                   "traffic_source": "profiles/rNN_pmc_traffic.json (newest round): committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
                                                     "(tools/collect_profiles.sh), NOT measured in this run; null when no pass for this kernel and batch is committed",
                                   "avg_launch_us": 1000.0 * top["total_ms"] / top["launches"],
                                   "launches_timed": top["launches"],
                                   "flops_per_launch": top["flops"] / top["launches"], "bytes_per_launch": top["bytes"] / top["launches"],
                                   "intensity_flop_per_byte": top["flops"] / max(top["bytes"], 1.0), "machine_balance": balance,
                                   "tflops": tflops, "algo_gbs": gbs,
                                   "peak_note": ("dense fp16 MFMA peak 2500 TFLOP/s / 3 MFMAs per useful product" if dt == "f32s" else
                                                 "MI355X_MICROARCH.md: dense MFMA peak of the arithmetic type; HBM3E 8 TB/s"),
                                   "timing": f"HIP events around every launch of every {args.trace_every}th pass inside the timed region "
                                             "(durations include the time a launch shares the GPU with the other streams)"}
                # the same family with every launch alone on its stream (gtx_detector_profile: HIP events around each launch of a few
                # passes, nothing else running): what rocprofv3's per-kernel average for this command agrees with
                prof_all = None
                try:
                    prof_all = engine.dets[0].profile(B, 4)
                    iso = {f["kernel"]: f for f in prof_all}.get(top["kernel"])
                except Exception:
                    iso = None
                if iso and iso["total_ms"] > 0:
                    a_tf = iso["flops"] / (iso["total_ms"] * 1e-3) / 1e12
                    a_gb = iso["bytes"] / (iso["total_ms"] * 1e-3) / 1e9
                    out["roofline"]["alone"] = {"avg_launch_us": 1000.0 * iso["total_ms"] / iso["launches"], "achieved": a_gb if hbm_bound else a_tf,
                                                "frac": (a_gb / HBM_PEAK_GBS) if hbm_bound else (a_tf / MFMA_PEAK_TFLOPS[dt]),
                                                "note": "after the timed region, launches back to back on one stream with nothing else on the GPU"}
                # what the pipeline executed, against what the graph nominally costs: the per-launch FLOPs the runtime
                # reports are those of the rows a launch computes (letterbox-padding rows skipped) with the Detect box branch at the candidates only
                try:
                    pr = prof_all if prof_all is not None else engine.dets[0].profile(B, 2)
                    it = 4 if prof_all is not None else 2
                    ex_g = sum(f["flops"] for f in pr) / it / B / 1e9
                    nominal = NOMINAL_GFLOP.get((args.model, det.net_hw[0], det.net_hw[1]), ex_g if args.model != "yolov8s" else None)
                    fps = out["value"] / max(world, 1)
                    R = out["roofline"]
                    R["executed_gflop_per_frame"] = ex_g
                    R["nominal_gflop_per_frame"] = nominal
                    R["executed_over_nominal"] = (ex_g / nominal) if nominal else None
                    R["pipeline_tflops_executed"] = ex_g * fps / 1e3
                    R["pipeline_frac"] = ex_g * fps / 1e3 / MFMA_PEAK_TFLOPS[dt]
                    R["pipeline_frac_nominal"] = (nominal * fps / 1e3 / MFMA_PEAK_TFLOPS[dt]) if nominal else None
                    R["pipeline_note"] = "executed = sum of the launches' own FLOPs per frame (gtx_detector_profile) x frames/s per GPU; frac against the same MFMA roof as `peak`"
                except Exception as e:
                    out["roofline"]["pipeline_note"] = f"unavailable: {type(e).__name__}: {e}"
                if args.model == "rtdetr-l":           # the two transformer kernels named on their own (in-situ timings of this run)
                    for key, kname in (("attention", "rt_mha_kernel"), ("deformable_sampling", "rt_deform_kernel"), ("token_linear", "rt_linear_kernel")):
                        d = merged.get(kname) or (merged.get("rt_mha32_kernel") if key == "attention" else None)
                        if d and d["total_ms"] > 0:
                            kname = d["kernel"]
                            out["roofline"][key] = {"kernel": kname, "avg_launch_us": 1000.0 * d["total_ms"] / d["launches"], "launches_timed": d["launches"],
                                                    "tflops": d["flops"] / (d["total_ms"] * 1e-3) / 1e12, "algo_gbs": d["bytes"] / (d["total_ms"] * 1e-3) / 1e9,
                                                    "bound": {"attention": "fp32 matrix pipe (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s): S^T = K Q^T and O^T += V^T P^T per 16 x 16 tile, probabilities stay in their lanes", "deformable_sampling": "latency (300 queries x 96 bilinear gathers per image)",
                                                              "token_linear": "fp32 matrix pipe, v_mfma_f32_16x16x4_f32 (157.3 TFLOP/s); launch latency at 300 rows"}[key]}
                out["kernels"] = [{"kernel": d["kernel"], "launches_timed": d["launches"],
                                   "avg_launch_us": 1000.0 * d["total_ms"] / d["launches"],
                                   "tflops": (d["flops"] / (d["total_ms"] * 1e-3) / 1e12) if d["total_ms"] > 0 else 0.0,
                                   "algo_gbs": (d["bytes"] / (d["total_ms"] * 1e-3) / 1e9) if d["total_ms"] > 0 else 0.0}
                                  for d in fam]
        if world == 1 and dist is None and not args.half and not args.no_f16_line:
            # secondary key: the same workload with ultralytics.half = true (a legitimate reference knob, not its default). Run as
            # a child process after this one has released the GPU: a second engine inside this process inherits its hardware-queue
            # mapping from the streams created (and not yet destroyed) above and measures ~25 % low.
            import subprocess

            engine.close()
            ctx.synchronize()
            n16 = max(args.steps // 2, 10)
            cmd = [sys.executable, str(ROOT / "bench.py"), "--model", args.model, "--half", "1", "--steps", str(n16), "--warmup", str(min(args.warmup, 10)), "--no-cpu-baseline",
                   "--no-profile", "--workload", args.workload, "--tracker", args.tracker, "--batch", str(B), "--det-streams", str(args.det_streams),
                   "--stab-streams", str(args.stab_streams), "--frames", str(args.frames), "--detections", str(args.detections),
                   "--imgsz", str(args.imgsz), "--rect", str(args.rect)]
            try:
                p16 = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                d16 = json.loads([ln for ln in p16.stdout.splitlines() if ln.startswith("{")][-1])
                out["f16"] = {"value": d16["value"], "unit": "frames/s", "steps": d16["steps"], "ms_per_step": d16["ms_per_step"], "dtype": "f16",
                              "note": "same workload, weights and pipeline with ultralytics.half = true (fp16 activations, fp16 MFMA), measured by a child "
                                      "process of this run after the primary measurement; secondary, not `value`"}
            except Exception as e:                                  # the secondary line must never void the primary one
                out["f16"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if world == 1 and dist is None and not args.half and det.fp32_split and not args.no_f16_line:
            # secondary key: the strict fp32 arithmetic (v_mfma_f32_32x32x2_f32, no split) of the same workload, so that the
            # driver's line carries the exact-fp32 number next to the split-f16x3 one
            import subprocess

            nx = max(args.steps // 4, 10)
            cmd = [sys.executable, str(ROOT / "bench.py"), "--model", args.model, "--fp32", "exact", "--steps", str(nx), "--warmup", str(min(args.warmup, 6)), "--no-cpu-baseline",
                   "--no-profile", "--no-f16-line", "--workload", args.workload, "--tracker", args.tracker, "--batch", str(B), "--det-streams", str(args.det_streams),
                   "--stab-streams", str(args.stab_streams), "--frames", str(args.frames), "--detections", str(args.detections),
                   "--imgsz", str(args.imgsz), "--rect", str(args.rect)]
            try:
                px = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                dx = json.loads([ln for ln in px.stdout.splitlines() if ln.startswith("{")][-1])
                out["f32_exact"] = {"value": dx["value"], "unit": "frames/s", "steps": dx["steps"], "ms_per_step": dx["ms_per_step"], "dtype": dx["dtype"],
                                    "note": "same workload with the exact-fp32 MFMA convolutions (fp32_split = 0), measured by a child process of this run "
                                            "after the primary measurement; secondary, not `value`"}
            except Exception as e:
                out["f32_exact"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if world == 1 and dist is None and args.model == "yolov8s" and not args.half and args.workload == "extract" and args.tracker == "bytetrack" and not args.no_f16_line \
                and not args.candidates:
            # secondary key: the same pipeline under the post-processing load SURVEY 8d names (1-3 k anchors above conf per frame). The
            # seeded weights cannot cluster them the way a trained model does (their boxes sit on their anchors: neighbours overlap at
            # IoU ~0.5, NMS keeps every second one), so the heavy end is measured as what it is: 2 000 candidates, ~1 000 boxes per frame
            # through NMS, tracker (1 000 detections per update) and box warp.
            import subprocess

            cmd = [sys.executable, str(ROOT / "bench.py"), "--candidates", "2000", "--steps", str(max(args.steps // 4, 10)), "--warmup", str(min(args.warmup, 6)),
                   "--no-cpu-baseline", "--no-profile", "--no-f16-line", "--workload", args.workload, "--tracker", args.tracker, "--batch", str(B),
                   "--det-streams", str(args.det_streams), "--stab-streams", str(args.stab_streams), "--frames", str(args.frames), "--imgsz", str(args.imgsz), "--rect", str(args.rect)]
            try:
                pn = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                dn = json.loads([ln for ln in pn.stdout.splitlines() if ln.startswith("{")][-1])
                out["nms_load"] = {"value": dn["value"], "unit": "frames/s", "steps": dn["steps"], "ms_per_step": dn["ms_per_step"],
                                   "candidates_per_frame": dn["config"]["candidates_per_frame"], "detections_per_frame": dn["config"]["detections_per_frame"],
                                   "nms_path": dn["config"]["nms_path"],
                                   "note": "same pipeline with the class bias set for 2 000 anchors above conf per frame (SURVEY 8d's range is 1-3 k): decode, NMS, "
                                           "tracker and box warp at ~8 x the golden clip's box count; measured by a child process, secondary, not `value`"}
            except Exception as e:
                out["nms_load"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if world == 1 and dist is None and args.model == "yolov8s" and not args.half and args.workload == "extract" and args.tracker == "bytetrack" and not args.no_f16_line \
                and os.environ.get("GTX_PAD_SKIP", "1") != "0":
            # secondary key: the same line with every row of every layer and every pixel of the Detect box branch computed in every pass
            import subprocess

            cmd = [sys.executable, str(ROOT / "bench.py"), "--steps", str(max(args.steps, 100)), "--warmup", str(args.warmup), "--no-cpu-baseline",
                   "--no-profile", "--no-f16-line", "--workload", args.workload, "--tracker", args.tracker, "--batch", str(B), "--det-streams", str(args.det_streams),
                   "--stab-streams", str(args.stab_streams), "--frames", str(args.frames), "--detections", str(args.detections),
                   "--imgsz", str(args.imgsz), "--rect", str(args.rect)]
            try:
                pe = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env={**os.environ, "GTX_PAD_SKIP": "0", "GTX_SPARSE_BOX": "0"})
                de = json.loads([ln for ln in pe.stdout.splitlines() if ln.startswith("{")][-1])
                out["every_row_every_pass"] = {"value": de["value"], "unit": "frames/s", "steps": de["steps"], "ms_per_step": de["ms_per_step"],
                                               "note": "GTX_PAD_SKIP=0 GTX_SPARSE_BOX=0: the letterbox-padding rows recomputed in every pass and the Detect box "
                                                       "branch evaluated at every anchor, as before round 5's last two changes; same detections bit for bit "
                                                       "(tests/test_detector_gpu.py); measured by a child process, secondary, not `value`"}
            except Exception as e:
                out["every_row_every_pass"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
            if "roofline" in out:                                 # next to executed / nominal: the frame rate when every nominal FLOP is executed
                out["roofline"]["value_every_row_every_pass"] = out["every_row_every_pass"].get("value")
        if world == 1 and dist is None and args.model == "yolov8s" and not args.half and args.workload == "extract" and args.tracker == "bytetrack" and not args.no_f16_line:
            # secondary key: the N > 1 default workload (BoT-SORT + GPU GMC, BASELINE configs[4]) on this one GPU, so that a scaling
            # series started from this line has its like-for-like single-GPU base in it (the primary line here is configs[2], ByteTrack)
            import subprocess

            cmd = [sys.executable, str(ROOT / "bench.py"), "--tracker", "botsort", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline",   # the primary line's own length: this number is a base others are divided by
                   "--no-profile", "--no-f16-line", "--workload", args.workload, "--batch", str(B), "--det-streams", str(args.det_streams),
                   "--stab-streams", str(args.stab_streams), "--frames", str(args.frames), "--detections", str(args.detections),
                   "--imgsz", str(args.imgsz), "--rect", str(args.rect)]
            try:
                pb = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                db = json.loads([ln for ln in pb.stdout.splitlines() if ln.startswith("{")][-1])
                out["botsort"] = {"value": db["value"], "unit": "frames/s", "steps": db["steps"], "ms_per_step": db["ms_per_step"], "dtype": db["dtype"],
                                  "note": "same pipeline with BoT-SORT + sparse-optical-flow GMC on the GPU, the default workload of `--gpus N` for N > 1 "
                                          "(BASELINE configs[4]): divide an N-GPU `value` by N times THIS number for a like-for-like scaling efficiency; "
                                          "measured by a child process of this run, secondary, not `value`"}
            except Exception as e:
                out["botsort"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if world == 1 and dist is None and args.model == "yolov8s" and not args.half and args.workload == "extract" and args.tracker == "bytetrack" and not args.no_f16_line \
                and not args.host_frames:
            # secondary keys: what the same pipeline delivers when the frames do not start in HBM. `host_frames`: pageable host arrays,
            # uploaded inside the timed region (PCIe-inclusive); `from_file`: the product's own loop on a clip file on local disk
            # (--workload cli: reader + PCIe + GPU + tracker, wall clock). Child processes, like the keys above; never `value`.
            import subprocess

            common = ["--no-cpu-baseline", "--no-profile", "--no-f16-line", "--batch", str(B), "--det-streams", str(args.det_streams), "--stab-streams",
                      str(args.stab_streams), "--frames", str(args.frames), "--detections", str(args.detections), "--imgsz", str(args.imgsz), "--rect", str(args.rect)]
            if args.fp32 is not None:
                common += ["--fp32", args.fp32]
            try:
                ph = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--host-frames", "--steps", str(max(args.steps, 100)), "--warmup", str(max(args.warmup, 10))] + common,
                                    capture_output=True, text=True, timeout=600)
                dh = json.loads([ln for ln in ph.stdout.splitlines() if ln.startswith("{")][-1])
                out["host_frames"] = {"value": dh["value"], "unit": "frames/s", "steps": dh["steps"], "ms_per_step": dh["ms_per_step"],
                                      "note": "same pipeline fed 3840x2160 frames from pageable host memory (24.9 MB per frame uploaded inside the timed region): "
                                              "the PCIe-inclusive rate; measured by a child process over its own `steps` (at least 100: with fewer timed steps than that, `value`'s region is shorter and "
                                              "the pipeline's fill and drain -- ~3 ms -- weigh more in it than here), secondary, not `value`"}
            except Exception as e:
                out["host_frames"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
            try:
                pf = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "cli", "--cli-formats", "y4m", "--cli-compare-sync", "0"] + common[3:],
                                    capture_output=True, text=True, timeout=900)
                df = json.loads([ln for ln in pf.stdout.splitlines() if ln.startswith("{")][-1])
                y = df["from_file"]["y4m"]
                out["from_file"] = {"value": y["frames_per_s"], "unit": "frames/s", "frames": df["steps"] * B, "container": ".y4m (I420, page-cached)",
                                    "reference_convention_fps": y["reference_convention_fps"], "file_gbs": y["file_gbs"],
                                    "note": "geotrax_amd.extract.track_with_model on a 150-frame 3840x2160 .y4m on local disk (bench.py --workload cli): wall clock from the "
                                            "open file to the aggregated tables, read-ahead feeder; measured by a child process, secondary, not `value`"}
            except Exception as e:
                out["from_file"] = {"value": None, "error": f"{type(e).__name__}: {e}"}
        if world == 1 and dist is None and args.model == "yolov8s" and not extract and B == 1 and args.det_streams == 1 and not args.no_f16_line:
            # BASELINE configs[1] is "batch=1": one frame per pass. With ONE pass in flight a pass is ~46 short launches and the chip
            # idles between them (latency-bound: `value`); with two or three single-frame passes in flight on streams of their own
            # the same kernels fill each other's gaps. Secondary key, child processes.
            import subprocess

            out["pipelined"] = {"note": "the same single-frame passes with 2 and 3 of them in flight (detector streams), frames/s; secondary, not `value`"}
            for ns in (2, 3):
                cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", "detect", "--batch", "1", "--det-streams", str(ns), "--steps", str(args.steps), "--warmup",
                       str(args.warmup), "--no-cpu-baseline", "--no-profile", "--no-f16-line", "--frames", str(args.frames), "--detections", str(args.detections),
                       "--imgsz", str(args.imgsz), "--rect", str(args.rect)] + (["--fp32", args.fp32] if args.fp32 else [])
                try:
                    pp = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                    out["pipelined"][f"{ns}_streams"] = json.loads([ln for ln in pp.stdout.splitlines() if ln.startswith("{")][-1])["value"]
                except Exception as e:
                    out["pipelined"][f"{ns}_streams"] = None
                    out["pipelined"]["error"] = f"{type(e).__name__}: {e}"
        if world == 1 and dist is None and "roofline" in out and not args.no_live_traffic and not args.no_f16_line and not args.host_frames:
            # roofline.traffic measured in THIS run: the engine has been closed above, the children run alone
            try:
                engine.close()
            except Exception:
                pass
            tr = live_traffic(out["roofline"]["kernel"], args, B)
            if tr:
                out["roofline"]["traffic_committed_summary"] = out["roofline"]["traffic"]
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_source"] = ("measured in this run: child passes `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate, --kernel-trace only) of "
                                                     "this command at 6 steps; (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes averaged over the kernel's dispatches; "
                                                     "`traffic_committed_summary` = the same quantity from profiles/rNN_pmc_traffic.json")
                out["roofline"]["traffic_over_algorithmic"] = tr / out["roofline"]["bytes_per_launch"]
        out["host"] = {"cores": host_cores(), "threads_per_rank": engine.host_threads + (1 if (sharded and extract and rank == 0) else 0) + 1,
                       "note": "engine stage threads (blocking waits: they sleep while the GPU works) + the main thread" +
                               (" + rank 0's tracker replay thread" if (sharded and extract) else "")}
        if not args.no_cpu_baseline and world == 1:             # rank 0 at N = 1 only (the other ranks would sit in the barrier below)
            out["cpu_baseline"] = cpu_baseline(weights, ref_frame, frames[1], args)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
