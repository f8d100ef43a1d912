"""bench.py end to end on the GPU box: the single-GPU line carries the contract's fields, and the N > 1 code paths
(frames of one clip sharded over ranks with the record gather and the tracker replay; one clip per rank) run to
completion with two ranks sharing the one GPU over gloo."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _last_json(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_single_gpu_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "120", "--warmup", "8", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 120 and d["value"] > 100 and d["unit"] == "frames/s" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # roofline.traffic is measured by this very run (two rocprofv3 --pmc child passes) and sits near the algorithmic bytes
    assert r["traffic_source"].startswith("measured in this run") and 0.8 < r["traffic_over_algorithmic"] < 1.6, r.get("traffic_source")
    # the default line is the reference's own precision (ultralytics.half: false, default.yaml:245); fp16 rides along
    assert d["dtype"].startswith("f32") and d["config"]["half"] is False
    assert d["f16"]["dtype"] == "f16" and d["f16"]["value"] > d["value"]
    # ... and so do the strict fp32 arithmetic and the N > 1 default workload (BoT-SORT + GMC: the like-for-like base of a scaling series)
    assert 0 < d["f32_exact"]["value"] < d["value"]
    assert d["botsort"]["value"] > 100 and d["botsort"]["dtype"] == d["dtype"]
    # what the pipeline delivers when the frames do not start in HBM rides along too: `host_frames` = pageable host
    # arrays (24.9 MB BGR each) uploaded inside the timed region, `from_file` = the product's own loop on a 150-frame .y4m (12.4 MB
    # I420 per frame) on local disk. Ordered: neither beats the resident rate (10 % slack for run-to-run noise) and both stay within
    # a quarter of it; between themselves they are not ordered (the file carries half the bytes per frame across PCIe)
    hf, ff = d["host_frames"]["value"], d["from_file"]["value"]
    assert hf > 100 and ff > 100 and d["from_file"]["frames"] == 150 and d["from_file"]["reference_convention_fps"] > 0
    assert ff <= 1.1 * d["value"] and hf <= 1.1 * d["value"], (d["value"], hf, ff)
    # with the engine's stream plan (every detector stream on a hardware queue of its own, whatever the process created before)
    # both sit within a few per cent of the resident rate; 0.85 leaves room for box-to-box noise (VERDICT r03's bar is 0.9 x 0.9 = 0.81)
    assert ff >= 0.85 * d["value"] and hf >= 0.85 * d["value"], (d["value"], hf, ff)
    # NMS sees clustered candidates: more candidates than detections in the calibration frame
    assert d["config"]["candidates_per_frame"] > 1.3 * d["config"]["detections_per_frame"]
    # ... and the heavy end of SURVEY 8d's post-processing range rides along: 2 000 anchors above conf per frame
    nl = d["nms_load"]
    assert 1800 <= nl["candidates_per_frame"] <= 2200 and nl["detections_per_frame"] > 400 and nl["value"] > 0.5 * d["value"], nl
    assert "nms_small" in nl["nms_path"] and "nms_small" in d["config"]["nms_path"]


@pytest.mark.parametrize("mode,tracker", [("frames", "bytetrack"), ("frames", "botsort"), ("videos", "bytetrack")])
def test_two_ranks_on_one_gpu(mode, tracker):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    tail = ["--gpus", "2", "--steps", "18", "--warmup", "2", "--gather-every", "4",
            "--no-cpu-baseline", "--no-profile", "--backend", "gloo", "--sharding", mode, "--tracker", tracker]
    if tracker == "bytetrack" and mode == "frames":
        # the driver's own spelling, no launcher: bench.py starts the two ranks itself (a child torch.distributed.run, before
        # the parent touches HIP) and forwards rank 0's line
        env = {k: v for k, v in env.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        cmd = [sys.executable, str(ROOT / "bench.py")] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29541", str(ROOT / "bench.py")] + tail
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 18 and d["value"] > 50 and d["scaling"] == "weak"
    assert ("one clip per rank" in d["config"]["sharding"]) == (mode == "videos")
    if tracker == "botsort":
        assert "GMC" in d["config"]["tracker"]


def test_configs3_extract_then_georeference_as_one_chained_run():
    """BASELINE configs[3] inside the GPU test tier: the 150-frame 4K clip through extract, straight into
    the georeference stage (RootSIFT registration against the synthetic orthophoto, row chain, CSV; geotrax/georeference.py:109-202)
    as one run, also paced as a 30 fps stream. Asserts the registration error against the known orthophoto camera, the row
    counts of both stages and that the stream keeps 30 fps."""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "extract+georef", "--steps", "75", "--warmup", "2", "--no-cpu-baseline",
                        "--no-profile"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _last_json(p.stdout)
    c = d["config"]
    assert "configs[3]" in c["workload"] and d["n_gpus"] == 1 and d["dtype"].startswith("f32")
    assert c["frames"] == 150 and c["track_rows"] > 100 * c["frames"] and 0.9 * c["track_rows"] < c["csv_rows"] <= c["track_rows"] and c["vehicles"] > 100
    reg = c["registration"]
    assert reg["inliers"] > 500 and reg["inliers"] > 0.9 * reg["matches"] and reg["max_grid_error_px_vs_known_orthophoto"] < 0.25
    assert d["value"] > 100
    assert d["paced"]["stream_fps"] == 30.0 and d["paced"]["sustained_fps"] > 29.5     # 150 frames arriving at 30 fps leave the pipeline at 30 fps
    # ... and soon after they arrive: a paced source makes the engine hand on everything in flight before it waits for the next batch and
    # take the stabilizers' pending frames while nothing comes in. Counted from each frame's own arrival (the first frame of a batch of 2
    # waits 33 ms for the second): ~24 ms median; it was 8 frame periods of pipeline depth.
    assert d["paced"]["frame_latency_ms"]["median"] < 60 and d["paced"]["frame_latency_ms"]["max"] < 120, d["paced"]


def test_configs1_detector_only_batch1_one_stream():
    """BASELINE configs[1]: YOLOv8s HIP inference only on 3840x2160 frames, batch 1, one stream (`value`: one pass in flight,
    latency-bound); the same single-frame passes with two and three in flight ride along under `pipelined`."""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "detect", "--batch", "1", "--det-streams", "1", "--steps", "60",
                        "--warmup", "10", "--no-cpu-baseline", "--no-profile"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _last_json(p.stdout)
    c = d["config"]
    assert "configs[1]" in c["workload"] and c["frames_per_step"] == 1 and c["net_input"] == [1920, 1920] and c["half"] is False
    assert d["steps"] == 60 and d["value"] > 200 and c["detections_per_frame"] > 100
    assert d["pipelined"]["2_streams"] > d["value"] and d["pipelined"]["3_streams"] > d["value"]


def test_bench_distributed_code_path_over_rccl_with_one_rank():
    """bench.py's N > 1 path (RCCL weight broadcast, per-run record gathers on a side stream, tracker replay thread, max-over-
    ranks timing) with ONE rank on the nccl backend: RCCL cannot put two ranks on one GPU, this is what a one-GPU box can run
    of it (GTX_BENCH_FORCE_DIST=1)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GTX_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29543", str(ROOT / "bench.py"), "--gpus", "1", "--steps", "16", "--warmup", "2", "--gather-every", "4",
           "--no-cpu-baseline", "--no-profile", "--no-f16-line", "--backend", "nccl", "--tracker", "botsort"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 16 and d["value"] > 100 and "error" not in d and "GMC" in d["config"]["tracker"]


def test_eight_ranks_on_one_gpu_botsort():
    """BASELINE configs[4]'s rank count and tracker: 8 ranks sharing the one GPU over gloo, a short last run (9 steps, gathers
    every 2): the record order / GMC priming arithmetic for world == 8 executes and the line is whole."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29545", str(ROOT / "bench.py"), "--gpus", "8", "--steps", "9", "--warmup", "1", "--gather-every", "2",
           "--no-cpu-baseline", "--no-profile", "--backend", "gloo", "--tracker", "botsort", "--det-streams", "1", "--stab-streams", "1", "--frames", "4"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 8 and d["steps"] == 9 and d["value"] > 20 and "error" not in d and d["scaling"] == "weak"
    assert d["host"]["threads_per_rank"] >= 3 and d["host"]["cores"] >= 1


def test_cli_workload_from_a_file_is_deterministic():
    """bench.py --workload cli on a shorter clip: the product's loop on a .y4m and a .npy, read-ahead feeder and synchronous
    reader; four runs per container produce the same tables byte for byte."""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--workload", "cli", "--cli-frames", "40"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _last_json(p.stdout)
    assert d["unit"] == "frames/s" and d["value"] > 50 and set(d["from_file"]) == {"y4m", "npy"}
    for fmt, r in d["from_file"].items():
        assert r["deterministic"] is True, fmt
        assert r["track_rows"] > 1000 and r["transforms"] == 39 and r["synchronous_reader"]["frames_per_s"] > 20
