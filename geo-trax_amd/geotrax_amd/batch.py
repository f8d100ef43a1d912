"""`geotrax batch` for the extract stage, mapped to one process per GPU.

Reference behaviour replaced: geotrax/batch_process.py:220-377 -- walk a directory for videos, drop the
ones in excluded folders / matching excluded patterns, skip what already has results unless
--overwrite (asking unless --yes), run the stages per file, never let one file's failure stop the
batch. Here the per-file stage is `geotrax_amd.extract.detect_track_stabilize` (the GPU hot path);
georeference / visualisation / plots of the reference's batch are outside this build and are not run.

Multi-GPU: started under `python -m torch.distributed.run --nproc-per-node N -m geotrax_amd.batch <dir>`
every rank scans the same sorted file list and takes the videos `i % N == rank` of the largest-first
order (the honest best case of §8e: one video per GPU, no collective on the data path); rank 0 prints the
summary after a barrier. Without a launcher it is a single process on GPU 0.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
from pathlib import Path

from . import __version__
from .config_utils import backfill_args_from_config, load_config
from .extract import add_common_args, add_processing_args, detect_track_stabilize, get_output_dir

VIDEO_FORMATS = {'.mp4', '.mov', '.avi', '.mkv', '.npy', '.y4m'}  # constants.py:10 plus this build's array / uncompressed clips
ACTION_EXTRACT = "Detecting, tracking, and stabilizing"


def discover(input_path: Path, folders_exclude, exclude_patterns, logger: logging.Logger) -> list[Path]:
    """Sorted videos below `input_path` after the two exclusion rules (batch_process.py:245-248, 325-341)."""
    files = []
    for f in input_path.rglob('*'):
        if not (f.is_file() and f.suffix.lower() in VIDEO_FORMATS):
            continue
        if f.parent.name in (folders_exclude or []):
            logger.info(f"Skipping '{f}' as it's in an excluded folder.")
            continue
        if exclude_patterns and any(p in f.name for p in exclude_patterns):
            logger.info(f"Skipping '{f}' due to matching exclusion pattern.")
            continue
        files.append(f)
    return sorted(files)


def results_exist(file: Path, out_cfg: dict) -> bool:
    """The processed-results file of `file` (file_utils.check_if_results_exist(file, 'processed'))."""
    return (get_output_dir(file, out_cfg) / f"{file.stem}{out_cfg.get('tracks_postfix', '')}.txt").exists()


def handle_existing_results(file: Path, args, logger, exists: bool, action: str, ask=input) -> bool:
    """batch_process.py:366-376: skip unless --overwrite; ask unless --yes."""
    if exists and not args.overwrite:
        logger.warning(f"'{file}' - {action} results already exist and overwrite not allowed.")
        return False
    if exists and args.overwrite and not args.yes:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:           # N ranks cannot all prompt on one terminal
            logger.warning(f"Skipping '{file}': --overwrite under a multi-rank launcher needs --yes")
            return False
        return ask(f"Overwrite {action} results for: '{file}'? [y/n]: ").lower() == 'y'
    return True


def shard(files: list[Path], rank: int, world: int) -> list[Path]:
    """Largest first, dealt round-robin: ranks finish close together without talking to each other."""
    order = sorted(files, key=lambda f: (-f.stat().st_size, str(f)))
    return [f for i, f in enumerate(order) if i % world == rank]


def process_file(file: Path, args, logger, out_cfg: dict, run=detect_track_stabilize) -> str:
    """-> 'done' | 'skipped' | 'failed' | 'dry'. One file's failure never stops the batch (batch_process.py:300-303)."""
    try:
        logger.info(f"Processing: '{file}'")
        if not handle_existing_results(file, args, logger, results_exist(file, out_cfg), "detection, tracking, and stabilization"):
            return 'skipped'
        logger.info(f"{ACTION_EXTRACT}: '{file}'")
        if args.dry_run:
            return 'dry'
        file_args = argparse.Namespace(**vars(args))
        file_args.source = file
        run(file_args, logger)
        return 'done'
    except Exception as e:
        logger.error(f"Error with {file}: {e}")
        return 'failed'
    except SystemExit as e:
        # the per-file stage exits on an unreadable video or a missing model (extract.py load_detector /
        # initialize_streams, like the reference); inside a batch -- and above all under a launcher, where the other
        # ranks wait in the closing all_reduce -- that is this file's failure, not the batch's
        logger.error(f"Error with {file}: the extraction stage exited with status {e.code}")
        return 'failed'


def process_input(args, logger: logging.Logger, run=detect_track_stabilize) -> dict:
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    input_path = Path(args.input)
    counts = dict(done=0, skipped=0, failed=0, dry=0)
    if not input_path.exists():
        logger.critical(f"File or directory '{input_path}' not found.")
        return counts
    cfg = load_config(args.cfg, logger)
    out_cfg_raw = cfg.get('output', {}) or {}
    batch_cfg = cfg.get('batch', {}) or {}
    backfill_args_from_config(args, {
        'folders_exclude': batch_cfg.get('folders_exclude', ['results']),
        'exclude_patterns': batch_cfg.get('exclude_patterns'),
        'output_folder': out_cfg_raw.get('folder', 'results'),
    })
    out_cfg = {**out_cfg_raw, 'folder': args.output_folder}
    if input_path.is_file():
        files = [input_path] if input_path.suffix.lower() in VIDEO_FORMATS else []
    else:
        logger.info(f"Batch processing all videos in: '{input_path}'")
        args.cut_frame_right = None
        files = discover(input_path, args.folders_exclude, args.exclude_patterns, logger)
    mine = shard(files, rank, world) if world > 1 else files
    if world > 1:
        os.environ.setdefault("GTX_DEVICE", os.environ.get("LOCAL_RANK", "0"))
        os.environ["GTX_FRAME_SHARDING"] = "0"                    # whole videos per rank here, not frames of one video (extract.py)
        logger.info(f"rank {rank}/{world}: {len(mine)} of {len(files)} videos")
    for f in mine:
        counts[process_file(f, args, logger, out_cfg, run)] += 1
    if world > 1:
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            dist.init_process_group("gloo")                       # control plane only: a barrier and four counters
        t = torch.tensor([counts[k] for k in ('done', 'skipped', 'failed', 'dry')], dtype=torch.int64)
        dist.all_reduce(t)
        counts = dict(zip(('done', 'skipped', 'failed', 'dry'), (int(v) for v in t)))
        dist.barrier()
    if rank == 0:
        logger.info(f"Batch finished: {counts['done']} processed, {counts['skipped']} skipped, {counts['failed']} failed"
                    + (f", {counts['dry']} listed (dry run)" if counts['dry'] else "") + ".")
    return counts


def parse_cli_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(prog="geotrax-amd batch", description=f"geo-trax_amd {__version__}: batch extraction over a directory of videos")
    ap.add_argument("input", type=Path, help="video file or directory (searched recursively)")
    add_common_args(ap)
    add_processing_args(ap)
    g = ap.add_argument_group("batch")
    g.add_argument("--overwrite", "-o", action="store_true", help="re-process videos that already have results")
    g.add_argument("--yes", "-y", action="store_true", help="do not ask before overwriting")
    g.add_argument("--dry-run", "-dr", action="store_true", help="list what would be processed")
    g.add_argument("--folders-exclude", nargs="*", default=None, help="sub-folder names to skip (cfg -> batch -> folders_exclude)")
    g.add_argument("--exclude-patterns", nargs="*", default=None, help="skip videos whose name contains any of these")
    return ap.parse_args(argv)


def main(argv=None) -> None:
    args = parse_cli_args(argv)
    logging.basicConfig(level=logging.DEBUG if getattr(args, "verbose", False) else logging.INFO, format="%(levelname)s %(message)s")
    process_input(args, logging.getLogger("geotrax_amd.batch"))


if __name__ == "__main__":
    main(sys.argv[1:])
