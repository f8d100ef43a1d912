"""`geotrax batch`: the stage chain of the reference over a directory of videos, one process per GPU.

Reference behaviour replaced: geotrax/batch_process.py:220-377 -- walk a directory for videos, drop the ones in excluded
folders / matching excluded patterns, and per file run the stages one after the other, EACH behind its own skip-if-exists
rule (`should_process_file`, :340-364): detection + tracking + stabilization (`geotrax_amd.extract.detect_track_stabilize`,
skipped when the tracks file exists) and then georeferencing (`geotrax_amd.georef_stage.georeference`, needs the tracks
file, skipped when the CSV exists); `--overwrite` re-runs a stage (asking unless `--yes`), `--no-geo` / `--geo-only` select
stages, `--dry-run` lists them; one file's failure never stops the batch. Visualisation and plots of the reference's batch
are not built (out of scope): `--viz-only` / `--plot-only` say so and do nothing.

Multi-GPU: started under `python -m torch.distributed.run --nproc-per-node N -m geotrax_amd.batch <dir>` every rank scans
the same file list and the videos are handed out from a SHARED COUNTER, largest first: a rank that finishes takes the next
video (the counter is an atomic add on the launcher's rendezvous store, which rank 0 serves) -- no rank idles for longer
than the last video it did not get. No collective on the data path (the honest best case of 8e: one video per GPU); rank 0
prints the summary after a barrier. Without a launcher it is a single process on GPU 0.
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
ACTION_GEOREF = "Georeferencing"
PROCESSING_STEPS = "detection, tracking, and stabilization"


def _georeference(args, logger):
    from .georef_stage import georeference

    georeference(args, logger)


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


def georeferenced_exist(file: Path, out_cfg: dict) -> bool:
    """check_if_results_exist(file, 'georeferenced')."""
    from .georef_stage import build_result_path

    return build_result_path(file, "georeferenced", out_cfg).exists()


def should_process_file(file: Path, args, logger, action: str, out_cfg: dict, ask=input) -> bool:
    """batch_process.py:340-364: a stage runs when its own output is missing (or --overwrite allows); georeferencing also
    needs the extraction stage's tracks file."""
    txt_exists = results_exist(file, out_cfg)
    if action == ACTION_EXTRACT:
        return handle_existing_results(file, args, logger, txt_exists, PROCESSING_STEPS, ask)
    if action == ACTION_GEOREF:
        if not txt_exists and not getattr(args, "dry_run", False):
            logger.error(f"'{file}' - No {PROCESSING_STEPS} results found. Skipping georeferencing.")
            return False
        return handle_existing_results(file, args, logger, georeferenced_exist(file, out_cfg), action, ask)
    return False


def order_largest_first(files: list[Path]) -> list[Path]:
    return sorted(files, key=lambda f: (-f.stat().st_size, str(f)))


def shard(files: list[Path], rank: int, world: int) -> list[Path]:
    """The static deal (largest first, round-robin): what a rank gets when no shared counter is available."""
    return [f for i, f in enumerate(order_largest_first(files)) if i % world == rank]


_queue_calls = 0


class WorkQueue:
    """Videos handed out one at a time, largest first, from a counter every rank of the job adds to atomically (the launcher's
    rendezvous store: `store.add` on rank 0's TCP store). A rank takes the next video when it has finished its last one, so
    unequal videos even out by themselves; with the static deal a rank that drew the long ones finished last while the others
    idled."""

    def __init__(self, files: list[Path], store, key: str = "gtx_batch_next", rank: int = 0):
        """The counter hands out INDICES, so every rank must index the same list: rank 0 publishes its ordered list through the
        store and the others take that one instead of their own scan (a file still being copied, an output landing in the scanned
        tree or an attribute cache would otherwise give ranks different orders -- videos done twice, others never -- silently).
        A rank whose own scan disagrees says so."""
        import json

        mine = order_largest_first(files)
        if rank == 0:
            store.set(key + "_list", json.dumps([str(f) for f in mine]))
        else:
            agreed = [Path(p) for p in json.loads(bytes(store.get(key + "_list")).decode())]     # blocks until rank 0 has published
            if agreed != mine:
                logging.getLogger(__name__).warning(f"rank {rank}: this rank's scan found {len(mine)} videos in another order or number than rank 0's "
                                                    f"{len(agreed)}; working from rank 0's list")
            mine = agreed
        self.files, self.store, self.key = mine, store, key

    def __iter__(self):
        while True:
            i = int(self.store.add(self.key, 1)) - 1
            if i >= len(self.files):
                return
            yield self.files[i]


def process_file(file: Path, args, logger, out_cfg: dict, run=detect_track_stabilize, run_geo=_georeference, stats: dict | None = None) -> str:
    """The stage chain of one file (batch_process.py:288-307). -> 'done' (a stage ran) | 'skipped' (every stage had its results)
    | 'failed' | 'dry'. `stats` (optional) counts per stage. One file's failure never stops the batch (:300-303)."""
    stats = stats if stats is not None else {}

    def note(stage, what):
        stats.setdefault(stage, dict(done=0, skipped=0, failed=0, dry=0))[what] += 1

    geo_only, no_geo = bool(getattr(args, "geo_only", False)), bool(getattr(args, "no_geo", False))
    if getattr(args, "viz_only", False) or getattr(args, "plot_only", False):
        logger.warning(f"'{file}': --viz-only / --plot-only: visualisation and plots are not part of this build; nothing to do.")
        return 'skipped'
    stages = ([] if geo_only else [("extract", ACTION_EXTRACT, run)]) + ([] if no_geo else [("georef", ACTION_GEOREF, run_geo)])
    ran = dry = False
    stage = "extract"
    try:
        logger.info(f"Processing: '{file}'")
        for stage, action, func in stages:
            if not should_process_file(file, args, logger, action, out_cfg):
                note(stage, 'skipped')
                continue
            logger.info(f"{action}: '{file}'")
            if args.dry_run:
                note(stage, 'dry')
                dry = True
                continue
            file_args = argparse.Namespace(**vars(args))
            file_args.source = file
            func(file_args, logger)
            note(stage, 'done')
            ran = True
        return 'done' if ran else ('dry' if dry else 'skipped')
    except Exception as e:
        note(stage, 'failed')
        logger.error(f"Error with {file}: {e}")
        return 'failed'
    except SystemExit as e:
        # a stage exits on an unreadable video, a missing model or missing orthophotos (extract.py load_detector /
        # initialize_streams, georef_stage, like the reference); inside a batch -- and above all under a launcher, where the
        # other ranks wait in the closing all_reduce -- that is this file's failure, not the batch's
        note(stage, 'failed')
        logger.error(f"Error with {file}: the {stage} stage exited with status {e.code}")
        return 'failed'


def process_input(args, logger: logging.Logger, run=detect_track_stabilize, run_geo=_georeference, stats: dict | None = None) -> dict:
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
    mine = files
    if world > 1:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("GTX_DEVICE", os.environ.get("LOCAL_RANK", "0"))
        os.environ["GTX_FRAME_SHARDING"] = "0"                    # whole videos per rank here, not frames of one video (extract.py)
        if not dist.is_initialized():
            dist.init_process_group("gloo")                       # control plane only: the shared counter, a barrier and four counters
        store = None
        if os.environ.get("GTX_BATCH_QUEUE", "1") != "0":
            try:
                store = dist.distributed_c10d._get_default_store()
            except Exception:                                     # a torch without that accessor: the static deal
                store = None
        global _queue_calls                                       # one counter per call: every rank calls process_input the same number of times
        _queue_calls += 1
        mine = WorkQueue(files, store, key=f"gtx_batch_next_{_queue_calls}", rank=rank) if store is not None else shard(files, rank, world)
        logger.info(f"rank {rank}/{world}: {len(files)} videos, " + ("taken from the shared counter, largest first" if store is not None else f"{len(mine)} dealt to this rank"))
    taken = []
    for f in mine:
        taken.append(f)
        counts[process_file(f, args, logger, out_cfg, run, run_geo, stats)] += 1
    if world > 1:
        t = torch.tensor([counts[k] for k in ('done', 'skipped', 'failed', 'dry')], dtype=torch.int64)
        dist.all_reduce(t)
        counts = dict(zip(('done', 'skipped', 'failed', 'dry'), (int(v) for v in t)))
        dist.barrier()
    if stats is not None:
        stats["taken"] = taken
    if rank == 0:
        logger.info(f"Batch finished: {counts['done']} processed, {counts['skipped']} skipped, {counts['failed']} failed"
                    + (f", {counts['dry']} listed (dry run)" if counts['dry'] else "") + ".")
    return counts


def parse_cli_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser(prog="geotrax-amd batch", description=f"geo-trax_amd {__version__}: batch extraction over a directory of videos")
    ap.add_argument("input", type=Path, help="video file or directory (searched recursively)")
    add_common_args(ap)
    add_processing_args(ap)
    from .georef_stage import add_georeferencing_args

    add_georeferencing_args(ap.add_argument_group("georeferencing"))
    g = ap.add_argument_group("batch")
    g.add_argument("--overwrite", "-o", action="store_true", help="re-process videos that already have results")
    g.add_argument("--yes", "-y", action="store_true", help="do not ask before overwriting")
    g.add_argument("--dry-run", "-dr", action="store_true", help="list which files and stages would be processed")
    g.add_argument("--viz-only", "-vo", action="store_true", help="(reference flag) visualisation is not part of this build: nothing is run")
    g.add_argument("--geo-only", "-go", action="store_true", help="only run georeferencing; skip detection, tracking, and stabilization")
    g.add_argument("--plot-only", "-po", action="store_true", help="(reference flag) plots are not part of this build: nothing is run")
    g.add_argument("--no-geo", "-ng", action="store_true", help="do not georeference the tracking data")
    g.add_argument("--folders-exclude", "-fe", nargs="*", default=None, help="sub-folder names to skip (cfg -> batch -> folders_exclude)")
    g.add_argument("--exclude-patterns", "-ep", nargs="*", default=None, help="skip videos whose name contains any of these")
    return ap.parse_args(argv)


def main(argv=None) -> None:
    args = parse_cli_args(argv)
    logging.basicConfig(level=logging.DEBUG if getattr(args, "verbose", False) else logging.INFO, format="%(levelname)s %(message)s")
    process_input(args, logging.getLogger("geotrax_amd.batch"))


if __name__ == "__main__":
    main(sys.argv[1:])
