"""Batch orchestration (geotrax_amd.batch; reference geotrax/batch_process.py:220-377): discovery and
exclusion rules, skip / overwrite / ask semantics, failure isolation, the per-rank split. The extract
stage itself is replaced by a recorder here (the GPU path has its own tests)."""
import argparse
import logging
import multiprocessing as mp
import os
from pathlib import Path

import numpy as np
import pytest

logger = logging.getLogger("batch-test")


def _tree(root: Path):
    for rel, size in (("a/one.npy", 300), ("a/two.mp4", 100), ("b/three.mov", 200), ("b/skipme_four.mkv", 50), ("results/five.mp4", 10),
                      ("a/notes.txt", 5), ("c/d/six.avi", 400)):
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(b"x" * size)


def _args(root, **over):
    a = argparse.Namespace(input=root, cfg=None, output_folder=None, log_path=None, verbose=False, model=None, class_names=None, conf=None,
                           classes=None, cut_frame_left=None, cut_frame_right=None, interpolate=None, overwrite=False, yes=False, dry_run=False,
                           folders_exclude=None, exclude_patterns=None)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_discovery_exclusions_and_skip_rules(tmp_path):
    from geotrax_amd import batch

    _tree(tmp_path)
    files = batch.discover(tmp_path, ["results"], ["skipme"], logger)
    assert [f.name for f in files] == ["one.npy", "two.mp4", "three.mov", "six.avi"]          # sorted by path, filters applied
    seen = []

    def run(file_args, log):
        seen.append(Path(file_args.source).name)
        if file_args.source.name == "two.mp4":
            raise RuntimeError("decoder exploded")
        if file_args.source.name == "three.mov" and not (file_args.source.parent / "ok").exists():
            (file_args.source.parent / "ok").write_text("")
            raise SystemExit(1)                       # extract.py exits on an unreadable video / missing model: this file's failure only
        out = file_args.source.parent / "results"
        out.mkdir(exist_ok=True)
        (out / f"{file_args.source.stem}.txt").write_text("0,1\n")

    counts = batch.process_input(_args(tmp_path, exclude_patterns=["skipme"]), logger, run=run)
    assert seen == ["one.npy", "two.mp4", "three.mov", "six.avi"]
    assert counts == dict(done=2, skipped=0, failed=2, dry=0)                                  # an exception or a sys.exit in one file does not stop the batch
    seen.clear()
    counts = batch.process_input(_args(tmp_path, exclude_patterns=["skipme"]), logger, run=run)
    assert seen == ["two.mp4", "three.mov"] and counts["skipped"] == 2 and counts["done"] == 1
    seen.clear()
    counts = batch.process_input(_args(tmp_path, exclude_patterns=["skipme"]), logger, run=run)
    assert seen == ["two.mp4"] and counts["skipped"] == 3                                      # results exist, no --overwrite
    seen.clear()
    counts = batch.process_input(_args(tmp_path, exclude_patterns=["skipme"], overwrite=True, yes=True), logger, run=run)
    assert len(seen) == 4 and counts["done"] == 3
    seen.clear()
    counts = batch.process_input(_args(tmp_path, exclude_patterns=["skipme"], overwrite=True, yes=True, dry_run=True), logger, run=run)
    assert seen == [] and counts["dry"] == 4
    # --overwrite without --yes asks per file
    f = tmp_path / "a" / "one.npy"
    out_cfg = {"folder": "results"}
    assert batch.results_exist(f, out_cfg)
    a = _args(tmp_path, overwrite=True)
    assert batch.handle_existing_results(f, a, logger, True, "x", ask=lambda _: "Y") is True
    assert batch.handle_existing_results(f, a, logger, True, "x", ask=lambda _: "n") is False
    os.environ["WORLD_SIZE"] = "2"                    # under a launcher nobody prompts: --overwrite needs --yes
    try:
        assert batch.handle_existing_results(f, a, logger, True, "x", ask=lambda _: pytest.fail("prompted")) is False
    finally:
        del os.environ["WORLD_SIZE"]
    assert batch.process_input(_args(tmp_path / "nowhere"), logger, run=run) == dict(done=0, skipped=0, failed=0, dry=0)


def test_shard_deals_largest_first(tmp_path):
    from geotrax_amd import batch

    _tree(tmp_path)
    files = batch.discover(tmp_path, ["results"], None, logger)
    parts = [batch.shard(files, r, 2) for r in range(2)]
    assert sorted(f.name for p in parts for f in p) == sorted(f.name for f in files)
    assert [f.name for f in parts[0]] == ["six.avi", "three.mov", "skipme_four.mkv"]            # sizes 400, 200, 50
    assert [f.name for f in parts[1]] == ["one.npy", "two.mp4"]                                # 300, 100


def _rank_main(rank, world, root, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "geo-trax_amd"))
    from geotrax_amd import batch

    mine = []
    counts = batch.process_input(_args(Path(root)), logging.getLogger(f"r{rank}"), run=lambda a, log: mine.append(Path(a.source).name))
    q.put((rank, mine, counts, os.environ.get("GTX_DEVICE")))


def test_two_ranks_split_the_directory(tmp_path):
    _tree(tmp_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_rank_main, args=(r, 2, str(tmp_path), port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == ["six.avi", "three.mov", "skipme_four.mkv"] and got[1][1] == ["one.npy", "two.mp4"]
    assert got[0][2] == got[1][2] == dict(done=5, skipped=0, failed=0, dry=0)                  # all-reduced totals on every rank
    assert got[0][3] == "0" and got[1][3] == "1"                                               # one GPU per rank
