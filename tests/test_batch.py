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
                           folders_exclude=None, exclude_patterns=None, no_geo=True, geo_only=False, viz_only=False, plot_only=False)
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


def test_two_ranks_split_the_directory(tmp_path, monkeypatch):
    monkeypatch.setenv("GTX_BATCH_QUEUE", "0")        # the static deal (what a job without the shared counter falls back to)
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


def test_stage_chain_runs_each_stage_behind_its_own_skip_rule(tmp_path, caplog):
    """batch_process.py:288-307, 340-377: extract then georeference per file, a stage is skipped when ITS output exists,
    georeferencing needs the tracks file, --no-geo / --geo-only select stages, --overwrite --yes re-runs, --dry-run lists."""
    from geotrax_amd import batch

    for name in ("A1.npy", "B2.npy"):
        (tmp_path / name).write_bytes(b"x" * 10)
    log = []

    def run(a, lg):
        log.append(("extract", Path(a.source).name))
        out = Path(a.source).parent / "results"
        out.mkdir(exist_ok=True)
        (out / f"{Path(a.source).stem}.txt").write_text("0,1\n")

    def run_geo(a, lg):
        log.append(("georef", Path(a.source).name))
        if Path(a.source).name == "B2.npy" and not (tmp_path / "ortho_ok").exists():
            raise SystemExit(1)                      # no orthophoto for this location: georef_stage exits like the reference
        (Path(a.source).parent / "results" / f"{Path(a.source).stem}.csv").write_text("x\n")

    def go(**over):
        log.clear()
        stats = {}
        counts = batch.process_input(_args(tmp_path, **{'no_geo': False, **over}), logger, run=run, run_geo=run_geo, stats=stats)
        return counts, stats

    counts, stats = go()
    assert log == [("extract", "A1.npy"), ("georef", "A1.npy"), ("extract", "B2.npy"), ("georef", "B2.npy")]
    assert counts == dict(done=1, skipped=0, failed=1, dry=0)                # B2's georeference failed: the file counts as failed, the batch went on
    assert stats["extract"] == dict(done=2, skipped=0, failed=0, dry=0) and stats["georef"] == dict(done=1, skipped=0, failed=1, dry=0)
    (tmp_path / "ortho_ok").write_text("")
    counts, stats = go()
    assert log == [("georef", "B2.npy")]                                     # every other stage has its results
    assert counts == dict(done=1, skipped=1, failed=0, dry=0) and stats["extract"]["skipped"] == 2 and stats["georef"] == dict(done=1, skipped=1, failed=0, dry=0)
    counts, _ = go()
    assert log == [] and counts["skipped"] == 2
    (tmp_path / "results" / "A1.csv").unlink()
    counts, _ = go()
    assert log == [("georef", "A1.npy")]                                     # only the missing stage
    (tmp_path / "results" / "A1.txt").unlink()
    (tmp_path / "results" / "A1.csv").unlink()
    with caplog.at_level(logging.ERROR):
        counts, stats = go(geo_only=True)
    assert log == [] and "No detection, tracking, and stabilization results found. Skipping georeferencing." in caplog.text
    counts, _ = go(no_geo=True)
    assert log == [("extract", "A1.npy")]
    counts, stats = go(overwrite=True, yes=True)
    assert log == [("extract", "A1.npy"), ("georef", "A1.npy"), ("extract", "B2.npy"), ("georef", "B2.npy")] and counts["done"] == 2
    counts, stats = go(overwrite=True, yes=True, dry_run=True)
    assert log == [] and counts["dry"] == 2 and stats["extract"]["dry"] == 2 and stats["georef"]["dry"] == 2
    counts, _ = go(viz_only=True)
    assert log == [] and counts["skipped"] == 2
    # the CLI has the reference's stage flags and the georeferencing group
    a = batch.parse_cli_args([str(tmp_path), "-go", "-y", "-orf", str(tmp_path), "-nm"])
    assert a.geo_only and not a.no_geo and a.no_master is True and a.ortho_folder == tmp_path


def _queue_rank(rank, world, root, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    import time

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "geo-trax_amd"))
    from geotrax_amd import batch

    mine = []
    t0 = time.time()

    def run(a, log):
        time.sleep(Path(a.source).stat().st_size / 1000.0)       # a video takes as long as it is big
        mine.append((Path(a.source).name, time.time() - t0))

    counts = batch.process_input(_args(Path(root)), logging.getLogger(f"r{rank}"), run=run)
    q.put((rank, mine, counts))


def test_three_ranks_take_unequal_videos_from_the_shared_counter(tmp_path):
    """One long video and eight short ones over three ranks: with the static deal the rank that drew the long one also drew two
    short ones and finished 0.8 s after the others; from the shared counter it takes the long one and nothing else, and no rank
    sits idle for longer than one short video while work is left."""
    (tmp_path / "long.npy").write_bytes(b"x" * 900)
    for k in range(8):
        (tmp_path / f"short{k}.npy").write_bytes(b"x" * 100)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_queue_rank, args=(r, 3, str(tmp_path), port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    names = [[n for n, _ in g[1]] for g in got]
    assert sorted(n for ns in names for n in ns) == sorted(f.name for f in tmp_path.glob("*.npy"))     # every video once
    long_rank = [ns for ns in names if "long.npy" in ns]
    assert long_rank == [["long.npy"]]                                        # whoever took the long one took nothing else
    ends = [g[1][-1][1] for g in got]
    assert max(ends) - min(ends) < 0.75                                       # static deal: 1.1 s against 0.3 s
    assert all(g[2] == dict(done=9, skipped=0, failed=0, dry=0) for g in got)
