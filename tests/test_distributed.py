"""Frame sharding across ranks (geotrax_amd/distributed.py): world-size-2 gloo processes on CPU
must produce exactly what one process produces. Detector and stabilizer are stood in by
deterministic numpy functions of the frame (the sharding logic does not care what computes them);
the tracker is the real C++ one."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from geotrax_amd.distributed import extract_sharded, pack_frame_record, shard_range, unpack_frame_record

N_FRAMES, MAX_DET = 23, 40


def _frame(i):
    rng = np.random.default_rng(1000 + i)
    return rng.integers(0, 255, (8, 8, 3), dtype=np.uint8), i


def _fake_detect(frame):
    _, i = frame
    rng = np.random.default_rng(7)                      # same objects every frame, drifting
    c = rng.uniform(100, 900, (12, 2)) + 1.5 * i
    s = rng.uniform(20, 60, (12, 2))
    xyxy = np.c_[c - s / 2, c + s / 2].astype(np.float32)
    conf = np.sort(rng.uniform(0.3, 0.9, 12))[::-1].astype(np.float32)
    keep = np.ones(12, bool)
    keep[(i * 5) % 12] = i % 3 != 0                     # occasional miss
    return xyxy[keep], conf[keep], (np.arange(12) % 4).astype(np.int32)[keep]


class _FakeStab:
    def set_ref(self, frame, boxes):
        self.ref = frame[1]

    def stabilize(self, frame, boxes):
        i = frame[1]
        if i % 7 == 3:
            return None                                 # a frame without a transform (extract.py:185)
        return np.array([[1, 0, -0.1 * (i - self.ref)], [0, 1, 0.2 * (i - self.ref)], [0, 0, 1.0]])


def _xywh(b):
    return np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float32) if len(b) else None


def _producer(first, fail_at=None):
    """What geotrax_amd.extract does with the HIP engine, with the stand-ins: this rank's frames -> packed records."""
    stab = _FakeStab()
    stab.set_ref(_frame(first), None)

    def produce(runs):
        for start, stop in runs:
            for f in range(start, stop):
                if f == fail_at:
                    raise RuntimeError("decoder exploded")
                xyxy, conf, cls = _fake_detect(_frame(f))
                H = None if f == first else stab.stabilize(_frame(f), _xywh(xyxy))
                yield pack_frame_record(MAX_DET, xyxy, conf, cls, H)
    return produce


def _run(dist_mod, fail_at=None, run_frames=None):
    from geotrax_amd.geometry import warp_boxes
    from geotrax_amd.tracker import Tracker

    return extract_sharded(N_FRAMES, 2, _producer(2, fail_at), Tracker("bytetrack"), warp_boxes, MAX_DET, dist=dist_mod, run_frames=run_frames)


def _worker(rank, world, port, q, fail_at=None, run_frames=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if fail_at is not None:
            try:
                _run(dist, fail_at, run_frames)
            except RuntimeError as e:                   # EVERY rank learns of the failure through the collective and returns
                q.put((rank, str(e)))
            else:
                q.put((rank, "no error"))
            return
        out = _run(dist, run_frames=run_frames)
        if rank == 0:
            q.put([[np.asarray(a) for a in lst] for lst in out])
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def _spawn(world, fail_at=None, run_frames=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, fail_at, run_frames)) for r in range(world)]
    for p in procs:
        p.start()
    return q, procs


def test_a_failing_rank_voids_the_video_on_every_rank_without_a_hang():
    """SURVEY.md section 5 (failure detection): a rank whose shard raises still joins the gather; all ranks then raise, so
    rank 0 writes no partial output (extract.py:198-200) and nobody sits in a collective until the timeout."""
    q, procs = _spawn(2, fail_at=N_FRAMES - 3)          # a frame of rank 1's range
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][0] == 0 and "another rank" in got[0][1]
    assert got[1][0] == 1 and "decoder exploded" in got[1][1]
    with pytest.raises(RuntimeError):
        _run(None, fail_at=5)                            # single process: same rule


def test_world2_gloo_equals_single_process():
    single = _run(None)
    q, procs = _spawn(2)
    multi = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(single) == len(multi) == 7
    for a, b in zip(single, multi):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(np.asarray(x), y)
    frames = np.concatenate(single[0])[:, 0]
    assert frames.min() == 2 and frames.max() == N_FRAMES - 1
    assert len(single[6]) == N_FRAMES - 4                # frames 4.. all carry a transform: frame 3 (no model) has none yet,
    #                                                       later failures (10, 17) fall back to the last known one


@pytest.mark.parametrize("world,run_frames", [(2, 3), (8, 2), (8, None)])
def test_round_robin_runs_and_eight_ranks_equal_single_process(world, run_frames):
    """The overlapped form (runs of run_frames consecutive frames dealt round-robin, one gather per round, rank 0 replaying
    each round in a side thread while the ranks produce the next) and 8 ranks -- BASELINE configs[4]'s rank count, with a
    short last round and ranks that get no frames in it -- give exactly what one process gives, in clip order."""
    single = _run(None)
    q, procs = _spawn(world, run_frames=run_frames)
    multi = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(single) == len(multi) == 7
    for a, b in zip(single, multi):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(np.asarray(x), y)
    frames = np.concatenate(multi[0])[:, 0]
    assert (np.diff(frames.astype(int)) >= 0).all()          # clip order


def test_a_rank_failing_in_a_later_round_stops_every_rank_in_that_round():
    q, procs = _spawn(2, fail_at=N_FRAMES - 3, run_frames=4)
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted("decoder exploded" in g[1] for g in got) == [False, True] and all("failed" in g[1] for g in got)


def test_shard_runs_partition_the_frames_in_clip_order():
    from geotrax_amd.distributed import shard_runs

    for n, first, world, rf in [(150, 0, 8, 16), (23, 2, 2, 4), (23, 2, 8, 2), (5, 0, 8, 16), (0, 0, 3, 4), (150, 0, 8, None)]:
        rounds = shard_runs(n, world, first, rf)
        seen = [f for row in rounds for (s, e) in row for f in range(s, e)]
        assert seen == list(range(first, n))                  # round by round, rank by rank = clip order
        assert all(len(row) == world for row in rounds)
        if rf is not None:
            assert all(e - s <= rf for row in rounds for (s, e) in row)


def test_shard_ranges_partition_the_frames():
    for n, first, world in [(150, 0, 8), (7, 0, 8), (23, 2, 2), (1, 0, 4), (0, 0, 3), (100, 99, 5)]:
        seen = []
        for r in range(world):
            s, e = shard_range(n, r, world, first)
            assert s <= e
            seen += list(range(s, e))
        assert seen == list(range(first, n))
        sizes = [shard_range(n, r, world, first)[1] - shard_range(n, r, world, first)[0] for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_frame_record_round_trip():
    rng = np.random.default_rng(0)
    xyxy, conf, cls = rng.uniform(0, 3840, (5, 4)).astype(np.float32), rng.uniform(0, 1, 5).astype(np.float32), np.arange(5, dtype=np.int32)
    H = rng.normal(size=(3, 3))
    rec = pack_frame_record(8, xyxy, conf, cls, H)
    a, b, c, d = unpack_frame_record(rec, 8)
    np.testing.assert_array_equal(a, xyxy)
    np.testing.assert_array_equal(b, conf)
    np.testing.assert_array_equal(c, cls)
    np.testing.assert_array_equal(d, H)
    a, b, c, d = unpack_frame_record(pack_frame_record(8, xyxy[:0], conf[:0], cls[:0], None), 8)
    assert len(a) == 0 and d is None


def test_frame_record_with_gmc_block_round_trips():
    """Records of the frame-sharded BoT-SORT run carry the rank's camera-motion warp next to the detections and H."""
    from geotrax_amd.distributed import pack_frame_record, unpack_frame_gmc, unpack_frame_record

    rng = np.random.default_rng(0)
    xyxy = rng.uniform(0, 3000, (5, 4)).astype(np.float32)
    conf, cls = rng.uniform(0.3, 1, 5).astype(np.float32), rng.integers(0, 4, 5).astype(np.int32)
    Hm, warp = rng.normal(size=(3, 3)), rng.normal(size=(2, 3))
    plain = pack_frame_record(8, xyxy, conf, cls, Hm)
    with_gmc = pack_frame_record(8, xyxy, conf, cls, Hm, warp, with_gmc=True)
    assert len(with_gmc) == len(plain) + 7
    for rec in (plain, with_gmc):
        b, c, k, H2 = unpack_frame_record(rec, 8)
        np.testing.assert_array_equal(b, xyxy)
        np.testing.assert_array_equal(c, conf)
        np.testing.assert_array_equal(k, cls)
        np.testing.assert_array_equal(H2, Hm)
    np.testing.assert_array_equal(unpack_frame_gmc(with_gmc), warp)
    none = pack_frame_record(8, xyxy[:0], conf[:0], cls[:0], None, None, with_gmc=True)
    assert unpack_frame_gmc(none) is None and unpack_frame_record(none, 8)[3] is None and len(unpack_frame_record(none, 8)[1]) == 0


def test_wire_format_is_the_record_at_its_real_length():
    """a frame's record travels at the length of what it holds (SURVEY 8e budgets 3.3 KB at 132 boxes), not
    as the 48 KB fixed-stride row; unpacking gives the row back bit for bit -- boxes / scores / classes are float32 at their
    source, H and the camera-motion warp stay float64."""
    from geotrax_amd.distributed import pack_frame_record, wire_pack, wire_unpack

    rng = np.random.default_rng(0)
    for with_gmc in (False, True):
        recs = []
        for n in (0, 3, 132, 1000, 0):
            xyxy = rng.uniform(0, 4000, (n, 4)).astype(np.float32)
            conf, cls = rng.uniform(0, 1, n).astype(np.float32), rng.integers(0, 4, n)
            Hm = rng.standard_normal((3, 3)) if n != 3 else None
            warp = rng.standard_normal((2, 3)) if (with_gmc and n) else None
            recs.append(pack_frame_record(1000, xyxy, conf, cls, Hm, warp, with_gmc=with_gmc))
        block = np.stack(recs)
        wire = wire_pack(block, 1000, with_gmc)
        assert wire.dtype == np.uint8
        np.testing.assert_array_equal(wire_unpack(wire, len(recs), 1000, with_gmc), block)
        one = wire_pack(block[2:3], 1000, with_gmc)
        assert len(one) == 8 + 132 * 24 + 72 + (48 if with_gmc else 0) <= 3400        # 3.3 KB at 132 boxes
        with pytest.raises(ValueError):
            wire_unpack(wire[:-8], len(recs), 1000, with_gmc)


def _exchange_worker(rank, world, port, q, fail):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geotrax_amd.distributed import agree_on_source, broadcast_weights, gather_records, pack_frame_record

        rng = np.random.default_rng(7)
        tensors = {"model.0.conv.weight": rng.standard_normal((32, 3, 3, 3)).astype(np.float32), "model.0.conv.bias": rng.standard_normal(32).astype(np.float32),
                   "model.22.cv3.0.2.weight": rng.standard_normal((4, 64, 1, 1)).astype(np.float32), "scalar": np.float32(3.5).reshape(())}
        names = {0: "car", 1: "bus", 2: "truck", 3: "motorcycle"}
        if fail:
            try:
                broadcast_weights(None, None, dist, error="no such file" if rank == 0 else None)
            except RuntimeError as e:
                q.put((rank, str(e)))
            return
        got, got_names = broadcast_weights(tensors if rank == 0 else None, names if rank == 0 else None, dist)
        same = all(np.array_equal(got[k], tensors[k]) and got[k].shape == tensors[k].shape and got[k].dtype == np.float32 for k in tensors)
        ok_all = agree_on_source(True, 150, dist)
        ok_one = agree_on_source(rank != 1, 150, dist)                 # rank 1 cannot open its copy
        ok_cnt = agree_on_source(True, 150 - rank, dist)               # the copies differ in length
        # a round's gather: ranks hold different numbers of boxes, so their packed sizes differ
        block = np.stack([pack_frame_record(50, rng.uniform(0, 99, (5 + 7 * rank, 4)).astype(np.float32), np.full(5 + 7 * rank, 0.5, np.float32),
                                            np.zeros(5 + 7 * rank, np.int32), np.eye(3) * (rank + 1)) for _ in range(3)])
        blocks, failed = gather_records(block, False, dist, None, 50, False)
        sizes = [int(b[0, 0]) for b in blocks] if rank == 0 else None
        q.put((rank, same and set(got) == set(tensors), got_names == names, ok_all, ok_one, ok_cnt, failed, sizes,
               bool(rank != 0 or all(np.array_equal(blocks[r][:, -9:], np.tile((np.eye(3) * (r + 1)).ravel(), (3, 1))) for r in range(world)))))
    finally:
        dist.destroy_process_group()


def _spawn_exchange(world, fail):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q, fail)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_weight_broadcast_source_agreement_and_packed_gather_over_gloo():
    """The product's exchanges of a frame-sharded run with 3 ranks (gloo here, RCCL on GPUs): rank 0's tensors arrive on every rank
    bit for bit (north star: "broadcast of weights"), with the class names; the ranks agree on whether the source opened
    everywhere and on its frame count before the first round; a round's records are gathered at their packed length."""
    got = _spawn_exchange(3, fail=False)
    for rank, same, names_ok, ok_all, ok_one, ok_cnt, failed, sizes, h_ok in got:
        assert same and names_ok and not failed and h_ok
        assert ok_all == (True, True, 150) and ok_one[0] is False and ok_cnt[:2] == (True, False)
        assert sizes == ([5, 12, 19] if rank == 0 else None)


def test_a_weight_file_rank0_cannot_load_stops_every_rank():
    got = _spawn_exchange(2, fail=True)
    assert [g[0] for g in got] == [0, 1] and all("no such file" in g[1] for g in got)


def test_replay_core_is_reserved_only_for_four_ranks_and_more(monkeypatch):
    from geotrax_amd.distributed import pin_to_core, release_replay_core, reserve_replay_core

    before = os.sched_getaffinity(0)
    try:
        monkeypatch.delenv("GTX_PIN_REPLAY", raising=False)
        assert reserve_replay_core(2) is None and os.sched_getaffinity(0) == before
        if len(before) >= 4:
            core = reserve_replay_core(8)
            assert core == max(before) and os.sched_getaffinity(0) == before - {core}       # the caller (and its later threads) stay off it
            # once per process: a folder of videos calls it per video; the mask must not shrink by a core each time
            for _ in range(5):
                assert reserve_replay_core(8) == core and os.sched_getaffinity(0) == before - {core}
            import threading

            seen = []
            t = threading.Thread(target=lambda: seen.append((pin_to_core(core), os.sched_getaffinity(0))))
            t.start()
            t.join()
            assert seen == [(True, {core})]
            release_replay_core()
            assert os.sched_getaffinity(0) == before
        monkeypatch.setenv("GTX_PIN_REPLAY", "0")
        os.sched_setaffinity(0, before)
        assert reserve_replay_core(8) is None
        assert pin_to_core(None) is False
    finally:
        release_replay_core()
        os.sched_setaffinity(0, before)
