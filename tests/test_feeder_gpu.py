"""The read-ahead feeder (csrc/feeder.cpp, geotrax_amd/feeder.py): `cap.read()` of the reference's loop (geotrax/extract.py:146)
off the detector stage thread. The bytes that reach HBM are the file's; the extract outputs are byte-identical to the
synchronous reader's, for every container, cut range and batch remainder; a read error voids the video like any exception
in the reference's loop (extract.py:198-200)."""
import logging
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
logger = logging.getLogger("test_feeder")


def _download(ctx, ptr, n, h, w):
    out = np.empty((n, h, w, 3), np.uint8)
    ctx.dev_download(out, ptr)
    return out


@pytest.mark.parametrize("kind", ["bgr", "i420"])
def test_file_mode_delivers_the_files_frames_in_order_through_a_small_ring(gtx_ctx, tmp_path, kind):
    """23 frames, batches of 3 (a remainder of 2), a ring of 4 batches and a consumer that holds two batches before it
    releases: every slot is reused several times; what arrives in HBM is the file's frames (I420: the oracle's conversion)."""
    from oracle.yuv_ref import i420_to_bgr

    from geotrax_amd.feeder import FrameFeeder
    from geotrax_amd.frames import open_source, write_y4m

    h, w, n = 70, 94, 23
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    if kind == "bgr":
        path = tmp_path / "clip.npy"
        np.save(path, frames)
        want = frames
    else:
        path = tmp_path / "clip.y4m"
        write_y4m(path, list(frames))
        r = open_source(path)
        want = np.stack([i420_to_bgr(r.read()[1].data, h, w) for _ in range(n)])
        r.release()
    reader = open_source(path)
    p, k, off = reader.raw_layout()
    assert k == kind and len(off) == n
    fd = FrameFeeder((h, w), kind=kind, batch=3, ring=4, device=gtx_ctx.device)
    fd.open_file(p, off[2:], n_threads=3)                       # frames 2..22: 7 batches of 3
    got = []
    for b in fd.batches(in_flight=2):
        b.wait_on(gtx_ctx)
        gtx_ctx.synchronize()
        got.append(_download(gtx_ctx, b.ptr, b.n, h, w))
    assert [len(g) for g in got] == [3] * 7
    np.testing.assert_array_equal(np.concatenate(got), want[2:])
    fd.close()
    reader.release()


def test_ring_reuse_under_an_uneven_consumer(gtx_ctx, tmp_path):
    """400 small frames through a ring of 3 batches with a consumer that sometimes dawdles and sometimes races: every slot
    is rewritten ~65 times while batches are still being read; every byte that arrives is the file's."""
    import time

    from geotrax_amd.feeder import FrameFeeder

    h, w, n = 48, 64, 400
    rng = np.random.default_rng(7)
    frames = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    off = 128 + h * w * 3 * np.arange(n, dtype=np.int64)
    assert np.load(path, mmap_mode="r").offset == 128
    fd = FrameFeeder((h, w), kind="bgr", batch=2, ring=3, device=gtx_ctx.device)
    fd.open_file(path, off, n_threads=4)
    k = 0
    for b in fd.batches(in_flight=2):
        if rng.random() < 0.15:
            time.sleep(float(rng.uniform(0, 0.004)))
        b.wait_on(gtx_ctx)
        got = _download(gtx_ctx, b.ptr, b.n, h, w)                 # dev_download runs on gtx_ctx's stream, behind the wait
        np.testing.assert_array_equal(got, frames[k:k + b.n], err_msg=f"batch at frame {k}")
        k += b.n
    assert k == n
    fd.close()


def test_push_mode_and_a_failing_source(gtx_ctx):
    from geotrax_amd.feeder import FrameFeeder

    h, w = 48, 64
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (11, h, w, 3), dtype=np.uint8)
    fd = FrameFeeder((h, w), kind="bgr", batch=2, ring=3, device=gtx_ctx.device)
    fd.open_reader(iter(frames))
    got = []
    for b in fd.batches(in_flight=1):
        b.wait_on(None)                                          # the calling thread waits
        got.append(_download(gtx_ctx, b.ptr, b.n, h, w))
    assert [len(g) for g in got] == [2, 2, 2, 2, 2, 1]
    np.testing.assert_array_equal(np.concatenate(got), frames)
    fd.close()

    # frames held in host memory, copied by the library's own threads (gtx_feeder_open_memory); the same array may repeat
    fd = FrameFeeder((h, w), kind="bgr", batch=2, ring=3, device=gtx_ctx.device)
    order = [0, 1, 2, 1, 0, 5, 5, 10, 3]
    fd.open_memory([frames[k] for k in order], n_threads=3)
    got = []
    for b in fd.batches(in_flight=1):
        b.wait_on(gtx_ctx)
        gtx_ctx.synchronize()
        got.append(_download(gtx_ctx, b.ptr, b.n, h, w))
    np.testing.assert_array_equal(np.concatenate(got), frames[order])
    fd.close()
    # several producer threads, frames pushed by number (gtx_feeder_push_at): still delivered in clip order
    fd = FrameFeeder((h, w), kind="bgr", batch=2, ring=3, device=gtx_ctx.device)
    fd.open_indexed(lambda i: frames[i], len(frames), threads=3)
    got = []
    for b in fd.batches(in_flight=1):
        b.wait_on(gtx_ctx)
        gtx_ctx.synchronize()
        got.append(_download(gtx_ctx, b.ptr, b.n, h, w))
    np.testing.assert_array_equal(np.concatenate(got), frames)
    fd.close()

    def bad_at(i):
        if i == 5:
            raise OSError("frame 5 is gone")
        return frames[i]

    fd = FrameFeeder((h, w), kind="bgr", batch=2, ring=3, device=gtx_ctx.device)
    fd.open_indexed(bad_at, len(frames), threads=2)
    with pytest.raises(OSError, match="frame 5 is gone"):
        for b in fd.batches(in_flight=1):
            pass
    fd.close()

    def broken():
        yield frames[0]
        yield frames[1]
        yield frames[2]
        raise OSError("decoder gave up")

    fd = FrameFeeder((h, w), kind="bgr", batch=2, ring=3, device=gtx_ctx.device)
    fd.open_reader(broken())
    seen = 0
    with pytest.raises(OSError, match="decoder gave up"):
        for b in fd.batches(in_flight=1):
            seen += b.n
    assert seen == 3                                             # the frames before the failure were delivered
    fd.close()
    # a consumer that walks away while the producer is blocked on a full ring
    fd = FrameFeeder((h, w), kind="bgr", batch=1, ring=2, device=gtx_ctx.device)
    fd.open_reader(iter(frames))
    assert fd.next().n == 1
    fd.close()


def test_a_file_that_shrinks_under_the_feeder_is_an_error_after_the_frames_before_it(gtx_ctx, tmp_path):
    from geotrax_amd import _lib
    from geotrax_amd.feeder import FrameFeeder

    h, w = 32, 40
    frames = np.arange(6 * h * w * 3, dtype=np.uint32).astype(np.uint8).reshape(6, h, w, 3)
    path = tmp_path / "clip.npy"
    np.save(path, frames)
    step = h * w * 3
    off = 128 + step * np.arange(8, dtype=np.int64)             # two frames beyond the end of the file
    fd = FrameFeeder((h, w), kind="bgr", batch=2, ring=6, device=gtx_ctx.device)
    fd.open_file(path, off, n_threads=2)
    n = 0
    with pytest.raises(_lib.GtxError, match="could not be read"):
        for b in fd.batches(in_flight=1):
            n += b.n
    assert n <= 6
    fd.close()


@pytest.mark.parametrize("container", ["y4m", "npy", "dir"])
def test_extract_with_the_feeder_equals_the_synchronous_reader(gtx_ctx, tmp_path, monkeypatch, container):
    """The product loop on a file: read-ahead feeder (default) vs GTX_FEEDER=0, 7 frames, cut to frames 1..5 (five frames:
    batches of 2 + a remainder), BoT-SORT (GMC on): the two result files are the same bytes."""
    from test_extract_gpu import H, W, _cfg_file, _weights_file

    from geotrax_amd import extract as ex
    from geotrax_amd.frames import write_y4m
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=5, h=H, w=W)
    frames = [sc.render(4 * t, 150) for t in range(7)]
    if container == "y4m":
        src = tmp_path / "U_clip.y4m"
        write_y4m(src, frames)
    elif container == "npy":
        src = tmp_path / "U_clip.npy"
        np.save(src, np.stack(frames))
    else:
        src = tmp_path / "U_clip"
        src.mkdir()
        for k, f in enumerate(frames):
            np.save(src / f"{k:04d}.npy", f)
    wpath, _ = _weights_file(tmp_path, gtx_ctx, frames[0])
    cfg_path, _ = _cfg_file(tmp_path, wpath, tracker="botsort")
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GTX_FEEDER", mode)
        ex.main([str(src), "--cfg", str(cfg_path), "--output-folder", str(tmp_path / f"out{mode}"), "--cut-frame-left", "1", "--cut-frame-right", "5"])
        outs[mode] = ((tmp_path / f"out{mode}" / "U_clip.txt").read_text(), (tmp_path / f"out{mode}" / "U_clip_vid_transf.txt").read_text())
    assert outs["1"] == outs["0"]
    rows = np.loadtxt(tmp_path / "out1" / "U_clip.txt", delimiter=",")
    assert set(np.unique(rows[:, 0]).astype(int)) <= {1, 2, 3, 4, 5} and len(rows) > 20
    tr = np.loadtxt(tmp_path / "out1" / "U_clip_vid_transf.txt", delimiter=",")
    np.testing.assert_array_equal(tr[:, 0], [2, 3, 4, 5])       # frame 1 is the reference frame of the cut clip


def test_a_read_error_in_the_feeder_voids_the_video(gtx_ctx, tmp_path, caplog):
    """extract.py:198-200: any exception inside the loop -> error line, empty tables. Here: the .y4m loses its tail between
    the index scan and the reads."""
    import argparse

    from test_extract_gpu import H, W, _cfg_file, _weights_file

    from geotrax_amd import extract as ex
    from geotrax_amd import frames as fr
    from geotrax_amd.config_utils import load_config_all
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=6, h=H, w=W)
    frames = [sc.render(4 * t, 150) for t in range(6)]
    src = tmp_path / "U_clip.y4m"
    fr.write_y4m(src, frames)
    wpath, _ = _weights_file(tmp_path, gtx_ctx, frames[0])
    cfg_path, _ = _cfg_file(tmp_path, wpath)
    args = argparse.Namespace(source=str(src), cfg=cfg_path, output_folder=None, log_path=None, verbose=False, model=None,
                              class_names=None, conf=None, classes=None, cut_frame_left=None, cut_frame_right=None, interpolate=None)
    model = ex.load_detector(args, logger)
    config = load_config_all(args, logger, model_names=model.names)
    real = fr.Y4mReader.raw_layout

    def shrunk(self):
        p, k, off = real(self)
        os.truncate(p, int(off[3]) + 100)                         # frame 3 onwards is gone
        return p, k, off

    fr.Y4mReader.raw_layout = shrunk
    try:
        with caplog.at_level(logging.ERROR):
            tracks, transforms = ex.track_with_model(model, config, logger)
    finally:
        fr.Y4mReader.raw_layout = real
    assert tracks.shape == (0, 12) and transforms.shape == (0, 10)
    assert "Error processing" in caplog.text and "could not be read" in caplog.text
