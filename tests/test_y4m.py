"""The uncompressed video source: YUV4MPEG2 reader, its host colour conversion against
oracle/yuv_ref.py (the published cvtColor(COLOR_YUV2BGR_I420) arithmetic), seeking, and the file written by write_y4m
coming back within the quantisation of 8-bit 4:2:0. CPU only; tests/test_y4m_gpu.py checks the HIP kernel and the engine."""
import numpy as np
import pytest


def _clip(n=4, h=37, w=50, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = 120 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0)
    return [np.clip(np.stack([base + 30 * np.sin(k + c) + 5 * rng.standard_normal((h, w)) for c in range(3)], -1), 0, 255).astype(np.uint8) for k in range(n)]


def test_y4m_round_trip_reader_and_seek(tmp_path):
    from geotrax_amd.frames import Y4mReader, open_source, write_y4m, yuv420_to_bgr_host
    from oracle.yuv_ref import i420_to_bgr

    frames = _clip()
    p = tmp_path / "clip.y4m"
    write_y4m(p, frames, fps=(30000, 1001))
    r = open_source(p)
    assert isinstance(r, Y4mReader) and r.frame_count == 4 and r.frame_hw == (37, 50) and abs(r.fps - 29.97) < 0.01 and r.isOpened()
    got = []
    for k in range(4):
        ok, f = r.read()
        assert ok and f.shape == (37, 50, 3) and f.nbytes == 37 * 50 * 3
        bgr = f.bgr()
        np.testing.assert_array_equal(bgr, i420_to_bgr(f.data, 37, 50))          # host conversion == the oracle, odd sizes included
        np.testing.assert_array_equal(bgr, yuv420_to_bgr_host(f.data, 37, 50))
        # smooth content survives 8-bit limited-range 4:2:0 within a few grey levels
        assert np.abs(bgr.astype(int) - frames[k].astype(int)).mean() < 4.0
        got.append(bgr)
    assert r.read() == (False, None)
    r.seek(2)
    ok, f = r.read()
    np.testing.assert_array_equal(f.bgr(), got[2])
    r.release()
    assert not r.isOpened()
    (tmp_path / "bad.y4m").write_bytes(b"RIFF....")
    with pytest.raises(ValueError):
        open_source(tmp_path / "bad.y4m")
    (tmp_path / "deep.y4m").write_bytes(b"YUV4MPEG2 W8 H8 F30:1 C420p10\nFRAME\n" + bytes(8 * 8 * 3))
    with pytest.raises(NotImplementedError):
        open_source(tmp_path / "deep.y4m")


def test_conversion_known_answers():
    """Limited-range BT.601: (Y,U,V) = (16,128,128) is black, (235,128,128) white, (81,90,240) pure red within a level."""
    from oracle.yuv_ref import i420_to_bgr

    def px(y, u, v):
        return i420_to_bgr(np.array([y] * 4 + [u] + [v], np.uint8), 2, 2)[0, 0]

    np.testing.assert_array_equal(px(16, 128, 128), [0, 0, 0])
    np.testing.assert_array_equal(px(235, 128, 128), [255, 255, 255])
    assert np.abs(px(81, 90, 240).astype(int) - [0, 0, 255]).max() <= 1
    np.testing.assert_array_equal(px(0, 128, 128), [0, 0, 0])                         # below black clamps


def test_frame_headers_with_parameters_and_damaged_files(tmp_path, caplog):
    """YUV4MPEG2 allows parameters on FRAME lines: the reader indexes the headers instead of assuming a fixed stride, so every
    frame stays reachable by number (frame-sharded ranks seek) and the count is the same however it is derived; a file that
    ends inside a frame or has garbage where a header should be is played up to its last complete frame with a warning (the
    reference's cv2 loop reads until the first failed read, extract.py:146-148) -- the same count on every rank --, and only a
    file without one readable frame is refused (with its handle closed)."""
    from geotrax_amd.frames import Y4mReader, write_y4m

    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (18, 26, 3), dtype=np.uint8) for _ in range(5)]
    plain = tmp_path / "plain.y4m"
    write_y4m(plain, frames)
    raw = plain.read_bytes()
    head_end = raw.index(b"\n") + 1
    fb = 18 * 26 + 2 * 9 * 13
    parts = [raw[:head_end]]
    for k in range(5):
        body = raw[head_end + k * (6 + fb) + 6:head_end + (k + 1) * (6 + fb)]
        parts += [b"FRAME Ip\n" if k in (1, 3) else b"FRAME\n", body]        # two frames carry a parameter
    odd = tmp_path / "odd.y4m"
    odd.write_bytes(b"".join(parts))
    a, b = Y4mReader(plain), Y4mReader(odd)
    assert a.frame_count == b.frame_count == 5
    for k in (4, 1, 3, 0, 2):                                                 # any order
        a.seek(k)
        b.seek(k)
        (oka, fa), (okb, fb_) = a.read(), b.read()
        assert oka and okb
        np.testing.assert_array_equal(fa.bgr(), fb_.bgr())
    a.release()
    b.release()
    (tmp_path / "short.y4m").write_bytes(raw[:-7])
    with caplog.at_level("WARNING"):
        s = Y4mReader(tmp_path / "short.y4m")
    assert s.frame_count == 4 and "cut short" in s.truncated and "cut short" in caplog.text
    s.seek(3)
    ok, f3 = s.read()
    assert ok and s.read() == (False, None)
    s.release()
    (tmp_path / "bad.y4m").write_bytes(raw[:head_end + 6 + fb] + b"JUNK!\n" + raw[head_end + 6 + fb + 6:])
    g = Y4mReader(tmp_path / "bad.y4m")
    assert g.frame_count == 1 and "FRAME header" in g.truncated
    g.release()
    (tmp_path / "none.y4m").write_bytes(raw[:head_end + 6 + fb - 3])              # not even one complete frame
    with pytest.raises(ValueError, match="cut short"):
        Y4mReader(tmp_path / "none.y4m")
    (tmp_path / "junk.y4m").write_bytes(raw[:head_end] + b"JUNK!\n" + raw[head_end + 6:])
    with pytest.raises(ValueError, match="FRAME header"):
        Y4mReader(tmp_path / "junk.y4m")
    # the layout the read-ahead feeder reads from: payload offsets of the complete frames
    path, kind, off = Y4mReader(odd).raw_layout()
    assert kind == "i420" and len(off) == 5 and path == odd
    for k in range(5):
        assert odd.read_bytes()[off[k]:off[k] + fb] == raw[head_end + k * (6 + fb) + 6:head_end + (k + 1) * (6 + fb)]


def test_compressed_sources_need_a_decoder_and_say_so(tmp_path):
    """.mp4 and friends go through cv2.VideoCapture when OpenCV is installed (frames.Cv2Reader, the reference's own reader,
    extract.py:248); this image has no decoder, so the source is refused with the way out spelled out. Where cv2 exists the
    reader is exercised instead."""
    from geotrax_amd.frames import open_source

    clip = tmp_path / "clip.mp4"
    clip.write_bytes(b"\x00" * 64)
    try:
        import cv2  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="video decoder"):
            open_source(clip)
        return
    r = open_source(clip)                                   # not a video: VideoCapture reports it closed, read() fails
    assert not r.isOpened() or r.read()[0] is False
    r.release()
