"""The .y4m route on the GPU: gtx_yuv420_to_bgr_dev against oracle/yuv_ref.py at 4K and at
ragged sizes, and the extract CLI on a .y4m clip producing exactly what it produces on the same frames given as BGR."""
import ctypes as C
import logging

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
logger = logging.getLogger("y4m-gpu")


@pytest.mark.parametrize("hw", [(2160, 3840), (37, 50), (64, 130), (9, 6)])
def test_yuv_kernel_matches_oracle(gtx_ctx, hw):
    from geotrax_amd import _lib
    from oracle.yuv_ref import i420_to_bgr

    h, w = hw
    rng = np.random.default_rng(h * w)
    n = h * w + 2 * ((h + 1) // 2) * ((w + 1) // 2)
    data = rng.integers(0, 256, n, dtype=np.uint8)
    src, dst = gtx_ctx.dev_alloc(n), gtx_ctx.dev_alloc(h * w * 3)
    try:
        gtx_ctx.dev_upload(src, data)
        _lib.check(gtx_ctx.lib.gtx_yuv420_to_bgr_dev(gtx_ctx.handle, C.c_void_p(src), h, w, C.c_void_p(dst)))
        out = np.empty((h, w, 3), np.uint8)
        gtx_ctx.dev_download(out, dst)
    finally:
        gtx_ctx.dev_free(src)
        gtx_ctx.dev_free(dst)
    np.testing.assert_array_equal(out, i420_to_bgr(data, h, w))


def test_extract_on_y4m_equals_extract_on_the_converted_frames(gtx_ctx, tmp_path):
    from test_extract_gpu import H, W, _cfg_file, _weights_file

    from geotrax_amd import extract as ex
    from geotrax_amd.frames import open_source, write_y4m
    from geotrax_amd.synth import make_scene

    sc = make_scene(seed=4, h=H, w=W)
    y4m = tmp_path / "U_clip.y4m"
    write_y4m(y4m, [sc.render(3 * t, 150) for t in range(6)])
    r = open_source(y4m)
    frames = []
    while True:
        ok, f = r.read()
        if not ok:
            break
        frames.append(f.bgr())
    r.release()
    npy = tmp_path / "npy" / "U_clip.npy"
    npy.parent.mkdir()
    np.save(npy, np.stack(frames))
    wpath, _ = _weights_file(tmp_path, gtx_ctx, frames[0])
    cfg_path, _ = _cfg_file(tmp_path, wpath, tracker="botsort")
    ex.main([str(y4m), "--cfg", str(cfg_path), "--output-folder", str(tmp_path / "a")])
    ex.main([str(npy), "--cfg", str(cfg_path), "--output-folder", str(tmp_path / "b")])
    a, b = (tmp_path / "a" / "U_clip.txt").read_text(), (tmp_path / "b" / "U_clip.txt").read_text()
    assert a == b and len(a.splitlines()) > 30                       # GPU conversion == host conversion, byte for byte downstream
    assert (tmp_path / "a" / "U_clip_vid_transf.txt").read_text() == (tmp_path / "b" / "U_clip_vid_transf.txt").read_text()
