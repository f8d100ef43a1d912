"""The C-ABI library loads without a GPU and exports every symbol include/gtx.h declares (and
nothing in the ctypes table is missing from the header); host-only entry points behave."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _header_symbols():
    text = (ROOT / "include" / "gtx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(gtx_[a-z0-9_]+)\s*\(", text))


def test_library_exports_every_declared_symbol():
    from geotrax_amd import _lib

    lib = _lib.load()
    declared = _header_symbols()
    assert len(declared) > 40
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/gtx.h but not exported by libgtx.so"
    assert declared == set(_lib._SIGNATURES), declared ^ set(_lib._SIGNATURES)
    assert lib.gtx_abi_version() == _lib.ABI_VERSION == 10


def test_errors_are_codes_with_messages_not_exceptions():
    from geotrax_amd import _lib

    lib = _lib.load()
    assert lib.gtx_last_error() is not None
    out = np.zeros(4, np.float32)
    rc = lib.gtx_warp_boxes(None, None, 1, _lib.ptr(out))          # NULL matrix
    assert rc == -1 and b"NULL" in lib.gtx_last_error()
    h = C.c_void_p()
    cfg = _lib.TrackerConfig(type=7)
    rc = lib.gtx_tracker_create(C.byref(cfg), C.byref(h))
    assert rc < 0 and b"tracker type" in lib.gtx_last_error()
    with pytest.raises(_lib.GtxError):
        _lib.check(rc)


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a GPU the compute entry points must fail with a message; nothing falls back to CPU."""
    from geotrax_amd import _lib

    lib = _lib.load()
    if lib.gtx_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.GtxError):
        _lib.Context(0)


def test_brief_table_equals_the_oracles_own():
    """The steered-BRIEF sampling table is generated independently by the oracle (published recipe: Gaussian pairs,
    sigma = patch/5, 256 orientation bins) and by the library; they must be the same bytes. Host only."""
    import sys

    sys.path.insert(0, str(ROOT))
    from geotrax_amd import _lib
    from oracle.stabilo_ref import brief_pattern

    lib = _lib.load()
    out = np.zeros((256, 256, 4), np.int8)
    _lib.check(lib.gtx_stabilizer_pattern(None, _lib.ptr(out)))
    want = brief_pattern()
    np.testing.assert_array_equal(out, want)
    assert np.abs(want.astype(int)).max() <= 17 and len(np.unique(want[0].reshape(256, 4), axis=0)) > 250


def test_graft_entry_build_runs():
    """The driver's build check: make (a no-op when the library is up to date), load, ABI version, package import."""
    import importlib
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    g = importlib.import_module("__graft_entry__")
    g.build()
