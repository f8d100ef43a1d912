"""pytest configuration: `-m gpu` tests need an MI355X and the built libgtx.so; everything else
runs on CPU. The package directory has a hyphen, so it is put on sys.path here."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "geo-trax_amd", ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gtx_ctx():
    from geotrax_amd import _lib

    return _lib.default_context(0)
