"""Host-side sanitizer builds (the build owns race / memory checking for its host C++; GPU
AddressSanitizer is not available on the pool). `make asan` / `make tsan` compile the tracker and the geometry helpers
with g++ -fsanitize=address,undefined / thread and drive them with long seeded streams (dense overlaps, empty frames,
track turnover, six trackers on six threads). Any report aborts the driver with a non-zero status."""
import shutil
import subprocess
from pathlib import Path

import pytest

PKG = Path(__file__).resolve().parent.parent / "geo-trax_amd"


@pytest.mark.parametrize("target", ["asan", "tsan"])
def test_host_code_is_clean_under_sanitizers(target):
    if shutil.which("g++") is None or not Path("/opt/rocm/include/hip/hip_runtime.h").exists():
        pytest.skip("needs g++ and the HIP headers")
    p = subprocess.run(["make", "-C", str(PKG), target], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "sanitize_host ok" in p.stdout and "ERROR" not in p.stderr and "WARNING: ThreadSanitizer" not in p.stderr
