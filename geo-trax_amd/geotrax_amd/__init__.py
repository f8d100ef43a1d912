"""geotrax_amd -- MI355X-native implementation of the geo-trax per-frame extraction hot path.

Mirrors the reference interface for that path only (geotrax/extract.py:134-214): a detector
object with ``track()``, a ``Stabilizer`` with the stabilo method names, and the numpy
post-processing / writers that define the output schema. All per-frame arithmetic runs in
libgtx.so (hand-written HIP for gfx950) behind the C ABI in include/gtx.h.
"""
__version__ = "0.1.0"
