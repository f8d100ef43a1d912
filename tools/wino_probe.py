"""Winograd F(2x2, 3x3) split-f16x3 convolution (csrc/conv_wino_split.hip, GTX_WINO=1) against a float64 direct convolution and
against the direct split-f16x3 kernel, on ragged shapes (partial tiles, several cout tiles, residual, channel slices).
Usage: python tools/wino_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'geo-trax_amd'))
import numpy as np
from geotrax_amd import _lib, ops

ctx = _lib.default_context(0)


def ref_conv(x, w, b, act, res):
    n, h, wd, cin = x.shape
    cout = w.shape[0]
    xp = np.zeros((n, h + 2, wd + 2, cin), np.float64)
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((n, h, wd, cout), np.float64)
    for a in range(3):
        for c in range(3):
            y += np.einsum('nhwc,oc->nhwo', xp[:, a:a + h, c:c + wd], w[:, a, c].astype(np.float64))
    y += b
    if act:
        y = y / (1 + np.exp(-y))
    if res is not None:
        y = y + res
    return y


def pairs(x):   # what the pair format keeps of a float32 array: hi + lo
    hi = x.astype(np.float16).astype(np.float32)
    lo = (x - hi).astype(np.float16).astype(np.float32)
    return (hi.astype(np.float64) + lo.astype(np.float64))


rng = np.random.default_rng(0)
worst = 0.0
for (n, h, wd, cin, cout, act, use_res) in [(1, 8, 16, 16, 64, False, False), (1, 8, 16, 16, 64, True, False), (2, 20, 30, 64, 64, True, True),
                                            (1, 60, 60, 32, 128, True, False), (2, 37, 53, 128, 192, True, True), (1, 16, 32, 256, 64, False, False)]:
    x = (rng.standard_normal((n, h, wd, cin)) * np.exp(rng.uniform(-3, 3, (n, h, wd, cin)))).astype(np.float32)
    w = (rng.standard_normal((cout, 3, 3, cin)) / np.sqrt(9 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    res = (rng.standard_normal((n, h, wd, cout))).astype(np.float32) if use_res else None
    want = ref_conv(pairs(x), w, b.astype(np.float64), act, None if res is None else pairs(res))
    os.environ["GTX_WINO"] = "0"
    y0 = ops.conv2d(x, w, b, act=act, residual=res, split=True, ctx=ctx)
    os.environ["GTX_WINO"] = os.environ.get("WINO_MODE", "1")          # 1: 8 x 16-pixel form, 3: 16 x 16-pixel form
    y1 = ops.conv2d(x, w, b, act=act, residual=res, split=True, ctx=ctx)
    os.environ["GTX_WINO"] = "0"
    scale = np.abs(want).max()
    e0, e1 = np.abs(y0 - want).max() / scale, np.abs(y1 - want).max() / scale
    worst = max(worst, e1)
    print(f"n{n} {h}x{wd} {cin}->{cout} act={int(act)} res={int(use_res)}: direct split err {e0:.2e}  winograd err {e1:.2e}  (of layer max {scale:.3g})  "
          f"wino vs direct {np.abs(y1 - y0).max() / scale:.2e}")
print("worst winograd error / layer max:", f"{worst:.2e}", "OK" if worst < 2e-5 else "FAIL")
